for cfg in 1024,512 1536,512 2048,512 2048,1024 1024,256 768,384; do
  echo -n "slots $cfg: "; IGI_SPLIT_SLOTS=$cfg python bench.py --no-cpu-baseline --no-student 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={c['name']:c for c in d['kernels']}
print(d['value'], 'wgrad %.1f'%k['gemm_dma_wgrad_multi_kernel']['avg_us'], 'reduce %.1f'%k['k_slab_reduce']['avg_us'])"
done
echo -n "per-product: "; IGI_JOINT_SPLIT=0 python bench.py --no-cpu-baseline --no-student 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k={c['name']:c for c in d['kernels']}
print(d['value'], 'wgrad %.1f'%k['gemm_dma_wgrad_multi_kernel']['avg_us'], 'reduce %.1f'%k['k_slab_reduce']['avg_us'])"
