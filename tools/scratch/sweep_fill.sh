for f in 128 256 512; do
  echo -n "BN64_FILL $f: "; IGI_BN64_FILL=$f python bench.py --no-cpu-baseline --no-student 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], [(c['name'][-22:], c['launches_per_update'], c['avg_us']) for c in d['kernels'] if 'gemm_dma_kernel' in c['name']])"
done
