# A/B of one environment switch on one box: bash tools/probes/ab_env.sh VAR "v1 v2 ..." [rounds]
# prints the update time, the per-level launch times and a few kernel classes for every value, `rounds` times.
VAR=$1; VALS=$2; ROUNDS=${3:-2}
for rep in $(seq $ROUNDS); do
for v in $VALS; do
env $VAR=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-student 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
lv={l['level'][:6]:l['avg_us'] for l in d['roofline']['levels']}
ks={k['name']:k['avg_us'] for k in d['kernels']}
print('$VAR=$v', d['ms_per_step'], lv, {n[:28]:a for n,a in ks.items() if n.startswith('gemm_dma_kernel') or n in ('k_slab_reduce','k_env_fwd')})
"
done; done
