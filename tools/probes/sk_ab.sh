# A/B of the weight-gradient split factors (IGI_SK_OVERRIDE = "env0,env1,env2,trunk0,trunk1,trunk2", 0 = planner's choice):
# update time, per-level launch times and the slab sum, two rounds on one box.
for rep in 1 2; do
for v in ${SK_SET:-"0,0,0,0,0,0" "0,64,0,0,0,0" "0,32,0,0,0,0" "0,64,32,0,0,0" "0,64,0,64,0,0" "0,64,0,0,32,0" "0,64,0,0,0,128"}; do
IGI_SK_OVERRIDE=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-student 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
lv={l['level'][:6]:l['avg_us'] for l in d['roofline']['levels']}
ks={k['name']:k['avg_us'] for k in d['kernels']}
print('$v', d['ms_per_step'], lv, 'slab', ks.get('k_slab_reduce'))
"
done; done
