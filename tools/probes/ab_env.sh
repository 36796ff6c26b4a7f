#!/bin/bash
# same-box A/B of environment switches: ab_env.sh "IGI_LOSS_FUSED=0" "IGI_LOSS_FUSED=1" ...  (each argument: one setting,
# several VAR=value separated by commas); prints the update time and the per-kernel table of bench.py for each
cd "$(dirname "$0")/../.."
for setting in "$@"; do
  env $(echo "$setting" | tr ',' ' ') python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-student 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('== $setting : update ms', d['ms_per_step'], 'updates/s', d['value'], 'frac', d['whole_update_mfma_frac'])
for k in d['kernels']:
    if k['ms_per_update'] > 0.05: print('   %-34s n=%4d avg %7.2f us  %7.3f ms' % (k['name'][:34], k['launches_per_update'], k['avg_us'], k['ms_per_update']))
for l in d['roofline'].get('levels', []): print('   level %-28s %7.2f us frac %.3f' % (l['level'][:28], l['avg_us'], l['frac']))
"
done
