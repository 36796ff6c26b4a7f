#!/bin/bash
# same-box A/B of environment switches on the student legs: ab_student.sh "IGI_MLP_CHAIN=0" "IGI_MLP_CHAIN=1" ...
cd "$(dirname "$0")/../.."
for setting in "$@"; do
  for leg in "4 512 32 64" "3 2048 32 64"; do
    set -- $leg
    env $(echo "$setting" | tr ',' ' ') python tools/bench_student.py --config $1 --envs $2 --hw $3 $4 --updates 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('== $setting config $1 envs $2: ms/step', d['ms_per_optimizer_step'], 'frac', d.get('frac_of_f32_mfma_peak'))
"
  done
done
