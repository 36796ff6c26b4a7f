cd $GRAFT_REPO_ROOT
run() { echo "== c$1 ${3:+envs $3 }IGI_TAC_SK=$2"; IGI_TAC_SK=$2 timeout 300 python tools/bench_student.py --config $1 ${3:+--envs $3} 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_optimizer_step'], d['frac_of_f32_mfma_peak'], [ (k['name'][16:],k['avg_us']) for k in d['native_kernels'] if 'false,false,5' in k['name']])"; }
run 4 512,0,0 512
run 4 512,256,0 512
run 4 512,256,170 512
run 4 512,512,340 512
run 4 512,192,128 512
run 3 512,512,340
run 3 512,384,255
