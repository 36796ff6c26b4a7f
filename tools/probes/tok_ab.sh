cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_token_encoder.py -x -q 2>&1 | tail -15
for m in 0 1; do echo "== IGI_TOKEN_FUSED_BWD=$m"; IGI_TOKEN_FUSED_BWD=$m timeout 300 python tools/bench_student.py --config 4 --envs 512 2>&1 | tail -1 | cut -c1-250; done
for m in 0 1; do echo "== cfg3 IGI_TOKEN_FUSED_BWD=$m"; IGI_TOKEN_FUSED_BWD=$m timeout 300 python tools/bench_student.py --config 3 2>&1 | tail -1 | cut -c1-250; done
