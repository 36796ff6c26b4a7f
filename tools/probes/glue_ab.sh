cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_glue.py tests/test_gpu_ops.py tests/test_gpu_token_encoder.py tests/test_gpu_student_scale.py tests/test_torch_library_cpp.py -x -q 2>&1 | tail -15
for c in 4 3; do echo "== cfg $c"; timeout 300 python tools/bench_student.py --config $c $( [ $c = 4 ] && echo --envs 512 ) 2>&1 | tail -1 | cut -c1-330; done
