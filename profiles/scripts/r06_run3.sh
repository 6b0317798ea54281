cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q 2>&1 | tail -40 > gpurun_out/r6_multirank.log
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_multirank.py 2>&1 | tail -15 > gpurun_out/r6_gpu_tests2.log
