cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -s -k "c32 or independent_en or stage2 or resblock" 2>&1 | grep -E "measured|passed|failed|Error|assert" | tail -40 > gpurun_out/r6_en_tests.log
