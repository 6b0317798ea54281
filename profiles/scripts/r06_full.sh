cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6_gpu_tests.log
ROUND_TAG=${ROUND_TAG:-r06_a} PMC_COMMIT=${PMC_COMMIT:-unknown} timeout 2400 bash profiles/scripts/collect_r06.sh > gpurun_out/r6_collect.log 2>&1
