#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_f16.py tests/test_gpu_baseline_workloads.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r06_t1.log
python bench.py --no-cpu-baseline --no-secondary --no-power-state --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['gpu_metrics_last_batch'], d['roofline']['streaming_kernels']['conv1_3to128_gdn hi/lo (n2w, x3)']['avg_launch_us'])" >> gpurun_out/r06_t1.log
