#!/bin/bash
cd /root/repo
mkdir -p gpurun_out; : > gpurun_out/r06_t1.log
cp hesic_amd/libhesic_hip_f16.so /tmp/keep.so
for rep in 1 2; do for v in base nb; do
  if [ $v = base ]; then cp /tmp/keep.so hesic_amd/libhesic_hip_f16.so; else cp profiles/scripts/micro/libhesic_hip_f16_$v.so hesic_amd/libhesic_hip_f16.so; fi
  python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-secondary --no-power-state --exec eager 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], d['roofline']['streaming_kernels']['conv1_3to128_gdn hi/lo (n2w, x3)']['avg_launch_us'], d['gpu_metrics_last_batch'])" >> gpurun_out/r06_t1.log
done; done
cp /tmp/keep.so hesic_amd/libhesic_hip_f16.so
