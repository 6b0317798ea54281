#!/bin/bash
cd /root/repo
mkdir -p gpurun_out; : > gpurun_out/r06_t1.log
cp hesic_amd/libhesic_hip_f16.so /tmp/keep.so
for rep in 1 2; do for v in base b256; do
  if [ $v = base ]; then cp /tmp/keep.so hesic_amd/libhesic_hip_f16.so; else cp profiles/scripts/micro/libhesic_hip_f16_$v.so hesic_amd/libhesic_hip_f16.so; fi
  python bench.py --model joint --batch 4 --steps 100 --warmup 20 --no-cpu-baseline --no-secondary --no-power-state 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], d['gpu_metrics_last_batch'], d['config']['issue'])" >> gpurun_out/r06_t1.log
done; done
cp /tmp/keep.so hesic_amd/libhesic_hip_f16.so
