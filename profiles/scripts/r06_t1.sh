#!/bin/bash
cd /root/repo
mkdir -p gpurun_out; : > gpurun_out/r06_t1.log
cp hesic_amd/libhesic_hip_f16.so /tmp/keep.so
for rep in 1 2; do for v in base $VARIANTS; do
  if [ $v = base ]; then cp /tmp/keep.so hesic_amd/libhesic_hip_f16.so; else cp profiles/scripts/micro/libhesic_hip_f16_$v.so hesic_amd/libhesic_hip_f16.so; fi
  echo -n "$v  " >> gpurun_out/r06_t1.log; python profiles/scripts/en_conv_time.py 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r06_t1.log
done; done
cp /tmp/keep.so hesic_amd/libhesic_hip_f16.so
