#!/bin/bash
cd /root/repo
mkdir -p gpurun_out; : > gpurun_out/r06_t1.log
run() { for sz in 128 256; do for l in deconv3 deconv_plain; do python profiles/scripts/conv_layer_time.py --layer $l --dtype f16 --graph --size $sz 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_t1.log; done; done; }
cp hesic_amd/libhesic_hip_f16.so /tmp/keep.so; cp profiles/scripts/micro/libhesic_hip_f16_hdma.so hesic_amd/libhesic_hip_f16.so
for i in 1 2 3; do python profiles/scripts/tr4_determinism.py 2>&1 | grep "float16" | grep -c "10 / 10   equals unfused: True" >> gpurun_out/r06_t1.log; done
echo dma >> gpurun_out/r06_t1.log; run
cp /tmp/keep.so hesic_amd/libhesic_hip_f16.so
echo regs >> gpurun_out/r06_t1.log; run
cp profiles/scripts/micro/libhesic_hip_f16_hdma.so hesic_amd/libhesic_hip_f16.so
echo dma >> gpurun_out/r06_t1.log; run
cp /tmp/keep.so hesic_amd/libhesic_hip_f16.so
