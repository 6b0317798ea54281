#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python profiles/scripts/tr4_determinism.py 2>&1 | grep -v amdgpu.ids | grep -v "   phase" > gpurun_out/r06_t1.log
for i in 1 2 3; do python profiles/scripts/tr4_determinism.py 2>&1 | grep -c "10 / 10   equals unfused: True" >> gpurun_out/r06_t1.log; done
run() { for sz in 128 256; do for l in deconv3 deconv_plain; do python profiles/scripts/conv_layer_time.py --layer $l --dtype f16 --graph --size $sz 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_t1.log; done; done; }
echo counted >> gpurun_out/r06_t1.log; run
cp hesic_amd/libhesic_hip_f16.so /tmp/keep.so; cp profiles/scripts/micro/libhesic_hip_f16_wait0.so hesic_amd/libhesic_hip_f16.so
echo wait0 >> gpurun_out/r06_t1.log; run
cp /tmp/keep.so hesic_amd/libhesic_hip_f16.so
echo counted >> gpurun_out/r06_t1.log; run
