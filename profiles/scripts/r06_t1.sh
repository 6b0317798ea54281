#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python profiles/scripts/layer_power.py 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r06_t1.log
