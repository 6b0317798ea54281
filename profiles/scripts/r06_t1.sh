#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python profiles/scripts/memcpy_in_forward.py 2>&1 | grep -v amdgpu.ids | tail -40 > gpurun_out/r06_t1.log
