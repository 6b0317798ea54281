#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python profiles/scripts/aten_ops_in_forward.py > gpurun_out/r06_aten.log 2>&1
