#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "error_feedback" 2>&1 | tail -40 > gpurun_out/r06_t1.log
