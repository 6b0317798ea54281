#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r06_t1.log
