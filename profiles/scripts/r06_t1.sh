#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python bench.py --parity-trained 4 --no-secondary --no-power-state > gpurun_out/r06_trained.json 2> gpurun_out/r06_trained.err
