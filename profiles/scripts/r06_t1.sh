#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python bench.py --mode train > gpurun_out/r06_train_ps.json 2> gpurun_out/r06_train_ps.err
