#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python bench.py > gpurun_out/r06_bench_f.json 2> gpurun_out/r06_bench_f.err
