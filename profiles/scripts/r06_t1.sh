#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r06_bench_ps.json 2> gpurun_out/r06_bench_ps.err
