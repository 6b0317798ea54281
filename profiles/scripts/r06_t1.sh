#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -k "en or c32 or enh or stage2 or Enh" 2>&1 | tail -4 > gpurun_out/r06_t1.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p5 -o e --output-format csv -- python /root/repo/profiles/scripts/en_forward_n.py 5 > /dev/null 2>&1
head -8 /tmp/p5/e_kernel_stats.csv | cut -c1-200 >> /root/repo/gpurun_out/r06_t1.log
