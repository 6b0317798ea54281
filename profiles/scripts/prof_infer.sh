#!/bin/bash
# single-stream kernel stats of the default inference step:  TAG=r04_b ANALYSIS=x3c2 DTYPE=f16 bash profiles/scripts/prof_infer.sh
O=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r04}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
HESIC_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --exec eager --dtype ${DTYPE:-f16} --analysis ${ANALYSIS:-x3c2} > /dev/null 2>&1
cp /tmp/p2/s_kernel_stats.csv $O/single_stream_kernel_stats_${DTYPE:-f16}_${ANALYSIS:-x3c2}.csv
head -40 $O/single_stream_kernel_stats_${DTYPE:-f16}_${ANALYSIS:-x3c2}.csv | cut -c1-200
