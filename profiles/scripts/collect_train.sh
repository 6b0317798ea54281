# training-side collection: bench line (+ with the one-rank RCCL group in the graph) and the rocprofv3 kernel stats of the graphed step
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r03h}; mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py --mode train > $O/bench_train.json 2>/dev/null
HESIC_FORCE_COLLECTIVES=1 python bench.py --mode train > $O/bench_train_rccl.json 2>/dev/null
python bench.py > $O/bench.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p3 -o t --output-format csv -- python $GRAFT_REPO_ROOT/profiles/scripts/train_step.py --size 512 --only g --steps 10 > /dev/null 2>&1
cp /tmp/p3/t_kernel_stats.csv $O/graphed_train_step_512_kernel_stats.csv
