#!/bin/bash
# Round-4 collection: bench lines of every mode, rocprofv3 kernel stats, HBM traffic (PMC) and SQ counters of the default (f16, x3c2) step.
#   gpurun -- 'ROUND_TAG=r04_d bash profiles/scripts/collect_r04.sh'
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r04}; mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py --parity-trained 2 > $O/bench.json 2> $O/bench.err
python bench.py --analysis x3 --no-secondary > $O/bench_f16_x3.json 2>/dev/null
python bench.py --analysis x1 --no-secondary --no-cpu-baseline > $O/bench_f16_x1.json 2>/dev/null
python bench.py --dtype bf16 --no-secondary > $O/bench_bf16_x3.json 2>/dev/null
python bench.py --model joint --batch 4 > $O/bench_joint.json 2>/dev/null
python bench.py --model joint --batch 8 --no-cpu-baseline > $O/bench_joint_b8.json 2>/dev/null
python bench.py --dtype f32 --steps 20 --no-secondary > $O/bench_f32.json 2>/dev/null
python bench.py --sweep --batch 4 --steps 16 > $O/bench_c5_sweep.json 2>/dev/null
python bench.py --sweep --model joint --batch 4 --steps 16 > $O/bench_c5_sweep_joint.json 2>/dev/null
HESIC_FORCE_COLLECTIVES=1 python bench.py --mode train > $O/bench_train.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o d --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --exec eager > /dev/null 2>&1
cp /tmp/p1/d_kernel_stats.csv $O/default_kernel_stats.csv
HESIC_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --exec eager > /dev/null 2>&1
cp /tmp/p2/s_kernel_stats.csv $O/single_stream_kernel_stats.csv
RX="igemm_glds_kernel|igemm_tr4_kernel|n2w_gdn_hilo"
HESIC_NO_OVERLAP=1 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RX" --output-format csv -d $O/pmc_f -- python $GRAFT_REPO_ROOT/profiles/scripts/forward_n.py hsic 4 > /dev/null 2>&1
HESIC_NO_OVERLAP=1 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$RX" --output-format csv -d $O/pmc_w -- python $GRAFT_REPO_ROOT/profiles/scripts/forward_n.py hsic 4 > /dev/null 2>&1
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  HESIC_NO_OVERLAP=1 timeout 300 rocprofv3 --pmc $set --kernel-include-regex "$RX|sconv_n2w|sconv_w2n|warp_fwd|sconv_6to3" --output-format csv -d $O/pmcsq/p$i -- python $GRAFT_REPO_ROOT/profiles/scripts/forward_n.py hsic 3 > /dev/null 2>&1
done
python $GRAFT_REPO_ROOT/profiles/make_pmc_sq_json.py $O/pmcsq $O/pmc_sq.json > $O/pmc_sq.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p3 -o t --output-format csv -- python $GRAFT_REPO_ROOT/profiles/scripts/train_step.py --size 512 --only g --steps 10 > /dev/null 2>&1
cp /tmp/p3/t_kernel_stats.csv $O/graphed_train_step_512_kernel_stats.csv
# keep the per-dispatch PMC csvs small: only the counter collection files
find $O/pmc_f $O/pmc_w $O/pmcsq -type f ! -name "*counter_collection.csv" -delete 2>/dev/null
du -sh $O
