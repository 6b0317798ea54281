#!/bin/bash
# Round-6 collection: the default bench line (f16 maps, pair analysis; its `secondary` block carries C4, the training step, the other C2 modes,
# the C5 sweeps, path A and the RCCL-in-graph step), rocprofv3 kernel stats of the inference step and of the graphed training step, HBM traffic
# (PMC, separate passes) and SQ counters of the heavy kernels of both.
#   gpurun -- 'ROUND_TAG=r06_a bash profiles/scripts/collect_r05.sh'
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r06}; mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --mode train > $O/bench_train.json 2>/dev/null
python bench.py --model joint --batch 4 > $O/bench_joint.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
HESIC_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d /tmp/p2 -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --exec eager > /dev/null 2>&1
cp /tmp/p2/s_kernel_stats.csv $O/single_stream_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p1 -o d --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --exec eager > /dev/null 2>&1
cp /tmp/p1/d_kernel_stats.csv $O/default_kernel_stats.csv
# round 6: the reference's call order over the drop-in modules (path A, hand-over between modules), and the enhancement stage
rocprofv3 --kernel-trace --stats -d /tmp/p4 -o a --output-format csv -- python $GRAFT_REPO_ROOT/profiles/scripts/path_a_n.py hsic 5 > /dev/null 2>&1
cp /tmp/p4/a_kernel_stats.csv $O/path_a_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p5 -o e --output-format csv -- python $GRAFT_REPO_ROOT/profiles/scripts/en_forward_n.py 5 > /dev/null 2>&1
cp /tmp/p5/e_kernel_stats.csv $O/independent_en_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d /tmp/p3 -o t --output-format csv -- python $GRAFT_REPO_ROOT/profiles/scripts/train_step.py --size 512 --only g --steps 10 > /dev/null 2>&1
cp /tmp/p3/t_kernel_stats.csv $O/graphed_train_step_512_kernel_stats.csv
python $GRAFT_REPO_ROOT/profiles/scripts/train_timeline.py /tmp/p3 > $O/train_step_timeline.txt 2>&1
RX="igemm_glds_kernel|igemm_tr4_kernel|n2w_gdn_hilo"
HESIC_NO_OVERLAP=1 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RX" --output-format csv -d $O/pmc_f -- python $GRAFT_REPO_ROOT/profiles/scripts/forward_n.py hsic 4 > /dev/null 2>&1
HESIC_NO_OVERLAP=1 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$RX" --output-format csv -d $O/pmc_w -- python $GRAFT_REPO_ROOT/profiles/scripts/forward_n.py hsic 4 > /dev/null 2>&1
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  HESIC_NO_OVERLAP=1 timeout 300 rocprofv3 --pmc $set --kernel-include-regex "$RX|sconv_n2w|sconv_w2n|warp_fwd|sconv_6to3" --output-format csv -d $O/pmcsq/p$i -- python $GRAFT_REPO_ROOT/profiles/scripts/forward_n.py hsic 3 > /dev/null 2>&1
  timeout 400 rocprofv3 --pmc $set --kernel-include-regex "wgrad_tr_|wgrad_row_kernel|wgrad_nw_fused|gdn128_bwd_kernel|wgrad_finish" --output-format csv -d $O/pmctr/p$i -- python $GRAFT_REPO_ROOT/profiles/scripts/train_step.py --size 512 --only e --steps 2 > /dev/null 2>&1
done
j=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  j=$((j+1))
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "c32_" --output-format csv -d $O/pmcen/p$j -- python $GRAFT_REPO_ROOT/profiles/scripts/en_forward_n.py 2 > /dev/null 2>&1
done
python $GRAFT_REPO_ROOT/profiles/make_pmc_sq_json.py $O/pmcen $O/pmc_sq_en.json > $O/pmc_sq_en.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "c32_" --output-format csv -d $O/pmc_en_f -- python $GRAFT_REPO_ROOT/profiles/scripts/en_forward_n.py 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "c32_" --output-format csv -d $O/pmc_en_w -- python $GRAFT_REPO_ROOT/profiles/scripts/en_forward_n.py 2 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/profiles/make_pmc_sq_json.py $O/pmcsq $O/pmc_sq.json > $O/pmc_sq.txt 2>&1
python $GRAFT_REPO_ROOT/profiles/make_pmc_sq_json.py $O/pmctr $O/pmc_sq_train.json > $O/pmc_sq_train.txt 2>&1
PMC_COMMIT=${PMC_COMMIT:-unknown} PMC_COLLECTED=${ROUND_TAG:-r06} python $GRAFT_REPO_ROOT/profiles/make_pmc_json.py $O/pmc_f $O/pmc_w hsic_f16_b8_512 > $O/pmc_igemm.txt 2>&1 || true
cp $GRAFT_REPO_ROOT/profiles/pmc_igemm.json $O/pmc_igemm.json
find $O/pmc_f $O/pmc_w $O/pmcsq $O/pmctr $O/pmcen $O/pmc_en_f $O/pmc_en_w -type f ! -name "*counter_collection.csv" -delete 2>/dev/null
du -sh $O
