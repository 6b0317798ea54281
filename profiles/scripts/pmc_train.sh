#!/bin/bash
# SQ counters of the training step's heavy kernels (wgrad_tr, gdn128_bwd, igemm forward / data gradient):
#   gpurun -- 'bash profiles/scripts/pmc_train.sh r04_e'
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04}_train_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-include-regex "wgrad_tr_kernel|gdn128_bwd_kernel|igemm_glds_kernel|igemm_tr4_kernel|wgrad_finish" --output-format csv -d $O/p$i -- python $GRAFT_REPO_ROOT/profiles/scripts/train_step.py --size 512 --only e --steps 2 > /dev/null 2>&1
done
python $GRAFT_REPO_ROOT/profiles/make_pmc_sq_json.py $O $O/pmc_sq_train.json > $O/pmc_sq_train.txt 2>&1
find $O -type f ! -name "*counter_collection.csv" ! -name "pmc_sq_train.*" -delete 2>/dev/null
du -sh $O
