#!/bin/bash
# A/B of the per-pixel scaling of the pair (I)GDN squares (binary16 build): the f16 library against one built with -DHESIC_NO_DYN_SQ=1
# (conv_igemm.hip + sconv_hilo.hip recompiled with that flag, linked with the other f16 objects into profiles/abl_build/libhesic_hip_f16_nodyn.so),
# swapped in on the (scratch) GPU box: HESIC B=8 and HESIC+ B=4 bench lines, twice each.
cd $GRAFT_REPO_ROOT
cp hesic_amd/libhesic_hip_f16.so /tmp/base.so
for r in 1 2; do
  for v in dyn nodyn; do
    if [ $v = nodyn ]; then cp profiles/abl_build/libhesic_hip_f16_nodyn.so hesic_amd/libhesic_hip_f16.so; else cp /tmp/base.so hesic_amd/libhesic_hip_f16.so; fi
    python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v hsic b8', d['value'], d['ms_per_step'])"
    python bench.py --model joint --batch 4 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v joint b4', d['value'], d['ms_per_step'])"
  done
done
cp /tmp/base.so hesic_amd/libhesic_hip_f16.so
