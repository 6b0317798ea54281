#!/bin/bash
# As dyn_sq_ab.sh, per kernel: rocprofv3 kernel stats of 8 HESIC+ B=4 forwards with either library (the pair GDN launches and the total).
cd $GRAFT_REPO_ROOT
cp hesic_amd/libhesic_hip_f16.so /tmp/base.so
export TMPDIR=/tmp
for v in dyn nodyn; do
  if [ $v = nodyn ]; then cp profiles/abl_build/libhesic_hip_f16_nodyn.so hesic_amd/libhesic_hip_f16.so; else cp /tmp/base.so hesic_amd/libhesic_hip_f16.so; fi
  (cd /tmp; HESIC_BATCH=4 HESIC_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats -d /tmp/k$v -o t --output-format csv -- python $GRAFT_REPO_ROOT/profiles/scripts/forward_n.py joint 8 > /dev/null 2>&1)
  python - <<PY
import csv,glob
f=glob.glob("/tmp/k$v/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    tot+=float(r["TotalDurationNs"])
    n=r["Name"]
    if "n2w_gdn_hilo" in n or "ELi3ELi" in n.replace(" ","") or ", 3, 4, 0, 1>" in n or ", 3, 8, 0, 1>" in n:
        print("$v", n[:75], r["Calls"], r["AverageNs"])
print("$v total kernel ns per forward", tot/8)
PY
done
cp /tmp/base.so hesic_amd/libhesic_hip_f16.so
