cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp hesic_amd/libhesic_hip.so /tmp/lib_base.so
: > gpurun_out/r6_wg.log
for rep in 1 2; do
for v in base wg01 wg10 wg11; do
  if [ $v = base ]; then cp /tmp/lib_base.so hesic_amd/libhesic_hip.so; else cp profiles/scripts/micro/libhesic_hip_$v.so hesic_amd/libhesic_hip.so; fi
  python bench.py --mode train --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], {k:round(v,4) for k,v in d['losses_last_step'].items()})" >> gpurun_out/r6_wg.log 2>&1
done; done
cp /tmp/lib_base.so hesic_amd/libhesic_hip.so
