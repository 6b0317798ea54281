cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp hesic_amd/libhesic_hip_f16.so /tmp/lib_base.so
: > gpurun_out/r6_ig.log
for rep in 1 2; do
for v in base $VARIANTS; do
  if [ $v = base ]; then cp /tmp/lib_base.so hesic_amd/libhesic_hip_f16.so; else cp profiles/scripts/micro/libhesic_hip_f16_$v.so hesic_amd/libhesic_hip_f16.so; fi
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-secondary --exec eager 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$v', d['ms_per_step'], r['kernel'], r['avg_launch_us'], {k:v['avg_us'] for k,v in r['other_conv_kernels'].items()}, d['gpu_metrics_last_batch'], d.get('parity'))" >> gpurun_out/r6_ig.log 2>&1
done; done
cp /tmp/lib_base.so hesic_amd/libhesic_hip_f16.so
