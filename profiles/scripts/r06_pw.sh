cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r6_pw.log
[ -x profiles/scripts/micro/mfma_peak ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w profiles/scripts/micro/mfma_peak.hip -o profiles/scripts/micro/mfma_peak
poll() { for i in $(seq 1 $1); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | tr '\n' ' ' ; echo; done; }
for mode in "r" "r small" "z small"; do
  ( sleep 2; poll 3 ) >> gpurun_out/r6_pw.log &
  profiles/scripts/micro/mfma_peak 5 $mode >> gpurun_out/r6_pw.log 2>&1
  wait; sleep 2
done
