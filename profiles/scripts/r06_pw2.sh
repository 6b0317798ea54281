cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/r6_pw2.log
poll() { for i in $(seq 1 $1); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/clock level: 1://; s/Current Socket Graphics Package Power//'; echo; done; }
for mode in r m3 m5 m7 m10 r; do
  ( sleep 1.5; poll 2 ) >> gpurun_out/r6_pw2.log &
  profiles/scripts/micro/mfma_peak 4 $mode >> gpurun_out/r6_pw2.log 2>&1
  wait; sleep 1
done
