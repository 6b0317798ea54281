#!/bin/bash
# Ablations of igemm_tr4_kernel<IGDN> (g_s deconv3 + IGDN, 128 -> 128 5x5 s2 on 128^2 inputs, B = 8, f16): the f16 library rebuilt with
# -DTR4_ABL=<bits> (conv_igemm.hip: 1 no LDS-DMA, 2 no K-loop MFMAs, 4 no fragment reads, 8 no epilogue), each variant swapped in for
# hesic_amd/libhesic_hip_f16.so on the (scratch) GPU box and timed with profiles/scripts/conv_layer_time.py from a HIP graph.
# Build the variants first (CPU container): profiles/scripts/tr4_ablation_build.sh; they live in
# profiles/abl_build/ (git-ignored, travels to the box).  Output: gpurun_out/tr4_ablation.txt
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/tr4_ablation.txt
mkdir -p gpurun_out
: > $OUT
cp hesic_amd/libhesic_hip_f16.so /tmp/base_f16.so
run() {
    for layer in deconv3 deconv_plain; do
        echo -n "abl=$1 " >> $OUT
        python profiles/scripts/conv_layer_time.py --layer $layer --size 128 --batch 8 --dtype f16 --graph --iters 20 2>&1 | tail -1 >> $OUT
    done
}
run 0
for abl in 1 2 4 8 3 15; do
    f=profiles/abl_build/libhesic_hip_f16_abl$abl.so
    [ -f $f ] || continue
    cp $f hesic_amd/libhesic_hip_f16.so
    run $abl
done
cp /tmp/base_f16.so hesic_amd/libhesic_hip_f16.so
run 0
cat $OUT
