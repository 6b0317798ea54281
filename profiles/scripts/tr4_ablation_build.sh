#!/bin/bash
# CPU container: build the f16 library variants profiles/scripts/tr4_ablation.sh times (conv_igemm.hip with -DTR4_ABL=<bits>, the other objects
# of the normal f16 build re-used).  ~2 minutes per variant; output profiles/abl_build/ (git-ignored, ~6 MB each -- delete after the run).
set -e
cd "$(dirname "$0")/../../hesic_amd/csrc"
make -j4 > /dev/null
mkdir -p ../../profiles/abl_build
for abl in ${@:-1 2 4 8 3 15}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DHESIC_H16_IS_F16=1 -DTR4_ABL=$abl -c conv_igemm.hip -o /tmp/ci_f16_$abl.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls f16/*.o | grep -v conv_igemm) /tmp/ci_f16_$abl.o -o ../../profiles/abl_build/libhesic_hip_f16_abl$abl.so
    echo "built abl=$abl"
done
