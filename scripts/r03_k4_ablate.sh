#!/bin/bash
# K4 ablation (round 3): which part of the main pass carries its time?  Builds one library per K4_ABLATE mask (k_backward.hip; the results of
# these builds are WRONG by construction, only their timing is read) and times form 2 with each.
#   scripts/r03_k4_ablate.sh build        (here: cross-compiles into scripts/micro/k4ab/)
#   scripts/r03_k4_ablate.sh run          (on the GPU box)
set -e
cd "$(dirname "$0")/.."
MASKS="0 1 2 3 4 8 16 32 7 63"
D=scripts/micro/k4ab
if [ "$1" = build ]; then
    mkdir -p $D
    FL="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
    for m in $MASKS; do
        ( cd dsac_amd/csrc && /opt/rocm/bin/hipcc $FL -DK4_ABLATE=$m -c k_backward.hip -o ../../$D/k_backward_$m.o 2>&1 | grep -v warning | head -5
          /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wl,--version-script=exports.map build/k_forward.o build/k_sample.o ../../$D/k_backward_$m.o \
              build/k_refine.o build/k_loss.o build/k_patches.o build/api.o -o ../../$D/libdsac_hip_ab$m.so ) &
    done
    wait
    ls -la $D/*.so
else
    for m in $MASKS; do
        echo "== K4_ABLATE=$m"
        DSAC_HIP_LIB=$PWD/$D/libdsac_hip_ab$m.so python scripts/r03_k4_sweep.py 2 2>/dev/null | grep "N= 256" | tail -2
    done
fi
