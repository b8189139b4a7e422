#!/bin/bash
# Build mlp_bf16.hip once per flag set into nerf_amd/lib/ab_<name>.so (the other objects come from the last make).
#   bash tools/debug/bf16_ab.sh base= drain=-DBF_DRAIN_LDS look5=-DBF_LOOK=5
# then on the GPU box:  python tools/debug/bf16_ablate.py nerf_amd/lib/ab_*.so nerf_amd/lib/ab_*.so   (interleaved, one process)
set -e
cd "$(dirname "$0")/../../nerf_amd/csrc"
OTHERS="build/api.o build/mlp_fp32.o build/mlp_bwd.o build/train.o build/kilo.o build/kilo_bwd.o build/occupancy.o build/composite.o build/sampling.o build/raygen.o"
rm -f ../lib/ab_*.so
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 $flags -c mlp_bf16.hip -o build/ab_$name.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab_$name.so $OTHERS build/ab_$name.o
  rm -f build/ab_$name.o
done
ls ../lib
