#!/bin/bash
# Build mlp_fp16x2.hip once per flag set into nerf_amd/lib/ab_<name>.so (a library of that one object).
#   bash tools/debug/h2_ab.sh base= look3=-DH2_LOOK=3 ring8=-DH2_RING=8
# then on the GPU box:  python tools/debug/x3_ablate.py --kind fp16x2 nerf_amd/lib/ab_*.so nerf_amd/lib/ab_*.so
set -e
cd "$(dirname "$0")/../../nerf_amd/csrc"
rm -f ../lib/ab_*.so
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -pragma-unroll-threshold=1000000 $flags -shared mlp_fp16x2.hip -o ../lib/ab_$name.so 2>&1 | grep -E "error" || true
done
ls ../lib
