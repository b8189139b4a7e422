#!/bin/bash
# Build the whole library once per flag set on mlp_bwd.hip into nerf_amd/lib/wg_<name>.so (timing-only ablations of
# wgrad_h2p_kernel: WG_ABL_NOLOAD / NOSPLIT / NOBAR).   bash tools/debug/wg_ab.sh base= noload=-DWG_ABL_NOLOAD
# On the GPU box copy one over nerf_amd/lib/libnerf_amd.so per bench run; rebuild with make afterwards.
set -e
cd "$(dirname "$0")/../../nerf_amd/csrc"
rm -f ../lib/wg_*.so
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -pragma-unroll-threshold=1000000 $flags -c mlp_bwd.hip -o build/wg_$name.o
  objs=$(ls build/*.o | grep -v "build/wg_\|build/mlp_bwd.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/wg_$name.so $objs build/wg_$name.o
  rm build/wg_$name.o
done
ls ../lib
