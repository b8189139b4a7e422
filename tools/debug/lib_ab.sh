#!/bin/bash
# Build the whole library once per flag set on ONE source file into nerf_amd/lib/ab_<name>.so (timing ablations / A-B):
#   bash tools/debug/lib_ab.sh mlp_fp16x2.hip base= wait60=-DH2_ACQ_WAIT=60
# then on the GPU box  bash tools/debug/lib_ab_run.sh <bench args>   (whole bench, alternating libraries)
#                  or  bash tools/debug/lib_ab_prof.sh <bench args>  (per-kernel times under rocprofv3).
# Both copy each variant over nerf_amd/lib/libnerf_amd.so ON THE BOX; here, delete the ab_*.so afterwards.
set -e
cd "$(dirname "$0")/../../nerf_amd/csrc"
src=$1; shift
rm -f ../lib/ab_*.so
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -pragma-unroll-threshold=1000000 $flags -c $src -o build/ab_$name.o
  objs=$(ls build/*.o | grep -v "build/ab_\|build/${src%.hip}.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab_$name.so $objs build/ab_$name.o
  rm build/ab_$name.o
done
ls ../lib
