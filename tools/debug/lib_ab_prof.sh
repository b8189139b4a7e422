#!/bin/bash
# on the GPU box: per-kernel times of `python bench.py "$@"` (pass --steps N --warmup 2) once per nerf_amd/lib/ab_*.so
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
steps=10
for so in $R/nerf_amd/lib/ab_*.so; do
  n=$(basename $so .so)
  cp $so $R/nerf_amd/lib/libnerf_amd.so
  rm -rf $R/gpurun_out/$n
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$n -- python3 $R/bench.py "$@" --steps $steps --warmup 2 > $R/gpurun_out/$n.json 2> $R/gpurun_out/$n.err
  echo $n
  python3 $R/tools/debug/kernel_stats_per_step.py $R/gpurun_out/$n $((steps + 2)) | sed -n 1,5p
done
