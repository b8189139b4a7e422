#!/bin/bash
# on the GPU box: time the training bench's kernels once per nerf_amd/lib/wg_*.so (see wg_ab.sh)
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for so in $R/nerf_amd/lib/wg_*.so; do
  n=$(basename $so .so)
  cp $so $R/nerf_amd/lib/libnerf_amd.so
  rm -rf $R/gpurun_out/$n
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$n -- python3 $R/bench.py --mode train --steps 10 --warmup 2 > $R/gpurun_out/$n.json 2> $R/gpurun_out/$n.err
  echo $n
  python3 $R/tools/debug/kernel_stats_per_step.py $R/gpurun_out/$n 12 | sed -n 1,4p
done
