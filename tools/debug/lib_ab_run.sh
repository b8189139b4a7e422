#!/bin/bash
# on the GPU box: run `python bench.py "$@"` once per nerf_amd/lib/ab_*.so, twice round-robin (one box, one process each):
# boxes differ by +-3 %, so only runs on the same box compare.  Rebuild with make afterwards.
set -e
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for so in $R/nerf_amd/lib/ab_*.so; do
    cp $so $R/nerf_amd/lib/libnerf_amd.so
    echo "$(basename $so .so) $(timeout -k 10 200 python3 $R/bench.py "$@" 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["frac"])')"
  done
done
