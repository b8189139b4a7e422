#!/bin/bash
# on the GPU box: `bench.py --mode kilo` for the three tiny-MLP forms, once per nerf_amd/lib/ab_*.so, twice round-robin;
# prints frame ms and the stage times.  Build the variants with tools/debug/lib_ab.sh kilo.hip name=flags ...
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for so in $R/nerf_amd/lib/ab_*.so; do
    cp $so $R/nerf_amd/lib/libnerf_amd.so
    for dt in ${KILO_AB_DTYPES:-bf16x3 fp16x2 fp32}; do
      echo "$(basename $so .so) $dt $(timeout -k 10 100 python3 $R/bench.py --mode kilo --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), {k: round(v,4) for k,v in d["stage_ms"].items()})')"
    done
  done
done
