#!/bin/bash
# Per-kernel time of one bench.py mode (rocprofv3 --kernel-trace --stats), digested per step.
#   tools/profile_bench.sh <tag> <steps> <warmup> <bench args...>   -> gpurun_out/<tag>_per_step.txt, _kernel_stats.csv, _bench.json
set -e
TAG=$1; STEPS=$2; WARM=$3; shift 3
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py "$@" --steps $STEPS --warmup $WARM --no-cpu-baseline > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_prof.err"
python3 $R/tools/debug/kernel_stats_per_step.py /tmp/prof_$TAG $((STEPS + WARM)) > "$OUT/${TAG}_per_step.txt"
cp "$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
head -16 "$OUT/${TAG}_per_step.txt"
