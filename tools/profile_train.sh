#!/bin/bash
# Per-kernel time of the training step (rocprofv3 --kernel-trace --stats), digested per step.
#   tools/profile_train.sh <dtype> <tag>     -> gpurun_out/<tag>_per_step.txt, gpurun_out/<tag>_kernel_stats.csv
set -e
DTYPE=${1:-bf16x3}; TAG=${2:-train_prof}
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --mode train --dtype "$DTYPE" --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_prof.err"
python3 $R/tools/debug/kernel_stats_per_step.py /tmp/prof_$TAG 23 > "$OUT/${TAG}_per_step.txt"
cp "$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
head -14 "$OUT/${TAG}_per_step.txt"
