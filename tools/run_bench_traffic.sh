#!/bin/bash
# HBM traffic of the MLP launches of the bench command itself: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes,
# plus the kernel-trace stats of the same command.  Output under gpurun_out/<tag>/; tools/bench_traffic_digest.py turns
# it into profiles/<round>_<dtype>_traffic_pmc.json.
#   bash tools/run_bench_traffic.sh <tag> <dtype> 
set -e
TAG=$1; DT=$2
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
[ -z "$ROOT" ] && ROOT=/root/repo
cd /tmp && export TMPDIR=/tmp
ARGS="--dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-also"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
echo "done $OUT"
