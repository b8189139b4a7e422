#!/bin/bash
# Clock and power of the GPU while a bench mode runs: rocm-smi sampled twice a second beside `python bench.py "$@"`.
#   bash tools/debug/smi_during_bench.sh <tag> <bench args...>   -> gpurun_out/<tag>_smi.txt
TAG=$1; shift
R="${GRAFT_REPO_ROOT:-$PWD}"; OUT=$R/gpurun_out; mkdir -p $OUT
( while true; do /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|Power \(|Socket Power|mclk|Temperature \(Sensor junction\)" | sed 's/GPU\[0\]\t*: //' | tr '\n' ' '; echo; sleep 0.5; done ) > $OUT/${TAG}_smi.txt &
SMI=$!
timeout -k 10 300 python3 $R/bench.py "$@" > $OUT/${TAG}_bench.json 2>/dev/null
kill $SMI
tail -n 25 $OUT/${TAG}_smi.txt | cut -c1-400
