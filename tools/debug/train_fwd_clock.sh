#!/bin/bash
# sclk (rocm-smi, twice a second) while the training forward / the inference forward of one dtype runs alone on the card:
#   bash tools/debug/train_fwd_clock.sh bf16x3   -> gpurun_out/fwdclk_<kind>_{train,infer}.txt
KIND=$1
R="${GRAFT_REPO_ROOT:-$PWD}"; OUT=$R/gpurun_out; mkdir -p $OUT
for what in train infer; do
  ( while true; do /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | sed 's/GPU\[0\]\t*: //' | tr '\n' ' '; echo; sleep 0.5; done ) > $OUT/fwdclk_${KIND}_$what.txt &
  SMI=$!
  timeout -k 10 120 python3 $R/tools/debug/train_fwd_loop.py $KIND 8 $what || { kill $SMI; exit 1; }
  kill $SMI
  tail -n 8 $OUT/fwdclk_${KIND}_$what.txt | cut -c1-200
done
