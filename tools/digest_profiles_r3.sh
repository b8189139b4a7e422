#!/bin/bash
# In the build container, after tools/collect_profiles_r3.sh ran on the GPU box: gpurun_out/r3_* -> profiles/r03_* (each
# traffic / counter record carries the hash of the kernel sources in the tree, nerf_amd/source_hash.py).
set -e
cd "$(dirname "$0")/.."
declare -A KERNEL=([bf16x3]=mlp_bf16x3_m16_kernel [bf16]=mlp_bf16_kernel [fp16x2]=mlp_fp16x2_kernel [fp32]=mlp_fwd_kernel)
for dt in bf16x3 bf16 fp16x2 fp32; do
  if [ $dt = bf16x3 ]; then extra=profiles/r03_composite_traffic_pmc.json; else extra=; fi
  python3 tools/bench_traffic_digest.py gpurun_out/r3_traffic_$dt ${KERNEL[$dt]} $dt $extra > profiles/r03_${dt}_traffic_pmc.json
  cp gpurun_out/r3_traffic_$dt/kernel_stats.csv profiles/r03_bench_${dt}_kernel_stats.csv
done
python3 tools/pmc_digest.py gpurun_out/r3_kilo_h2 --family=kilo > profiles/r03_kilo_h2_pmc.json
python3 tools/pmc_digest.py gpurun_out/r3_kilo_f32 --family=kilo > profiles/r03_kilo_pmc.json
python3 tools/pmc_digest.py gpurun_out/r3_kilo_x3 --family=kilo > profiles/r03_kilo_x3_pmc.json
python3 tools/pmc_digest.py gpurun_out/r3_bf16 mlp_bf16 --family=bf16 > profiles/r03_bf16_pmc.json
python3 tools/pmc_digest.py gpurun_out/r3_bf16x3 mlp_bf16x3 --family=bf16x3 > profiles/r03_bf16x3_pmc.json
cp gpurun_out/r3_train_x3_per_step.txt profiles/r03_train_x3_per_step.txt
cp gpurun_out/r3_train_x3_kernel_stats.csv profiles/r03_train_x3_kernel_stats.csv
cp gpurun_out/r3_train_h2_per_step.txt profiles/r03_train_h2_per_step.txt
cp gpurun_out/r3_train_h2_kernel_stats.csv profiles/r03_train_h2_kernel_stats.csv
for dt in bf16x3 fp16x2 fp32; do
  cp gpurun_out/r3_kilo_${dt}_frame_per_step.txt profiles/r03_kilo_${dt}_per_step.txt
  cp gpurun_out/r3_kilo_${dt}_frame_kernel_stats.csv profiles/r03_kilo_${dt}_kernel_stats.csv
done
ls -la profiles | grep r03_
