#!/bin/bash
# On the GPU box: every rocprofv3 pass the committed round-3 profiles come from (traffic of the bench command per dtype,
# counters of the KiloNeRF frames and the bf16-rate MLP kernels, per-kernel times of the training steps and KiloNeRF frames).
# Output under gpurun_out/r3_*; tools/digest_profiles_r3.sh (run in the build container) turns it into profiles/r03_*.
set -e
cd $GRAFT_REPO_ROOT
for dt in bf16x3 bf16 fp16x2 fp32; do bash tools/run_bench_traffic.sh r3_traffic_$dt $dt > /dev/null; echo traffic $dt done; done
bash tools/run_pmc.sh r3_kilo_h2 kilo kilo > /dev/null; echo kilo h2 done
bash tools/run_pmc.sh r3_kilo_f32 kilo kilo 65536 "--kilo-dtype fp32" > /dev/null; echo kilo f32 done
bash tools/run_pmc.sh r3_kilo_x3 kilo kilo 65536 "--kilo-dtype bf16x3" > /dev/null; echo kilo x3 done
bash tools/run_pmc.sh r3_bf16 bf16 mlp_bf16 > /dev/null; echo bf16 done
bash tools/run_pmc.sh r3_bf16x3 bf16x3 mlp_bf16x3 > /dev/null; echo bf16x3 done
bash tools/profile_train.sh bf16x3 r3_train_x3 > /dev/null; echo train x3 done
bash tools/profile_train.sh fp16x2 r3_train_h2 > /dev/null; echo train h2 done
for dt in bf16x3 fp16x2 fp32; do bash tools/profile_bench.sh r3_kilo_${dt}_frame 20 3 --mode kilo --dtype $dt > /dev/null; echo kilo frame $dt done; done
