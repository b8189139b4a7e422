set -e
cd $GRAFT_REPO_ROOT
for dt in bf16x3 bf16 fp16x2 fp32; do bash tools/run_bench_traffic.sh r3_traffic_$dt $dt > /dev/null; echo traffic $dt done; done
bash tools/run_pmc.sh r3_kilo_h2 kilo kilo > /dev/null; echo kilo h2 done
bash tools/run_pmc.sh r3_kilo_f32 kilo kilo 65536 "--kilo-dtype fp32" > /dev/null; echo kilo f32 done
bash tools/run_pmc.sh r3_bf16 bf16 mlp_bf16 > /dev/null; echo bf16 done
bash tools/run_pmc.sh r3_bf16x3 bf16x3 mlp_bf16x3 > /dev/null; echo bf16x3 done
