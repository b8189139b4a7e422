#!/bin/bash
# rocprofv3 evidence for one workload of tools/profile_kernels.py: kernel-trace stats + separate PMC passes
# (SQ issue / wait counters, LDS, FETCH_SIZE, WRITE_SIZE), summarised into gpurun_out/<tag>/.
#   bash tools/run_pmc.sh <tag> <only> <kernel-name-pattern> [rays] [extra profile_kernels.py flags, e.g. "--kilo-dtype fp32"]
set -e
TAG=$1; ONLY=$2; PAT=$3; RAYS=${4:-65536}; EXTRA=${5:-}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
[ -z "$ROOT" ] && ROOT=/root/repo
run() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 $ROOT/tools/profile_kernels.py --only $ONLY --rays $RAYS --reps 2 $EXTRA > $OUT/$name.log 2>&1
}
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $ROOT/tools/profile_kernels.py --only $ONLY --rays $RAYS --reps 2 $EXTRA > $OUT/trace.log 2>&1
run pmc1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run pmc2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU
run pmc3 FETCH_SIZE
run pmc4 WRITE_SIZE
run pmc5 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $ROOT
python3 tools/pmc_summary.py "$PAT" $(find $OUT -name '*counter_collection.csv' | sort) > $OUT/pmc_summary.json
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
echo "summary in $OUT"
