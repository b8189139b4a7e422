#!/bin/bash
# SQ-only PMC passes (pmc1 + pmc2 of tools/run_pmc.sh) for quick kernel iterations.
#   bash tools/run_pmc_sq.sh <tag> <only> <kernel-name-pattern> [rays]
set -e
TAG=$1; ONLY=$2; PAT=$3; RAYS=${4:-65536}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
[ -z "$ROOT" ] && ROOT=/root/repo
run() {
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 $ROOT/tools/profile_kernels.py --only $ONLY --rays $RAYS --reps 2 > $OUT/$name.log 2>&1
}
run pmc1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run pmc2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
cd $ROOT
python3 tools/pmc_summary.py "$PAT" $(find $OUT -name '*counter_collection.csv' | sort) > $OUT/pmc_summary.json
python3 - <<PY
import json
d=json.load(open("$OUT/pmc_summary.json"))
by={}
for e in d: by.setdefault(e['pass'],[]).append(e)
for p,es in sorted(by.items()):
    e=sorted(es,key=lambda e:-e['ns'])[0]
    print(p, {k:(v if not isinstance(v,float) else round(v/1e6,2)) for k,v in e.items() if k not in('kernel','pass','dispatch','grid','wg','lds','sgpr')})
PY
