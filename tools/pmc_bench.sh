#!/bin/bash
# PMC passes over bench.py (same command as the bench line; one counter group per run as the
# skills guide prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass).  The summary is keyed by
# workload AND matrix order, which is how bench.py looks it up (profiles/rNN_pmc_bench_<wl>_n<order>.json).
# usage (on the GPU box): tools/pmc_bench.sh <workload c3|c4|c5> <matrix order> [bench args...]
set -u
WL=$1; ORDER=$2; shift; shift
TAG=${WL}_n${ORDER}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcb_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-c4 --no-c1 "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
done
python3 $GRAFT_REPO_ROOT/tools/pmc_to_json.py $OUT $GRAFT_REPO_ROOT/gpurun_out/pmc_bench_$TAG.json
rm -rf $OUT
