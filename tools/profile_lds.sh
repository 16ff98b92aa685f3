#!/bin/bash
# tools/profile_lds.sh <tag> <python script> [args...] -- kernel trace + one SQ pass + one LDS pass (bank conflicts, LDS busy) over a python
# command; for the LDS-table extend experiments.  Run through gpurun from the repo root.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift
cd /tmp
run() {
    local name=$1; shift
    timeout -s KILL 120 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$SCRIPT" $ARGS > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc"
    return $rc
}
ARGS="$*"
run trace --kernel-trace --stats || exit 1
run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY || exit 1
run pmc_lds --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE || exit 1
