#!/bin/bash
# tools/profile_cmd.sh <tag> <python script> [args...] -- rocprofv3 kernel trace + PMC passes over one python command
# (run through gpurun from the repo root).  Same pass layout as tools/profile.sh; the program after `--` is python itself.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift
cd /tmp
run() { # name, rocprof args...
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$SCRIPT" $ARGS > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc"
    return $rc
}
ARGS="$*"
run trace --kernel-trace --stats || exit 1
run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU || exit 1
run pmc_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM || exit 1
run pmc_tcc1 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run pmc_tcc2 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
run pmc_tcp --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum || true
