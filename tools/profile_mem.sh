#!/bin/bash
# tools/profile_mem.sh <tag> <python script> [args...] -- the vector-memory path of a kernel: TA / TCP / TD counter passes
# (address unit busy and stalls, L1 latency and gate cycles, translation misses), within the per-block slot limits
# (2 TA, 4 TCP, 2 TD per pass).  Companion of tools/profile_cmd.sh; run through gpurun from the repo root.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SCRIPT=$ROOT/$1; shift
cd /tmp
ARGS="$*"
run() { # name, rocprof args...
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$SCRIPT" $ARGS > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc"
    return $rc
}
run pmc_mem1 --kernel-trace --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE || exit 1
run pmc_mem2 --kernel-trace --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TD_LOAD_WAVEFRONT_sum || exit 1
