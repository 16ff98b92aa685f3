#!/bin/bash
# tools/profile.sh <tag> -- rocprofv3 passes over bench.py on the GPU box (run through gpurun from the repo root).
# One pass = one process: kernel trace + stats, then PMC passes within the per-block slot limits
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ, 4 TCC with FETCH_SIZE=3 / WRITE_SIZE=2, 2 GRBM).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-other"
run() { # name, rocprof args...
    local name=$1; shift
    timeout -s KILL 180 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $BENCH > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc"
    return $rc
}
run trace --kernel-trace --stats || exit 1
run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU || exit 1
run pmc_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM || exit 1
run pmc_tcc1 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
run pmc_tcc2 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
find "$OUT" -name "*.csv" | head -40
