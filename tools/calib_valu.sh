#!/bin/bash
# tools/calib_valu.sh -- the pure v_fma_f32 loop of tools/ubench/valu_rates under rocprofv3 counters: what SQ_INSTS_VALU,
# SQ_ACTIVE_INST_VALU and SQ_BUSY_CYCLES read for a kernel whose VALU issue utilisation is 1 by construction (8 waves per
# SIMD, independent FMAs only).  Output: gpurun_out/calib_valu/{trace,pmc}/...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/calib_valu
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout -s KILL 60 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- "$ROOT/tools/ubench/valu_rates" calib > "$OUT/trace.log" 2>&1 || exit 1
timeout -s KILL 60 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc" -o pmc -- "$ROOT/tools/ubench/valu_rates" calib > "$OUT/pmc.log" 2>&1 || exit 1
timeout -s KILL 60 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc2" -o pmc2 -- "$ROOT/tools/ubench/valu_rates" calib > "$OUT/pmc2.log" 2>&1 || exit 1
cat "$OUT/trace.log" | grep calibration
