#!/bin/bash
# tools/exp/ab_libs.sh <old.so> <new.so> <config>... -- two library builds in one GPU run: wavefront accumulators bit for bit (tools/ab_wf.py), then timings
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
OLD=$1; NEW=$2; shift 2
RTC_LIB=$ROOT/raytracer_amd/$OLD timeout -k 10 120 python3 tools/ab_wf.py old || exit 1
RTC_LIB=$ROOT/raytracer_amd/$NEW timeout -k 10 120 python3 tools/ab_wf.py new || exit 1
python3 - <<'PY'
import numpy as np
for n in ("final", "book1", "box_field_smoke"):
    a = np.load(f"gpurun_out/abwf_old_{n}.npy"); b = np.load(f"gpurun_out/abwf_new_{n}.npy")
    print(n, "bit-identical" if np.array_equal(a, b) else f"DIFFERENT: max abs {np.abs(a - b).max():.3g}, pixels {int((a != b).any(-1).sum())}", flush=True)
PY
LIBS="$OLD $NEW" bash tools/ab_configs.sh "$@"
