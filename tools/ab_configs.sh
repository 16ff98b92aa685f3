#!/bin/bash
# tools/ab_configs.sh <config>... -- A/B timing of library builds in ONE GPU run (boxes differ by a few per cent): every raytracer_amd/librtc*.so
# named in $LIBS (default: all of them) renders the named tools/bench_configs.py configurations, twice, interleaved.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=${LIBS:-$(cd "$ROOT/raytracer_amd" && ls librtc*.so)}
for pass in 1 2; do
  for L in $LIBS; do
    echo "== $L (pass $pass)"
    RTC_LIB=$ROOT/raytracer_amd/$L timeout -k 10 300 python3 "$ROOT/tools/bench_configs.py" "$@" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(f\"   {d['config']:18s} {d['kernel_ms']:9.3f} ms  {d['mrays_per_s']:9.1f} Mrays/s\")" || exit 1
  done
done
