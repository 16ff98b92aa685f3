#!/bin/bash
# tools/exp/primary_ab.sh -- camera rays: through the slot lists (RTC_WF_PRIMARY=0), traced ahead per lane (RTC_WF_PRIMARY_WAVE=0), traced
# ahead with one traversal per wave (default): accumulators bit for bit, then timings, all in one GPU run
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
RTC_WF_PRIMARY=0 timeout -k 10 120 python3 tools/ab_wf.py p0 || exit 1
RTC_WF_PRIMARY_WAVE=0 timeout -k 10 120 python3 tools/ab_wf.py p1 || exit 1
timeout -k 10 120 python3 tools/ab_wf.py p2 || exit 1
python3 - <<'PY' || exit 1
import numpy as np, sys
bad = 0
for n in ("final", "book1", "box_field_smoke"):
    a = np.load(f"gpurun_out/abwf_p0_{n}.npy")
    for tag in ("p1", "p2"):
        b = np.load(f"gpurun_out/abwf_{tag}_{n}.npy")
        same = np.array_equal(a, b)
        print(n, tag, "bit-identical" if same else f"DIFFERENT: max abs {np.abs(a - b).max():.3g}, pixels {int((a != b).any(-1).sum())}", flush=True)
        bad += not same
sys.exit(1 if bad else 0)
PY
for pass in 1 2; do
  for v in "RTC_WF_PRIMARY=0" "RTC_WF_PRIMARY_WAVE=0" "RTC_WF_PRIMARY_WAVE=1"; do
    echo "== $v (pass $pass)"
    env $v timeout -k 10 300 python3 tools/bench_configs.py final_1024 final_earth_1080p book1_1080p book1_1080p_8spp million_1080p | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(f\"   {d['config']:18s} {d['kernel_ms']:9.3f} ms  {d['mrays_per_s']:9.1f} Mrays/s  rays {d['rays_per_sample']}\")" || exit 1
  done
done
