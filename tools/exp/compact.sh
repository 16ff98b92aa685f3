#!/bin/bash
# A/B of RTC_WF_COMPACT_EVERY (rounds between two compaction points of a wavefront render) in one GPU run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in 8 2 4 6 12 16 8; do echo "== RTC_WF_COMPACT_EVERY=$v"; RTC_WF_COMPACT_EVERY=$v timeout -k 10 300 python3 "$ROOT/tools/bench_configs.py" "$@" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(f\"   {d['config']:18s} {d['kernel_ms']:9.3f} ms  {d['mrays_per_s']:9.1f} Mrays/s\")" || exit 1; done
