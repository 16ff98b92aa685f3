#!/bin/bash
# A/B of RTC_WF_SLOTS (cap on the path slots of a wavefront render; samples per slot follow) and RTC_WF_PARTS in one GPU run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "RTC_WF_PARTS=3" "RTC_WF_PARTS=4" "RTC_WF_PARTS=2" "RTC_WF_SLOTS=16777216" "RTC_WF_SLOTS=50331648" "RTC_WF_SLOTS=67108864" "RTC_WF_SLOTS=67108864 RTC_WF_PARTS=4" "RTC_WF_PARTS=3"; do echo "== $v"; env $v timeout -k 10 300 python3 "$ROOT/tools/bench_configs.py" "$@" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(f\"   {d['config']:18s} {d['kernel_ms']:9.3f} ms  {d['mrays_per_s']:9.1f} Mrays/s\")" || exit 1; done
