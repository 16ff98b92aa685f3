#!/bin/bash
# A/B of RTC_WF_COMPACT_EVERY x RTC_WF_COMPACT_DIV in one GPU run, two passes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for pass in 1 2; do for v in ${SWEEP:-"8 4" "4 4" "3 4" "2 4" "1 4" "2 8" "4 8" "2 16" "1 8"}; do set -- ${v/:/ }; echo "== every $1 div $2 (pass $pass)"; RTC_WF_COMPACT_EVERY=$1 RTC_WF_COMPACT_DIV=$2 timeout -k 10 300 python3 "$ROOT/tools/bench_configs.py" final_1024 million_1080p book1_1080p final_earth_1080p | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(f\"   {d['config']:18s} {d['kernel_ms']:9.3f} ms\")" || exit 1; done; done
