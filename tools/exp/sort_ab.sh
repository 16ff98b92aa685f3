#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for pass in 1 2; do for v in "0 2" "1 2" "1 1" "0 1"; do set -- $v; echo "== sort $1 every $2 (pass $pass)"; RTC_WF_SORT=$1 RTC_WF_COMPACT_EVERY=$2 timeout -k 10 300 python3 "$ROOT/tools/bench_configs.py" final_1024 million_1080p book1_1080p 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-18s %9.3f ms' % (d['config'], d['kernel_ms']))" || exit 1; done; done
