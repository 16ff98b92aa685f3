#!/bin/bash
# A/B of RTC_WF_PARTS (parts a wavefront render is split into) in one GPU run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for pass in 1 2; do for n in 1 2 3 4; do echo "== parts $n (pass $pass)"; RTC_WF_PARTS=$n timeout -k 10 300 python3 "$ROOT/tools/bench_configs.py" "$@" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(f\"   {d['config']:18s} {d['kernel_ms']:9.3f} ms  {d['mrays_per_s']:9.1f} Mrays/s\")" || exit 1; done; done
