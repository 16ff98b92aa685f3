#!/bin/bash
# A/B of RTC_BVH_LEAF (records per BVH leaf) in one GPU run, two passes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for pass in 1 2; do for v in 2 1 3 4; do echo "== RTC_BVH_LEAF=$v (pass $pass)"; RTC_BVH_LEAF=$v timeout -k 10 300 python3 "$ROOT/tools/bench_configs.py" final_1024 million_1080p book1_1080p final_earth_1080p | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(f\"   {d['config']:18s} {d['kernel_ms']:9.3f} ms  nodes {d['bvh']['nodes']}\")" || exit 1; done; done
