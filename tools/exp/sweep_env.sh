#!/bin/bash
# tools/exp/sweep_env.sh <ENV_NAME> "<v1> <v2> ..." <config>... -- A/B of one environment knob of the library over tools/bench_configs.py
# configurations inside ONE GPU run, two passes (kernel ms per configuration)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; VALUES=$2; shift 2
for pass in 1 2; do for v in $VALUES; do echo "== $NAME=$v (pass $pass)"; env $NAME=$v timeout -k 10 300 python3 "$ROOT/tools/bench_configs.py" "$@" 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-18s %9.3f ms' % (d['config'], d['kernel_ms']))" || exit 1; done; done
