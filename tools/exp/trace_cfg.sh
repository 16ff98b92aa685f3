#!/bin/bash
# tools/exp/trace_cfg.sh <tag> <config>... -- rocprofv3 kernel trace + stats of tools/bench_configs.py for the named configurations
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$ROOT/tools/bench_configs.py" "$@" --reps 2 > "$OUT/log.txt" 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
