#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for L in ${LIBS:-librtc.so librtc_w7.so librtc_w6.so librtc_w5.so}; do
  for C in final_1024 million_1080p; do
    OUT=/tmp/tr_${L}_${C}
    RTC_LIB=$ROOT/raytracer_amd/$L timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $ROOT/tools/bench_configs.py $C --reps 2 > $OUT.log 2>&1 || { echo fail $L $C; tail -3 $OUT.log; exit 1; }
    echo "== $L $C"
    python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_wf_" in r["Name"] and float(r["Percentage"]) > 1.0:
        import re
        print(f"   {re.search(r'k_wf_[a-z_]+(<[^>]*>)?', r['Name']).group(0):32s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
  done
done
