#!/bin/bash
# tools/exp/trace_timeline.sh <tag> <config> -- rocprofv3 kernel trace of ONE configuration of tools/bench_configs.py, then the launch-duration
# histogram of the extend kernels and the busy / idle account of the GPU over the render (development aid)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/timeline_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python3 "$ROOT/tools/bench_configs.py" "$@" --reps 2 > "$OUT/log.txt" 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys
import collections
f = glob.glob(sys.argv[1] + "/**/t_kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
# the last render = from the last k_wf_init group backwards: take the final third of the trace's wf kernels
wf = [r for r in rows if "k_wf_" in r[0]]
fin = [i for i, r in enumerate(wf) if "k_wf_finish" in r[0]]
n_parts = int(__import__("os").environ.get("TL_PARTS", "3"))
start_i = fin[-n_parts - 1] + 1 if len(fin) > n_parts else 0
last = wf[start_i:]
t0, t1 = min(r[1] for r in last), max(r[2] for r in last)
print(f"last render: {len(last)} kernels, wall {(t1 - t0) / 1e6:.2f} ms")
by = collections.defaultdict(list)
import re
for n, s, e in last: by[re.search(r"k_wf_\w+(<[^>]*>)?", n).group(0)].append((e - s) / 1e3)
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    d.sort()
    print(f"  {n:60s} n {len(d):4d} sum {sum(d) / 1e3:8.2f} ms  min {d[0]:8.1f} med {d[len(d) // 2]:8.1f} max {d[-1]:8.1f} us  <50us: {sum(x < 50 for x in d)}  <100us: {sum(x < 100 for x in d)}  sum of those {sum(x for x in d if x < 100) / 1e3:.2f} ms")
# busy account: union of intervals
ev = sorted((s, e) for _, s, e in last)
busy = 0; cur_s, cur_e = ev[0]
for s, e in ev[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"  some kernel running {busy / 1e6:.2f} ms of {(t1 - t0) / 1e6:.2f} ms; idle gaps {(t1 - t0 - busy) / 1e6:.2f} ms")
# concurrency over time: time-weighted number of kernels in flight
pts = sorted([(s, 1) for _, s, e in last] + [(e, -1) for _, s, e in last])
lvl = 0; prev = pts[0][0]; hist = collections.Counter()
for t, d in pts:
    hist[lvl] += t - prev; prev = t; lvl += d
print("  time with k kernels in flight:", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
PY
