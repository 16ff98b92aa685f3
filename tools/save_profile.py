"""Copy what is to be judged from a tools/profile.sh output directory into profiles/<tag>/ (tracked): the kernel stats of the
trace pass, one condensed CSV per PMC pass (kernel, counter, launches, mean per launch) and prof_summary.py's JSON.
    python tools/save_profile.py gpurun_out/prof_r02a profiles/r02a [--latest]"""
import collections, csv, glob, json, os, shutil, subprocess, sys
src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(dst, os.path.basename(d) + ".csv"), "w") as out:
        out.write("kernel,counter,launches,mean_per_launch\n")
        for (k, c), v in sorted(agg.items()):
            out.write(f'"{k}",{c},{len(v)},{sum(v) / len(v)}\n')
here = os.path.dirname(os.path.abspath(__file__))
summary = subprocess.run([sys.executable, os.path.join(here, "prof_summary.py"), src] + [a for a in sys.argv[3:] if not a.startswith("--")], capture_output=True, text=True, check=True).stdout
s = json.loads(summary)
s["tag"] = os.path.basename(dst.rstrip("/"))
json.dump(s, open(os.path.join(dst, "summary.json"), "w"), indent=1)
if "--latest" in sys.argv:
    json.dump(s, open(os.path.join(os.path.dirname(dst.rstrip("/")), "latest.json"), "w"), indent=1)
print(json.dumps({k: s[k] for k in ("kernel", "avg_ms", "active_lane_fraction") if k in s}))
