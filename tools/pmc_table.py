"""tools/pmc_table.py <prof dir> <kernel substring> -- average per launch of every counter of the matching kernel, over the pmc_* passes."""
import csv, glob, os, sys
d, key = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "*_counter_collection.csv"))):
    acc = {}
    for row in csv.DictReader(open(f)):
        if key in row["Kernel_Name"]:
            acc.setdefault((row["Kernel_Name"][:60], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:60s} {c:28s} {sum(v) / len(v):16.1f}  x{len(v)}")
for f in glob.glob(os.path.join(d, "trace", "*kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if key in row["Name"]:
            print(row["Name"][:60], "calls", row["Calls"], "avg ns", row["AverageNs"])
