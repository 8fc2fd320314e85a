"""Per-kernel sums of the rocprofv3 counter_collection.csv files under a directory (tools/pmc_net.sh)."""
import collections, csv, glob, sys
root = sys.argv[1]
agg = collections.defaultdict(float)
n = collections.defaultdict(int)
for f in sorted(glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void omok::", "")[:36]
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for (k, c), v in sorted(agg.items()):
    if any(t in k for t in ("fc0", "trunk", "gemm")):
        print(f"{k:38s} {c:32s} {v:14.5g}  ({n[(k, c)]} dispatch rows)")
