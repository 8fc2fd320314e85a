"""Per-ply averages of chosen kernels from a rocprofv3 kernel trace of ONE episode (tools/kernel_by_ply.sh): the k-th launch of k_round belongs to ply k // rounds_per_ply.
usage: python tools/kernel_by_ply.py TRACE.csv rounds_per_ply"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rpp = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = {"k_round<": "k_round", "k_sib_children2": "children2", "k_sib_children<": "children1", "k_fc0_mx<0, 0, true>": "win", "k_softmax_scatter": "scatter", "k_scan(": "scan",
         "k_gemm_t<16, 0": "fc1", "k_gemm_t<16, 2": "fc1_split", "k_fc0_x3<2, false>": "fc0_full", "k_trunk<15, false, 112>": "base", "k_trunk<15, false, 80>": "base_copy"}
ply, nround = -1, 0
acc = collections.defaultdict(lambda: collections.defaultdict(float))
t_first, t_last = {}, {}
for r in rows:
    k = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_round<" in k:
        if nround % rpp == 0:
            ply += 1
            t_first[ply] = int(r["Start_Timestamp"])
        nround += 1
    if ply < 0:
        continue
    t_last[ply] = int(r["End_Timestamp"])
    for key, nm in names.items():
        if key in k:
            acc[ply][nm] += dur
            break
    else:
        acc[ply]["other"] += dur
cols = ["k_round", "scatter", "scan", "base", "base_copy", "children2", "children1", "fc0_full", "win", "fc1", "fc1_split", "other"]
print("ply  wall_us_per_round  " + "  ".join(f"{c:>9s}" for c in cols) + "   (us per round, kernel time)")
for p in sorted(acc):
    wall = (t_last[p] - t_first[p]) / 1e3 / rpp
    print(f"{p:3d}  {wall:17.1f}  " + "  ".join(f"{acc[p][c] / rpp:9.1f}" for c in cols))
