"""Per-kernel sums of the rocprofv3 counter_collection.csv files under a directory (tools/collect_profiles.sh).

usage: python tools/pmc_summary.py <dir with pmc passes> [--json out.json --board 15 --rows R --sims S --rounds N]
Text: one line per (kernel, counter).  --json: the HBM bytes bench.py's `roofline*.traffic` uses, per net row / per simulation,
with the corrections MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE is tallied at 64 B per 128-B request: x2 for wide
coalesced reads; WRITE_SIZE exact for 16-B-per-lane stores), in bytes (the counters are KiB)."""
import argparse, collections, csv, glob, json

ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("--json")
ap.add_argument("--board", type=int, default=15)
ap.add_argument("--rows", type=float, default=0.0, help="net rows evaluated during the profiled command (bench.py nn evals)")
ap.add_argument("--sims", type=float, default=0.0, help="simulations run during the profiled command")
a = ap.parse_args()
agg = collections.defaultdict(float)
n = collections.defaultdict(int)
for f in sorted(glob.glob(f"{a.root}/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void omok::", "").replace("omok::", "")[:40]
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for (k, c), v in sorted(agg.items()):
    print(f"{k:42s} {c:32s} {v:14.5g}  ({n[(k, c)]} dispatch rows)")
if a.json:
    def tot(prefixes, counter):
        return sum(v for (k, c), v in agg.items() if c == counter and any(k.startswith(p) for p in prefixes))
    out = {"board": a.board, "rows": a.rows, "sims": a.sims, "unit": "bytes", "source": a.root,
           "method": "FETCH_SIZE [KiB] x 1024 x 2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE [KiB] x 1024; raw (x1) fetch kept beside it"}
    for name, pre in (("k_trunk", ["k_trunk", "k_sib_children", "k_group", "k_bin_prefix"]),
                      ("k_fc0_mx", ["k_fc0_mx", "k_fc0_x3", "k_splitk_finish", "k_facc_reduce", "k_win_finish"]),
                      ("tree", ["k_round", "k_scan", "k_fill", "k_scatter", "k_softmax_scatter", "k_add_evals"])):
        f, w = tot(pre, "FETCH_SIZE") * 1024.0, tot(pre, "WRITE_SIZE") * 1024.0
        # unit utilisation of the group over the profiled launches: MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x shader-engine-clock cycles);
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, a launch keeps 1024 SIMDs: busy / (1024 x GRBM / 8)
        grbm, mf = tot(pre, "GRBM_GUI_ACTIVE"), tot(pre, "SQ_VALU_MFMA_BUSY_CYCLES")
        if grbm > 0 and name != "tree":
            out[f"{name}_mfma_busy"] = mf / (1024.0 * grbm / 8.0)
            out[f"{name}_valu_busy"] = 4.0 * tot(pre, "SQ_INSTS_VALU") / (1024.0 * grbm / 8.0)
            out[f"{name}_lds_busy"] = tot(pre, "SQ_LDS_IDX_ACTIVE") / (256.0 * grbm / 8.0)
        denom = a.sims if name == "tree" else a.rows
        key = "tree_hbm_bytes_per_sim" if name == "tree" else f"{name}_hbm_bytes_per_row"
        if denom > 0 and (f > 0 or w > 0):
            out[key] = (2.0 * f + w) / denom
            out[key + "_fetch_raw_x1"] = (f + w) / denom
            out[key.replace("hbm_bytes", "fetch_bytes_x2")] = 2.0 * f / denom
            out[key.replace("hbm_bytes", "write_bytes")] = w / denom
            if name == "k_fc0_mx":  # a third of its sample stream (the fp6 residual part: 128 B per pixel) is fetched in exact 64-B requests,
                # which the counter tallies in full: x2 only applies to the rest (calibration of profiles/README.md, round 1)
                # (N = 15, difference path: the residual parts of a request are those of its 49-pixel difference row + 1/15 of a full row)
                resid = 128.0 * 49 + 128.0 * a.board * a.board / 15.0 if a.board == 15 else 128.0 * a.board * a.board
                out[key + "_calibrated"] = (2.0 * f + w) / denom - resid
    json.dump({str(a.board): out}, open(a.json, "w"), indent=1)
    print(json.dumps(out, indent=1))
