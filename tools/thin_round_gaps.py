"""tools/thin_round_gaps.py TRACE.csv G: launch gaps of small search rounds from a rocprofv3 kernel trace (see thin_round_gaps.sh).
A round = the kernels from one k_round launch to the next; only the rounds of the last two plies are counted (the first ply warms up)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void omok::", "").replace("omok::", "")) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_round<")]
rounds = [ev[a:b] for a, b in zip(starts, starts[1:])]
rounds = rounds[-100:]  # the last two plies
span = sum(r[-1][1] - r[0][0] for r in rounds) / len(rounds) / 1e3
busy = sum(sum(e[1] - e[0] for e in r) for r in rounds) / len(rounds) / 1e3
period = (rounds[-1][0][0] - rounds[0][0][0]) / (len(rounds) - 1) / 1e3
n = sum(len(r) for r in rounds) / len(rounds)
print(f"G={sys.argv[2]}: {len(rounds)} rounds, {n:.1f} kernels per round; round period {period:.1f} us, kernels' durations {busy:.1f} us, gaps {period - busy:.1f} us ({100 * (period - busy) / period:.0f} %)")
by = {}
for r in rounds:
    for e in r:
        by[e[2]] = by.get(e[2], 0) + (e[1] - e[0])
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]:
    print(f"    {k[:50]:52s} {v / len(rounds) / 1e3:7.1f} us per round")
