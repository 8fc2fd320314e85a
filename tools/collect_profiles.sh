#!/bin/bash
# Collects the round's judged evidence on the GPU box (run from the repo root):
#   tools/collect_profiles.sh gpurun_out/r02
# 1. bench.py line (full C2 episode, 1 warm-up + 1 timed step)  -> bench_c2_full_episode.json
# 2. rocprofv3 --kernel-trace --stats (first 4 plies of C2)       -> kernel_stats_c2_first4plies.csv
# 3. rocprofv3 --pmc passes over the SAME self-play workload (first 2 plies: 100 rounds of 65536 rows, tree kernels included),
#    one counter group per pass, --kernel-trace only (no other trace domain)  -> pmc_selfplay_c2_first2plies.txt, pmc_bytes.json
out=${1:-gpurun_out/r02}; mkdir -p $out; R=$PWD
echo "[1] bench"; python3 bench.py --steps 1 --warmup 1 > $out/bench_c2_full_episode.json 2> $out/bench.err || exit 1
tail -c 300 $out/bench_c2_full_episode.json; echo
cd /tmp; export TMPDIR=/tmp
echo "[2] kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/kstats -- python3 $R/bench.py --max-plies 4 --cpu-seconds 0 --precision-rows 0 --train-steps 0 > $R/$out/kstats.log 2>&1 || echo "kernel-trace pass failed"
f=$(ls $R/$out/kstats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_c2_first4plies.csv
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); echo "[3] pmc pass $i: $grp"
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$out/pmc/p$i -- python3 $R/bench.py --max-plies 2 --cpu-seconds 0 --precision-rows 0 --train-steps 0 > $R/$out/pmc_p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
# rows / sims of the profiled command from its own result line (the same in every pass: deterministic workload)
read rows sims < <(python3 - <<PY
import json
for l in open("$out/pmc_p5.log"):
    if l.startswith("{"):
        o = json.loads(l); t = o["ms_per_step"] / 1e3 * o["steps"]
        print(o["nn_evals_per_s"] * t, o["mcts_sims_per_s"] * t); break
PY
)
python3 tools/pmc_summary.py $out/pmc --json $out/pmc_bytes.json --board 15 --rows ${rows:-0} --sims ${sims:-0} > $out/pmc_selfplay_c2_first2plies.txt
tail -30 $out/pmc_selfplay_c2_first2plies.txt
rm -rf $out/pmc/*/*/*.db 2>/dev/null
