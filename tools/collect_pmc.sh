#!/bin/bash
# tools/collect_pmc.sh OUTDIR BOARD [bench.py args...]: the five rocprofv3 --pmc passes of tools/collect_profiles.sh (one counter group per pass,
# --kernel-trace only) over `bench.py --max-plies 2 <args>`, summarised per kernel, and the per-row / per-simulation HBM bytes merged into
# OUTDIR/pmc_bytes.json under the board's key.
out=$1; board=$2; shift 2; R=$PWD; mkdir -p $R/$out
cd /tmp; export TMPDIR=/tmp OMOK_BENCH_CLOCKS=0 # (bench.py starts no child process under the profiler)
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); echo "pmc pass $i: $grp"
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$out/pmc$board/p$i -- python3 $R/bench.py --max-plies 2 --cpu-seconds 0 --precision-rows 0 --train-steps 0 --slots-multiple 0 --window-plies 0 "$@" > $R/$out/pmc${board}_p$i.log 2>&1 || echo "pass $i failed"
done
cd $R
read rows sims < <(python3 - <<PY
import json
for l in open("$out/pmc${board}_p5.log"):
    if l.startswith("{"):
        o = json.loads(l); t = o["ms_per_step"] / 1e3 * o["steps"]
        print(o["nn_evals_per_s"] * t, o["mcts_sims_per_s"] * t); break
PY
)
python3 tools/pmc_summary.py $out/pmc$board --json $out/pmc_bytes_$board.json --board $board --rows ${rows:-0} --sims ${sims:-0} > $out/pmc_selfplay_board${board}_first2plies.txt
tail -12 $out/pmc_selfplay_board${board}_first2plies.txt
rm -rf $out/pmc$board/*/*/*.db 2>/dev/null
