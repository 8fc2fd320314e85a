#!/bin/bash
# Collects the round's judged evidence on the GPU box (run from the repo root):
#   tools/collect_profiles.sh gpurun_out/r01
# 1. bench.py line (full C2 episode)            -> bench_c2_full_episode.json
# 2. rocprofv3 --kernel-trace --stats (4 plies)  -> kernel_stats_c2_first4plies.csv
# 3. rocprofv3 --pmc passes over bench_net.py    -> pmc_net_b65536.txt   (one pass per counter group, no trace domains)
out=${1:-gpurun_out/r01}; mkdir -p $out; R=$PWD
echo "[1] bench"; python3 bench.py > $out/bench_c2_full_episode.json 2> $out/bench.err || exit 1
tail -c 400 $out/bench_c2_full_episode.json; echo
cd /tmp; export TMPDIR=/tmp
echo "[2] kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/kstats -- python3 $R/bench.py --max-plies 4 --cpu-seconds 0 > $R/$out/kstats.log 2>&1 || echo "kernel-trace pass failed"
f=$(ls $R/$out/kstats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/$out/kernel_stats_c2_first4plies.csv
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); echo "[3] pmc pass $i: $grp"
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$out/pmc/p$i -- python3 $R/tools/bench_net.py 65536 1 > $R/$out/pmc_p$i.log 2>&1 || echo "pass $i failed"
done
cd $R; python3 tools/pmc_summary.py $out/pmc > $out/pmc_net_b65536.txt; cat $out/pmc_net_b65536.txt | head -70
