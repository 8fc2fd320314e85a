#!/bin/bash
# (SB2_LINES / SB2_ABL are the switches of docs/experiments/r06_base_slot_whole_line_stores.diff: apply it before building the variants)
# A-B of the BASE pass's base-slot stores (SB2_LINES: whole lines through the halo grid / 0: lane-per-pixel pieces): parity tests of the product build first, then
# three interleaved pairs of the first 3 plies of configs[1] (tools/ab_lib.py), then the kernel's own time by rocprofv3 for both builds
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_headline_path.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for i in 1 2 3; do AB_BOARDS=15 python tools/ab_lib.py tools/ab/libomok_lines0.so tools/ab/libomok_lines1.so; done
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in lines0 lines1; do
  OMOK_MI355X_LIB=$R/tools/ab/libomok_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sb2/$v -- python3 $R/tools/play_plies.py 15 4096 800 16 3 > $R/gpurun_out/sb2/$v.log 2>&1
  f=$(ls $R/gpurun_out/sb2/$v/*/*kernel_stats.csv | head -1); echo "== $v"; grep -E 'k_trunk<15, false, 112>|k_sib_children2<true, 15|k_fc0_x3<2, false>' $f | cut -d, -f1-4 | sed 's/(.*)//' | cut -c1-120
done
