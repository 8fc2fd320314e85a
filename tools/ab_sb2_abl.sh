#!/bin/bash
# (SB2_LINES / SB2_ABL are the switches of docs/experiments/r06_base_slot_whole_line_stores.diff: apply it before building the variants)
# timing-only: what the BASE pass's stores cost (SB2_ABL builds; mixed format forced so that the commit probe does not fall back on the wrong rows)
R=$PWD; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/sb2
for v in "$@"; do
  OMOK_MI355X_LIB=$R/tools/ab/libomok_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sb2/$v -- python3 $R/tools/play_plies.py 15 4096 800 16 2 5 > $R/gpurun_out/sb2/$v.log 2>&1
  f=$(ls $R/gpurun_out/sb2/$v/*/*kernel_stats.csv | head -1); echo "== $v"
  python3 - "$f" <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    nm=re.sub(r'\(.*','',r['Name']).replace('void omok::','')
    if any(k in nm for k in ('k_trunk<15, false, 112>','k_sib_children2<true, 15, false, false>','k_fc0_x3<2, false>')):
        print(f"  {nm[:48]:50s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
  rm -rf $R/gpurun_out/sb2/$v
done
