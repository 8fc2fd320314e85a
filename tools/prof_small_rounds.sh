#!/bin/bash
# per-kernel stats of rounds with few live games (first 2 plies of engines with G games)
R=$PWD; out=gpurun_out/small; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for g in 64 256 1024 2048; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/ks_$g -- python3 $R/tools/play_plies.py 15 $g 800 16 2 > $R/$out/ks_$g.log 2>&1 || echo "pass $g failed"
  f=$(ls $R/$out/ks_$g/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/$out/g${g}_kernel_stats.csv
  rm -rf $R/$out/ks_$g
  tail -1 $R/$out/ks_$g.log
done
