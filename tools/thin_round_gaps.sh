#!/bin/bash
# tools/thin_round_gaps.sh G...: rocprofv3 --kernel-trace of the first 3 plies of engines with few games; tools/thin_round_gaps.py reads the trace:
# per search round, the sum of the kernels' durations against the span from the first kernel's start to the last one's end (what a fused launch could close at most)
R=$PWD; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/thin
for g in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/thin/g$g -- python3 $R/tools/play_plies.py 15 $g 800 16 3 > $R/gpurun_out/thin/g$g.log 2>&1
  f=$(ls $R/gpurun_out/thin/g$g/*/*kernel_trace.csv | head -1)
  python3 $R/tools/thin_round_gaps.py $f $g
  rm -rf $R/gpurun_out/thin/g$g
done
