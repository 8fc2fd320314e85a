#!/bin/bash
# tools/kernel_by_ply.sh OUTDIR [bench.py args]: rocprofv3 --kernel-trace of one whole episode -> per-ply kernel averages (tools/kernel_by_ply.py)
out=$1; shift; R=$PWD; mkdir -p $R/$out
cd /tmp; export TMPDIR=/tmp OMOK_BENCH_CLOCKS=0
rocprofv3 --kernel-trace --output-format csv -d $R/$out/kt -- python3 $R/bench.py --steps 1 --warmup 0 --f16-leg 0 --cpu-seconds 0 --precision-rows 0 --train-steps 0 --slots-multiple 0 --window-plies 0 "$@" > $R/$out/kt.log 2>&1 || echo "trace failed"
cd $R
f=$(ls $out/kt/*/*kernel_trace.csv | head -1)
python3 tools/kernel_by_ply.py $f 50 > $out/kernel_by_ply.txt
rm -rf $out/kt
