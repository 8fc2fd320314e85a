#!/bin/bash
# tools/kstats.sh OUTDIR NAME [bench.py args...]: rocprofv3 --kernel-trace --stats of one bench.py invocation -> OUTDIR/NAME_kernel_stats.csv
out=$1; name=$2; shift 2; R=$PWD; mkdir -p $R/$out
cd /tmp; export TMPDIR=/tmp OMOK_BENCH_CLOCKS=0 # (bench.py starts no child process under the profiler)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/ks_$name -- python3 $R/bench.py --cpu-seconds 0 --precision-rows 0 --train-steps 0 --slots-multiple 0 --window-plies 0 "$@" > $R/$out/ks_$name.log 2>&1 || echo "kernel-trace pass failed"
f=$(ls $R/$out/ks_$name/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $R/$out/${name}_kernel_stats.csv
rm -rf $R/$out/ks_$name
cd $R
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/${name}_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("== $name: total kernel ms", tot/1e6)
for r in rows[:24]:
    print(f"{r['Name'][:64]:66s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):6.2f}%")
PY
