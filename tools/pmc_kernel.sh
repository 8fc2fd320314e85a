#!/bin/bash
# tools/pmc_kernel.sh OUTDIR NAME "COUNTERS" [play_plies.py args]: one rocprofv3 --pmc pass (with --kernel-trace only) over tools/play_plies.py,
# per-kernel sums -> OUTDIR/NAME.txt.  OMOK_MI355X_LIB in the environment selects the library build.
out=$1; name=$2; ctr=$3; shift 3; R=$PWD; mkdir -p $R/$out
cd /tmp; export TMPDIR=/tmp OMOK_BENCH_CLOCKS=0 # (bench.py starts no child process under the profiler)
timeout -k 10 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/$out/pmc_$name -- python3 $R/tools/play_plies.py "$@" > $R/$out/pmc_$name.log 2>&1 || echo "pmc pass $name failed"
cd $R
python3 tools/pmc_summary.py $out/pmc_$name > $out/$name.txt; grep -h "^{" $out/pmc_$name.log >> $out/$name.txt
rm -rf $out/pmc_$name
