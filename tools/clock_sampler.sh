#!/bin/bash
# Samples the GPU's clocks, power and temperature while a command runs (GPU box; rocm-smi reads need no privileges).
#   tools/clock_sampler.sh OUT.txt -- command...      a block of rocm-smi lines per ~0.5 s, each block headed by "t <milliseconds since start>"
out=$1; shift; shift
( t0=$(date +%s%N)
  while true; do
    echo "t $(( ($(date +%s%N) - t0) / 1000000 ))" >> $out
    rocm-smi -d 0 --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature" >> $out
    sleep 0.3
  done ) &
sp=$!
"$@"
rc=$?
kill $sp 2>/dev/null
exit $rc
