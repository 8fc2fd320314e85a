#!/bin/bash
# tools/ab_episode.sh LIB [LIB ...]: whole configs[1] episodes (bench.py --steps 2, no extra legs) per library build, in the order given -> games/s and category times
for lib in "$@"; do
  OMOK_MI355X_LIB=$PWD/$lib OMOK_BENCH_CLOCKS=0 python3 bench.py --steps 2 --warmup 1 --f16-leg 0 --cpu-seconds 0 --precision-rows 0 --train-steps 0 --slots-multiple 0 --window-plies 0 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['rank0_kernel_ms']
print('$lib', round(d['value'], 1), 'games/s', {x: round(k[x]) for x in k})"
done
