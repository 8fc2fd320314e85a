#!/bin/bash
# Clock and power of single kernels (GPU box): an A-B build (tools/build_variant.sh exp) repeats ONE idempotent kernel n times per round (OMOK_REPEAT_*), so that the
# rocm-smi samples taken meanwhile are that kernel's own.   tools/power_by_kernel.sh OUTDIR
out=${1:-gpurun_out/power}; mkdir -p $out
run() { # name, env assignment
  rm -f $out/$1.samples
  env $2 AB_BOARDS=15 tools/clock_sampler.sh $out/$1.samples -- python3 tools/ab_lib.py tools/ab/libomok_exp.so > $out/$1.ab 2>&1
  python3 - $out/$1.samples $1 "$(tail -1 $out/$1.ab)" <<'PY'
import re, sys
blocks, cur = [], None
for l in open(sys.argv[1]):
    if l.startswith('t '): cur = {'t': int(l.split()[1])}; blocks.append(cur)
    elif cur is not None:
        m = re.search(r'sclk.*\((\d+)Mhz\)', l)
        if m: cur['sclk'] = int(m.group(1))
        m = re.search(r'Power.*: ([0-9.]+)', l)
        if m: cur['W'] = float(m.group(1))
print(sys.argv[2], sys.argv[3])
print('   ', ' '.join(f"{b['t'] / 1000:.1f}:{b.get('sclk')}/{int(b.get('W', 0))}" for b in blocks))
PY
}
run plain OMOK_NOTHING=1
run children OMOK_REPEAT_CHILDREN=30
run window OMOK_REPEAT_WIN=40
run fc1 OMOK_REPEAT_FC1=200
run heads OMOK_REPEAT_HEADS=300
