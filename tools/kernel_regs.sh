#!/bin/bash
# tools/kernel_regs.sh [pattern]: VGPRs, spills and scratch of the kernels in net_kernels.hip (device-only compile, code-object notes)
pat=${1:-k_sib_children}
R=$(cd "$(dirname "$0")/.." && pwd); T=/tmp/kregs; rm -rf $T; mkdir -p $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt --cuda-device-only -w $EXTRA -c $R/omok-ai_amd/csrc/net_kernels.hip -I$R/omok-ai_amd/csrc -o $T/dev.o || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/net.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/net.co | grep -E "\.name:|\.vgpr_count|vgpr_spill|\.private_segment_fixed" | python3 -c "
import sys
cur={}; rows=[]
for line in sys.stdin:
    k,_,v=line.strip().partition(':'); k=k.strip('- .'); v=v.strip()
    if k in cur: rows.append(cur); cur={}
    cur[k]=v
rows.append(cur)
for r in rows:
    if '$pat' in r.get('name',''): print(r['name'][:64], 'vgpr', r.get('vgpr_count'), 'spill', r.get('vgpr_spill_count'), 'scratch', r.get('private_segment_fixed_size'))
"
