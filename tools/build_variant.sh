#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG=V ...]: an A-B build of the library with extra compile flags for net_kernels.hip -> tools/ab/libomok_NAME.so
# (tree_kernels.o and engine.o are those of the regular build; run `make -C omok-ai_amd/csrc` first).  Timed by tools/ab_lib.py.
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/omok-ai_amd/csrc; mkdir -p $R/tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -fhip-fp32-correctly-rounded-divide-sqrt -DOMOK_EXPERIMENT "$@" -c $C/net_kernels.hip -o /tmp/nk_$name.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libomok_$name.so $C/tree_kernels.o /tmp/nk_$name.o $C/engine.o && echo built tools/ab/libomok_$name.so
