#!/bin/bash
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_tail_gemm.py -x -q -m gpu 2>&1 | tail -15 || exit 1
for i in 1 2 3; do
  for w in 0 1; do echo "OMOK_GEMM_W=$w"; OMOK_GEMM_W=$w AB_BOARDS=15 python tools/ab_lib.py omok-ai_amd/libomok_mi355x.so; done
done
