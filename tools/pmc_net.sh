#!/bin/bash
# PMC passes over tools/bench_net.py (run on the GPU box from the repo root):  tools/pmc_net.sh <outdir> [batch]
# One rocprofv3 --pmc pass per counter group (never combined with trace domains); tools/pmc_summary.py prints per-kernel sums.
out=${1:-gpurun_out/pmc}; B=${2:-65536}
mkdir -p $out; R=$PWD; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" \
           "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$out/p$i -- python3 $R/tools/bench_net.py $B 1 > $R/$out/p$i.log 2>&1 || echo "pass $i failed"
done
cd $R; python3 tools/pmc_summary.py $out
