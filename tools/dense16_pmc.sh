#!/bin/bash
# SQ / LDS counters of the dense16 kernels (run ON the GPU box): two PMC passes, kernel-trace only.
OUT=${1:-gpurun_out/d16_pmc}
export TMPDIR=/tmp; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
i=1
for P in "$P1" "$P2"; do
  rocprofv3 --pmc $P --kernel-trace -d $OUT/p$i -o p -- python tools/dense16_probe.py > $OUT/p$i.log 2>&1
  python tools/pmc_agg.py $OUT/p$i/p_results.db d16 > $OUT/p$i.txt
  rm -rf $OUT/p$i
  i=$((i+1))
done
cat $OUT/p*.txt
