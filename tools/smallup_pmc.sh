#!/bin/bash
# SQ / LDS counters of the image-layer kernels (run ON the GPU box): PMC passes, kernel-trace only.  usage: tools/smallup_pmc.sh OUTDIR
OUT=${1:-gpurun_out/smallup_pmc}
export TMPDIR=/tmp; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
P3="SQ_WAVES SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --pmc $P --kernel-trace -d $OUT/p$i -o p -- python tools/smallup_probe.py > $OUT/p$i.log 2>&1
  python tools/pmc_agg.py $OUT/p$i/p_results.db | grep -E "n dur_us|small_up" > $OUT/p$i.txt
  i=$((i+1))
done
cat $OUT/p*.txt
