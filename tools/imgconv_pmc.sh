#!/bin/bash
# SQ / LDS counters of the convolution kernels (run ON the GPU box): two PMC passes per engine, kernel-trace only.
# usage: tools/imgconv_pmc.sh OUTDIR [new|old]
OUT=${1:-gpurun_out/imgconv_pmc}; W=${2:-new}
export TMPDIR=/tmp; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
i=1
for P in "$P1" "$P2"; do
  rocprofv3 --pmc $P --kernel-trace -d $OUT/${W}_p$i -o p -- python tools/imgconv_probe.py prof $W > $OUT/${W}_p$i.log 2>&1
  python tools/pmc_agg.py $OUT/${W}_p$i/p_results.db | grep -E "n dur_us|imgconv|imgwgrad|splitk|igemm_bf" > $OUT/${W}_p$i.txt
  i=$((i+1))
done
cat $OUT/${W}_p*.txt
