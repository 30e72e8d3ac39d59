#!/bin/bash
# Runs ON the GPU box (via gpurun): every measurement behind profiles/r06_* in one call.
#   bash tools/gpu_profile_r06.sh [TAG]      -> gpurun_out/<TAG>/...
TAG=${1:-r06}; OUT=gpurun_out/$TAG
export TMPDIR=/tmp; mkdir -p $OUT
cp .commit_stamp $OUT/commit.txt 2>/dev/null  # written by the caller before gpurun (.git does not travel)
# 1. kernel trace of the default bench command (the JSON line of the same run beside it)
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --steps 20 --warmup 5 > $OUT/bench_under_rocprof.log 2>&1
# 2. the unprofiled bench lines: headline + the other BASELINE configurations
for c in cfg3 cfg3k1 cfg2 cfg5 cfg4; do
  timeout 900 python bench.py --config $c --steps 20 --warmup 5 2> $OUT/bench_$c.err | tail -1 > $OUT/bench_$c.json
done
timeout 600 python bench.py --config cfg2 --family laplace_with_softmax --loss dreg_looser --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_cfg2_laplace_dreg.json
# 2b. kernel traces of the other configurations (per-kernel shares; the JSON lines above are the unprofiled ones)
for c in cfg5 cfg4 cfg2; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$c -o t -- python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace_$c.log 2>&1
  python tools/rocpd_summary.py $(find $OUT/trace_$c -name "*_results.db" | head -1) > $OUT/${c}_kernel_stats.md 2>/dev/null
  python tools/step_groups.py $(find $OUT/trace_$c -name "*_results.db" | head -1) 5 70 > $OUT/${c}_step_groups.md 2>/dev/null
  rm -rf $OUT/trace_$c
done
# 3. trainer loop throughput
timeout 300 python tools/trainer_bench.py cfg3 5 2>/dev/null | tail -1 > $OUT/trainer_cfg3.json
timeout 300 python tools/trainer_bench.py cfg1 4 2>/dev/null | tail -1 > $OUT/trainer_cfg1.json
# 4. SQ / LDS counters of the convolution kernels (two passes, kernel-trace only)
bash tools/imgconv_pmc.sh $OUT/pmc_conv new > $OUT/pmc_conv.txt 2>&1
# 4a. SQ / LDS counters of the dense16 kernels (the MLP decoder on fp16 pair planes) + their isolated timings
bash tools/dense16_pmc.sh $OUT/pmc_d16 > $OUT/pmc_dense16.txt 2>&1
python tools/dense16_probe.py > $OUT/dense16_probe.txt 2>&1
# 4b. SQ / LDS counters of the register-stationary 3x3 kernels
bash tools/conv3_pmc.sh $OUT/pmc_conv3 > $OUT/pmc_conv3.txt 2>&1
python tools/conv3_probe.py f16 > $OUT/conv3_probe_f16.txt 2>&1
python tools/conv3_probe.py f16cfg4 > $OUT/conv3_probe_f16cfg4.txt 2>&1
# 4c. the ResNet configurations with the 3x3 kernels on bf16 pieces (A/B of the scaled-fp16 form)
for c in cfg5 cfg4; do
  MVK_TUNE=1 MVK_C3_F16=0 timeout 900 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > $OUT/bench_${c}_bf16x3.json
done
# 5. HBM traffic counters of the headline step (separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
  python tools/pmc_agg.py $OUT/pmc_$c/p_results.db | grep -E "n dur_us|recon_nll|small_up|imgconv|imgwgrad|d16_|smallk" > $OUT/pmc_$c.txt
done
# 6. the single-GPU RCCL path
MVK_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_force_dist.json
ls -la $OUT | head -40
# 7. round 6: same-box A/B lines that DESIGN.md / profiles/NOTES_r06.md quote (alternating)
line() { grep '^{' | tail -1; }
for i in 1 2 3; do
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line > $OUT/bench_ab_shipped_$i.json
  MVK_TUNE=1 MVK_ASSEMBLY_LAST=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line > $OUT/bench_ab_assembly_at_head_of_backward_$i.json
  MVK_TUNE=1 MVK_HEADS_BWD_MLP=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line > $OUT/bench_ab_mlp_heads_bwd_fused_$i.json
  MVK_TUNE=1 MVK_WGRAD_PAIR=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | line > $OUT/bench_ab_wgrad_two_launches_$i.json
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>/dev/null | line > $OUT/bench_ab_rotated_all_leaves_$i.json
  MVK_TUNE=1 MVK_ROT_SVHN=2 MVK_ROT_MLP=2 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --rotate 2>/dev/null | line > $OUT/bench_ab_rotated_first_layer_leaves_$i.json
done
# 8. the assembled-configuration tests once more for the rectifier counts (gpurun_out/flip_counts.jsonl) and the IWAE / DReG float64 distances
rm -f gpurun_out/flip_counts.jsonl gpurun_out/iwae_float64.jsonl
timeout 1200 python -m pytest tests/test_assembled_configs.py -m gpu -q -k "golden_gpu" > $OUT/pytest_assembled.log 2>&1; tail -3 $OUT/pytest_assembled.log
timeout 900 python -m pytest tests/test_gpu_golden.py -m gpu -q -k "mmvae" > $OUT/pytest_mmvae.log 2>&1; tail -2 $OUT/pytest_mmvae.log
cp gpurun_out/flip_counts.jsonl $OUT/flip_counts.jsonl 2>/dev/null
cp gpurun_out/iwae_float64.jsonl $OUT/iwae_float64.jsonl 2>/dev/null
# 9. the fused-tail probe alone
python tools/smallup_probe.py 5120 15 > $OUT/smallup_probe.txt 2>&1
ls $OUT | wc -l
