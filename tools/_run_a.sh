OUT=gpurun_out/r02; mkdir -p $OUT
timeout 900 python bench.py --config cfg4 --batch 64 --no-graph --steps 5 --warmup 2 --no-cpu-baseline 2> $OUT/bench_cfg4_b64_eager.err | grep "^{" | tail -1 > $OUT/bench_cfg4_b64_eager.json; cut -c1-260 $OUT/bench_cfg4_b64_eager.json
