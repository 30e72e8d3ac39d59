OUT=gpurun_out/r02; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 900 python bench.py --config cfg4 --batch 128 --no-graph --steps 5 --warmup 2 --no-cpu-baseline 2> $OUT/bench_cfg4_b128_eager.err | grep '^{' | tail -1 > $OUT/bench_cfg4_b128_eager.json
MVK_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_force_dist.json
tail -5 $OUT/pytest_gpu.log; tail -3 $OUT/bench_cfg4_b128_eager.err; cat $OUT/bench_cfg4_b128_eager.json $OUT/bench_force_dist.json
