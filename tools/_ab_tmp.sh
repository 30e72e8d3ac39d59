python -m pytest tests -q -x -m gpu -k "cub or jmvae or mmvaeplus or assembled" 2>&1 | tail -3
for c in cfg5 cfg4; do echo "== $c"; timeout 600 python bench.py --config $c --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"], d.get(\"value\"))"; done
