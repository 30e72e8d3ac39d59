timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "^E  |FAILED|passed|failed" | head -10
for i in 1 2 3; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'])"; done
