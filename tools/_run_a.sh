for t in 256 384 512 768; do
MVK_SPLITK_TARGET_1024=$t timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('t1024=$t',d['value'],d['ms_per_step'])"
done
for t in 128 256; do
MVK_SPLITK_TARGET_512=$t timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('t512=$t',d['value'],d['ms_per_step'])"
done
