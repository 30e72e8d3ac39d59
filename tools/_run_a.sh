for i in 1 2 3 4; do for v in "MVK_SMALLK=0" "MVK_SMALLK_BWD=0" "MVK_SMALLK_BWD=1"; do env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',d['value'],d['ms_per_step'])"; done; done
