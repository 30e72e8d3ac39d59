timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_golden.py -x -q -m gpu -k "deferred or graph_replay or branch_streams or full_size or three_training or mnistsvhn or svhn" 2>&1 | tail -2
for i in 1 2 3; do for v in "MVK_LEAF_STREAM=0" "MVK_LEAF_STREAM=1"; do env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep -E '^\{|Error|error|Segm' | tail -1 | python -c "
import json,sys
t=sys.stdin.read()
try:
  d=json.loads(t);print('$v',d['value'],d['ms_per_step'])
except Exception: print('$v FAIL', t[:200])"; done; done
