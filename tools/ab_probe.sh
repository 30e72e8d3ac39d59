#!/bin/bash
# On the GPU box: tools/imgconv_f16_probe.py and one headline bench line per library variant given as arguments
OUT=gpurun_out/${TAG:-abp}; mkdir -p $OUT; export TMPDIR=/tmp
for lib in "$@"; do
  echo "== $lib" | tee -a $OUT/probe.txt
  MVK_LIB_PATH=$PWD/$lib timeout 300 python tools/imgconv_f16_probe.py 2>&1 | grep "^|" | tee -a $OUT/probe.txt
  MVK_LIB_PATH=$PWD/$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3', d['ms_per_step'], d.get('ms_per_step_median'))" | tee -a $OUT/probe.txt
done
