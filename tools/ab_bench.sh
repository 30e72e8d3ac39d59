#!/bin/bash
# On the GPU box: same-box A/B of two builds of the library (MVK_LIB_PATH), alternating pairs of `bench.py --steps 30`.
#   tools/ab_bench.sh <tag> <base.so> <new.so> [pairs] [configs...]
TAG=${1:-ab}; BASE=$2; NEW=$3; PAIRS=${4:-3}; shift 4; CFGS=${@:-cfg3}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for cfg in $CFGS; do
  for i in $(seq $PAIRS); do
    for which in base new; do
      lib=$BASE; [ $which = new ] && lib=$NEW
      MVK_LIB_PATH=$PWD/$lib timeout 900 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline 2>> $OUT/err.log | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg $which', d['ms_per_step'], d.get('ms_per_step_median'))" | tee -a $OUT/ab.txt
    done
  done
done
