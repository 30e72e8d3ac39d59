#!/bin/bash
# On the GPU box: headline bench under a list of environment settings ("A=1 B=2" per argument; "-" = none), alternating, N rounds.
#   ROUNDS=2 CFG=cfg3 tools/env_ab.sh - "MVK_IMGWGRAD_GRID=192"
OUT=gpurun_out/${TAG:-envab}; mkdir -p $OUT; export TMPDIR=/tmp MVK_TUNE=1
for i in $(seq ${ROUNDS:-2}); do
  for e in "$@"; do
    envs=""; [ "$e" != "-" ] && envs="$e"
    env $envs timeout 600 python bench.py --config ${CFG:-cfg3} --steps 30 --warmup 5 --no-cpu-baseline 2>> $OUT/err.log | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${CFG:-cfg3} [$e]', d['ms_per_step'], d.get('ms_per_step_median'))" | tee -a $OUT/envab.txt
  done
done
