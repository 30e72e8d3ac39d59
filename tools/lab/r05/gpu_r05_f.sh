#!/bin/bash
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05f; mkdir -p $OUT
bash tools/d16_run.sh base os os3 os4f osf 2>&1 | tee $OUT/d16_variants.txt
