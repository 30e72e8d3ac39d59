#!/bin/bash
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05z; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "mopoe or fullsize or trainer or fused or dense16 or mlp" 2>&1 | tail -4
