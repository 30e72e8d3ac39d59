#!/bin/bash
set -u
export TMPDIR=/tmp; OUT=gpurun_out/r05l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_trainer.py -m gpu -q -x -k "user_decoder or fused_decoder_tail or two_ranks" 2>&1 | tail -15
