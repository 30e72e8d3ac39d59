#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x -k "conv_down_up_wgrad or conv_wgrad or svhn or fullsize or mnistsvhn or trainer or mmvae or mopoe or golden" 2>&1 | tail -4
