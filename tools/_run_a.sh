timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "svhn or small or conv" 2>&1 | tail -1
python tools/smallup_probe.py 5120 7
