timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "imgconv or conv or svhn" 2>&1 | tail -1
