timeout 600 python -m pytest tests/test_gpu_golden.py -x -q -m gpu -k resnet_mmnist_nets 2>&1 | grep -E "AssertionError" | head -3
MVK_SMALLK=0 timeout 600 python -m pytest tests/test_gpu_golden.py -x -q -m gpu -k resnet_mmnist_nets 2>&1 | tail -1
