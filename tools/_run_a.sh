timeout 600 python -m pytest tests/test_gpu_trainer.py -x -q -m gpu -k "deferred" 2>&1 | tail -3
