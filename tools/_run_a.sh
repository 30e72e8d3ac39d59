timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "svhn or small or conv" 2>&1 | tail -1
for v in "MVK_SMALL_BWD_DENSE=0" "MVK_SMALL_BWD_DENSE=1" "MVK_SMALL_BWD_DENSE=0" "MVK_SMALL_BWD_DENSE=1"; do echo -n "$v  "; env $v python tools/smallup_probe.py 5120 7 | grep bwd | cut -c1-45; done
