set -x
timeout 1500 python -m pytest tests/test_gpu_loader.py tests/test_gpu_conv_local.py -x -q 2>&1 | tail -15
