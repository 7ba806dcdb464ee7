timeout 1200 python -m pytest tests/test_gpu_conv_local.py -x -q -k "without_edges" 2>&1 | tail -15
