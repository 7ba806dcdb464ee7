timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_dist.py tests/test_gpu_loader.py -x -q 2>&1 | grep -v Warning | tail -30
