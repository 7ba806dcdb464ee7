timeout 1200 python -m pytest tests/test_gpu_train_plan.py -x -q 2>&1 | tail -2
timeout 600 python tools/exp/train_plan_bench.py 3 fp32 40 2>&1 | grep -v amdgpu.ids | head -3
