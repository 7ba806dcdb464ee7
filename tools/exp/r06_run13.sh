timeout 1200 python -m pytest tests/test_gpu_train_plan.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3
timeout 600 python tools/exp/train_plan_bench.py 3 fp32 40 2>&1 | grep -v amdgpu.ids | head -4
timeout 600 python tools/exp/train_plan_bench.py 4 fp32 40 2>&1 | grep -v amdgpu.ids | head -2
timeout 600 python tools/exp/train_plan_bench.py 5 fp32 15 2>&1 | grep -v amdgpu.ids | head -2
timeout 600 python tools/exp/train_plan_bench.py 5 bf16 15 2>&1 | grep -v amdgpu.ids | head -2
