set -x
timeout 600 python tools/exp/train_plan_bench.py 3 fp32 40 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_train_plan_cfg3.txt
timeout 600 python tools/exp/train_plan_bench.py 5 fp32 15 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_train_plan_cfg5.txt
timeout 600 python tools/exp/train_plan_bench.py 5 bf16 15 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06_train_plan_cfg5.txt
timeout 600 python tools/exp/train_plan_bench.py 4 fp32 40 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_train_plan_cfg4.txt
timeout 2400 python -m pytest tests/test_gpu_dist.py tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_loader.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -15
