set -x
timeout 1200 python -m pytest tests/test_gpu_train_plan.py -x -q 2>&1 | tail -25
timeout 600 python tools/exp/bf16_contract.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_bf16_contract.txt
bash tools/exp/r06_lds_pmc.sh 2>&1 | tail -40
