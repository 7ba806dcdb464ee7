set -x
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_conv_local.py -x -q 2>&1 | tail -15
timeout 600 python tools/exp/cfg5_locality_bench.py 5 50 2>&1 | tail -60 | tee gpurun_out/cfg5_locality.txt
