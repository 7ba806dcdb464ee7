timeout 1500 python -m pytest tests/test_gpu_conv_local.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -4
CL_STAMPS=1 timeout 300 python tools/exp/conv_local_abl.py 5 0 20 2>&1 | grep -v amdgpu | sed -n 1,28p
timeout 600 python tools/exp/cfg5_locality_bench.py 5 50 2>&1 | grep -E "ms per forward|conv_local|bit-ident"
