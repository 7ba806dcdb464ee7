set -x
timeout 1500 python -m pytest tests/test_gpu_conv_local.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -5
timeout 600 python tools/exp/cfg5_locality_bench.py 5 50 2>&1 | grep -E "ms per forward|conv_local|bit-ident"
bash tools/exp/r06_lds_pmc.sh > /dev/null 2>&1
grep -E "^====|k_conv_local_h" gpurun_out/r06_conv_local_lds_by_phase.txt | cut -c1-260
