timeout 1500 python -m pytest tests/test_gpu_conv_local.py -x -q 2>&1 | tail -3
CL_STAMPS=1 timeout 300 python tools/exp/conv_local_abl.py 5 0 20 2>&1 | grep -E "per launch|tile 1|tile load|total"
timeout 600 python tools/exp/cfg5_locality_bench.py 5 50 2>&1 | grep -E "ms per forward|conv_local|graph_prep|bit-ident"
