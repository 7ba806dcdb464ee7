#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_configs.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -2
bash tools/r03_i.sh
