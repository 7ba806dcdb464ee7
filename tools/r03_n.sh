#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_configs.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --mode train --config 3 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train3', d['ms_per_step'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', d['value'], d['ms_per_step'])"
