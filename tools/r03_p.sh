#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -k "x6 or rows or node_uv or fusion" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --config 5 --precision fp32 --steps 30 --warmup 5 --streams 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 fp32', round(d['ms_per_step'],4), {k[:14]:round(v,1) for k,v in d['op_breakdown_us'].items()})"
timeout 300 python bench.py --config 1 --precision fp32 --steps 50 --warmup 5 --streams 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg1 fp32', round(d['ms_per_step'],4), {k[:14]:round(v,1) for k,v in d['op_breakdown_us'].items()})"
