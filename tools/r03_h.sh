#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "stream_kernel or factorised or node_uv" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
for prec in fp32 bf16; do
timeout 300 python bench.py --config 5 --precision $prec --steps 30 --warmup 5 --streams 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 $prec', d['ms_per_step']); print({k:round(v,1) for k,v in d['op_breakdown_us'].items()})"
done
