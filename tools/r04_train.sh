#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attr_dw or factorised" 2>&1 | tail -3
python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k "soak or chained" 2>&1 | tail -3
for prec in fp32 bf16; do
python bench.py --mode train --config 5 --precision $prec --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$prec', d['ms_per_step']); print(json.dumps(d.get('op_breakdown_us')))"
done
python bench.py --mode train --config 3 --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('cfg3', d['ms_per_step']); print(json.dumps(d.get('op_breakdown_us')))"
