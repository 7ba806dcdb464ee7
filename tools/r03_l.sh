#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -2
for pd in 1 3; do
YOLAT_HGEMM_PREFETCH=$pd timeout 300 python bench.py --config 5 --precision bf16 --steps 30 --warmup 5 --streams 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pd=$pd cfg5 bf16', round(d['ms_per_step'],4), {k[:14]:round(v,1) for k,v in d['op_breakdown_us'].items()})"
done
