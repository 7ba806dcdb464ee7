#!/bin/bash
cd $GRAFT_REPO_ROOT
for items in 1048576 20000000; do
for prec in fp32 bf16; do
YOLAT_POOL_RIDER_ITEMS=$items timeout 300 python bench.py --config 5 --precision $prec --steps 30 --warmup 5 --streams 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('items=$items cfg5 $prec', round(d['ms_per_step'],4), {k[:14]:round(v,1) for k,v in d['op_breakdown_us'].items()})"
done
done
