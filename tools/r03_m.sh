#!/bin/bash
cd $GRAFT_REPO_ROOT
for mr in 32768 8192; do
echo "min_rows=$mr"
YOLAT_NODE3_SMALLK_MIN_ROWS=$mr timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('headline', round(d['value']), round(d['ms_per_step'],4), d.get('csr_merged_mode'))"
done
