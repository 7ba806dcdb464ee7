#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "fusion or x6 or rows" 2>&1 | tail -4
timeout 300 python tools/exp/fx_abl.py 2 5 1 2>&1 | grep "^cfg"
for g in 2 8; do echo "groups=$g"; YOLAT_FUSION_X6_GROUPS=$g timeout 300 python tools/exp/fx_abl.py 2 5 2>&1 | grep "^cfg"; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --mode train --config 3 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train3', d['ms_per_step'])"
