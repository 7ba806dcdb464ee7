timeout 600 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -3
for v in "YOLAT_EDGE_H_V=1" "YOLAT_EDGE_H_V=3"; do
  echo "== $v"; env $v timeout 300 python bench.py --config 5 --steps 60 --warmup 10 --no-cpu-baseline --streams 1 --precision bf16 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/forward %.3f'%l['ms_per_forward']); [print('   %-80s %8.1f'%(k[:80],v)) for k,v in l['op_breakdown_us'].items()]"
  echo "== cfg2 $v"; env $v timeout 300 python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline --streams 1 --precision bf16 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/forward %.3f'%l['ms_per_forward']); [print('   %-80s %8.1f'%(k[:80],v)) for k,v in l['op_breakdown_us'].items() if 'edge' in k]"
done
python tools/exp/bf16_err.py 2>&1 | grep blocks
