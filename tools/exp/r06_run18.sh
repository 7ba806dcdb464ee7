timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -k "golden or floorplans or forward" 2>&1 | tail -3
for i in 1 2; do
python bench.py --config 2 --streams 1 --steps 3200 --warmup 100 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('tail on : graphs/s %.0f' % d['value'], 'us per forward %.1f' % (1e6 / d['value']))"
YOLAT_CLS_TAIL=0 python bench.py --config 2 --streams 1 --steps 3200 --warmup 100 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('tail off: graphs/s %.0f' % d['value'], 'us per forward %.1f' % (1e6 / d['value']))"
done
python bench.py --config 2 --streams 1 --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['op_breakdown_us'])"
python bench.py --config 1 --streams 1 --steps 800 --warmup 50 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('cfg1 tail on  us per forward %.1f' % (1e6 / d['value']))"
YOLAT_CLS_TAIL=0 python bench.py --config 1 --streams 1 --steps 800 --warmup 50 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('cfg1 tail off us per forward %.1f' % (1e6 / d['value']))"
