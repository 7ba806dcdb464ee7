for s in 1 2 4 8; do
python bench.py --config 2 --streams $s --steps 3200 --warmup 100 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('streams', d['config']['streams_in_flight'], 'graphs/s %.0f' % d['value'], 'us per forward %.1f' % (1e6 / d['value']))"
done
