for it in 1048576 4194304; do
  for rep in 1 2; do
    YOLAT_POOL_RIDER_ITEMS=$it python bench.py --config 1 --steps 300 --warmup 30 --streams 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('items $it', d['value'], d['ms_per_forward'])"
  done
done
