for f in 2.0 1.0; do for b in 2.0 1.0; do
  for cfg in 3 4; do
  YOLAT_FACT_FWD_RATIO=$f YOLAT_FACT_BWD_RATIO=$b python bench.py --mode train --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg $cfg fwd $f bwd $b', d['value'], d['ms_per_step'])"
  done
done; done
