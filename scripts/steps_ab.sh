#!/bin/bash
# bench.py's rate against the length of its timed region (same box): bash scripts/steps_ab.sh
for rep in 1 2; do
for args in "--steps 20 --warmup 3" "--steps 40 --warmup 5" "--steps 100 --warmup 5" "--steps 200 --warmup 5"; do
  timeout -k 10 200 python bench.py --gpus 1 $args --timed-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$args: %.1f searches/s  %.3f ms/step ' % (d['value'], d['ms_per_step']), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})"
done
done
