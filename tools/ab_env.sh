#!/bin/bash
# Same-box A/B of one environment switch on the headline train step: tools/ab_env.sh VAR [values...] (default 0 1 0 1)
VAR=$1; shift; VALS=${*:-"0 1 0 1"}
for z in $VALS; do
env $VAR=$z timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-modes --no-d4 --no-inference 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('$VAR=$z', d['value'], d['ms_per_step'])
"
done
