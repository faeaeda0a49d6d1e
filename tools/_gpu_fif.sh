#!/bin/bash
# diagnostic: frame period against frames in flight
for w in c2 c3; do for f in 2 3 4; do
  timeout -k 10 100 python bench.py --workload $w --frames-in-flight $f --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w frames_in_flight $f: %.1f us/step  %.0f Mpix/s' % (d['ms_per_step']*1e3, d['value']))"
done; done
