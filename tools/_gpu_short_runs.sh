#!/bin/bash
# diagnostic: what a short (driver-style) bench run costs per step, and where it goes
mkdir -p gpurun_out/r2
for v in "" "--no-timing-events" "--event-stride 5" "--event-stride 10"; do
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('steps 20 [$v]: %.1f us/step  value %.0f  k_shade %s us (n=%s)' % (d['ms_per_step']*1e3, d['value'], r.get('avg_kernel_ms') and round(r['avg_kernel_ms']*1e3,1), r.get('launches_timed')))"
  done
done
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('steps 200: %.1f us/step  value %.0f  k_shade %s us' % (d['ms_per_step']*1e3, d['value'], round(r['avg_kernel_ms']*1e3,1)))"
