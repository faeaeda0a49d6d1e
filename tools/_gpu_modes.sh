#!/bin/bash
# diagnostic (GPU box): bench lines and per-kernel times for several option sets of ONE build.
# usage: tools/_gpu_modes.sh <outdir> "<bench args A>" "<bench args B>" ...   (each string: extra bench.py arguments)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd $R
i=0
for a in "$@"; do
  i=$((i+1))
  for W in c3 c2 c5; do
    S=""; [ $W = c5 ] && S="--steps 60"
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload $W $S $a > $O/m${i}_$W.json 2> $O/m${i}_$W.err || { tail -5 $O/m${i}_$W.err; exit 1; }
  done
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 $a > $O/m${i}_c3drv.json 2>> $O/m${i}_c3.err || exit 1
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/m*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]; o = r.get("one_frame_in_flight", {})
        print(f"{os.path.basename(f):16s} {str(d['config'].get('options')):28s} tile {d['config']['tile']:6s} ms/step {d['ms_per_step']:.5f}  value {d['value']:.0f}  alone: shade {o.get('avg_kernel_ms')} raster {o.get('avg_raster_ms')} geom {o.get('avg_geometry_ms')} latency {o.get('avg_device_frame_latency_ms')}  fif2 {r.get('frames_in_flight_2', {}).get('ms_per_step')}")
    except Exception as e:
        print(f, "unreadable", e)
PY
