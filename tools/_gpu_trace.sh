#!/bin/bash
# diagnostic: kernel trace of N frames of a workload, one frame in flight (each kernel alone) -> per-kernel mean durations
R=${GRAFT_REPO_ROOT:-$PWD}
W=${1:-c3}
LIB=${2:+--lib $2}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_$W
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trace_$W --output-format csv -- python3 $R/tools/prof_frame.py --workload $W --frames 30 --tile-mode 1 --opt frames_in_flight=1 $LIB > /dev/null 2>&1
cd $R && python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/trace_$W/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v = v[len(v)//3:]
    print(f"{k:62s} n={len(v):4d} mean {sum(v)/len(v):8.2f} us  min {min(v):8.2f}")
PY
