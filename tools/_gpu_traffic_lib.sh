#!/bin/bash
# diagnostic: k_shade's FETCH_SIZE per launch for a given build of the library   usage: _gpu_traffic_lib.sh lib.so [workload]
R=${GRAFT_REPO_ROOT:-$PWD}
L=$1; W=${2:-c3}
O=$R/gpurun_out/traffic_lib_$(basename $L .so)
cd /tmp && export TMPDIR=/tmp
rm -rf $O
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O --output-format csv -- python3 $R/tools/prof_frame.py --workload $W --frames 20 --tile-mode 1 --lib $R/$L > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if "k_shade<" in n and not n.rstrip().endswith("true>"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = acc["FETCH_SIZE"][5:]
print("$L  k_shade FETCH_SIZE x2 = %.1f MB per launch (n=%d)" % (sum(v) / max(len(v), 1) * 1024 * 2 / 1e6, len(v)))
PY
