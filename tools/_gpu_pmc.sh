#!/bin/bash
# diagnostic (GPU box): per-kernel PMC counters of a build.  usage: tools/_gpu_pmc.sh <outfile> <workload> <lib.so> "<COUNTER ...>" ["<COUNTER ...>" ...]
R=${GRAFT_REPO_ROOT:-$PWD}; out=$R/gpurun_out/$1; W=$2; lib=$R/$3; shift 3
cd /tmp && export TMPDIR=/tmp
: > $out
k=0
for set in "$@"; do
  k=$((k+1)); d=/tmp/bbr_pmc_$$_$k; rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $d --output-format csv -- python3 $R/tools/_gpu_loop.py $W $lib 12 ${BBR_LOOP_FIF:-1} $BBR_LOOP_OPTS > /dev/null 2>&1 || { echo "pass $k failed" >> $out; continue; }
  python3 - $d >> $out <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        for k in ("k_geometry", "k_raster", "k_shade_items", "k_shade"):
            if k + "<" in name or name.endswith(k):
                short = k; break
        else:
            continue
        if short == "k_shade" and "ELb1ELb1E" in row["Kernel_Name"]: short = "k_shade_tail"
        acc[short][row["Counter_Name"]] += float(row["Counter_Value"]); n[short][row["Counter_Name"]] += 1
for kn in sorted(acc):
    print(kn, " ".join(f"{c}={acc[kn][c] / n[kn][c]:.0f}" for c in sorted(acc[kn])), f"(launches {max(n[kn].values())})")
PY
done
cat $out
