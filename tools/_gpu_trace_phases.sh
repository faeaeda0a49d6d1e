#!/bin/bash
# kernel trace of the default bench command for a workload -> per-kernel phases + timeline   usage: tools/_gpu_trace_phases.sh <tag> [workload] [extra bench args]
tag=$1; W=${2:-c3}; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=""; [ $W = c5 ] && S="--steps 60"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/${tag}_${W}_stats --output-format csv -- python3 $R/bench.py --workload $W $S --no-cpu-baseline "$@" > $O/${tag}_${W}_under_rocprof.json 2> $O/${tag}_${W}_under_rocprof.err
cd $R
python3 tools/profile_summary.py phases $O/${tag}_${W}_stats $O/${tag}_${W}_under_rocprof.json $O/${tag}_${W}_phases.txt
cat $O/${tag}_${W}_phases.txt
python3 tools/trace_timeline.py $O/${tag}_${W}_stats 100 160 | tail -22
