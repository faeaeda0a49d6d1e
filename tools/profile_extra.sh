#!/bin/bash
# Run on the GPU box from the repo root (via gpurun): the other single-GPU BASELINE configurations (C2 1080p / 1 light,
# C5 8K / 64 balls / 8 lights) under rocprofv3 -> gpurun_out/<tag>_<workload>_*.  C3 is tools/profile_round.sh.
# usage: tools/profile_extra.sh r01
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for wl in c2 c5; do
  steps=200; [ $wl = c5 ] && steps=60
  B="python3 $R/bench.py --workload $wl --steps $steps --warmup 10 --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats -d $O/${tag}_${wl}_stats --output-format csv -- $B > $O/${tag}_${wl}_bench_under_rocprof.json 2> $O/${tag}_${wl}_bench.err
  B="python3 $R/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${tag}_${wl}_pmc_fetch --output-format csv -- $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${tag}_${wl}_pmc_write --output-format csv -- $B > /dev/null 2>&1
  echo "$wl traced"
done
cd $R
for wl in c2 c5; do
  steps=200; [ $wl = c5 ] && steps=60
  python3 tools/profile_summary.py stats $O/${tag}_${wl}_stats $O/${tag}_${wl}_bench_kernel_stats.csv
  python3 tools/profile_summary.py phases $O/${tag}_${wl}_stats 533 $steps $O/${tag}_${wl}_bench_kernel_phases.txt
  python3 tools/profile_summary.py hbm $wl $O/${tag}_${wl}_pmc_fetch $O/${tag}_${wl}_pmc_write $O/${tag}_${wl}_pmc_hbm.json
  cat $O/${tag}_${wl}_bench_kernel_phases.txt $O/${tag}_${wl}_bench_under_rocprof.json
done
