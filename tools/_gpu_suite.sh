#!/bin/bash
# the whole GPU suite in one process, output kept under gpurun_out/r5/<tag>_pytest.txt   usage: tools/_gpu_suite.sh <tag> [pytest args]
# last line: the suite's wall time against the budget this repo holds itself to (400 s; the driver's limit is 900 s)
R=${GRAFT_REPO_ROOT:-$PWD}; tag=$1; shift; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
t0=$(date +%s)
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q --durations=15 "$@" > $O/${tag}_pytest.txt 2>&1; rc=$?
t1=$(date +%s)
echo "pytest rc=$rc" >> $O/${tag}_pytest.txt
echo "GPU suite wall time $((t1 - t0)) s (budget 400 s)" >> $O/${tag}_pytest.txt
tail -24 $O/${tag}_pytest.txt
exit $rc
