#!/bin/bash
# the whole GPU suite in one process, output kept under gpurun_out/<tag>_pytest.txt   usage: tools/_gpu_suite.sh <tag> [pytest args]
R=${GRAFT_REPO_ROOT:-$PWD}; tag=$1; shift; O=$R/gpurun_out/r5; mkdir -p $O; cd $R
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q "$@" > $O/${tag}_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/${tag}_pytest.txt
tail -6 $O/${tag}_pytest.txt
