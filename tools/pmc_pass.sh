#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...>   (run on the GPU box from the repo root; one rocprofv3 --pmc pass)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_$tag --output-format csv -- python3 $R/tools/prof_frame.py --workload ${WORKLOAD:-c3} --frames 12 --tile-mode 1 --opt frames_in_flight=1 > /dev/null 2>&1
rc=$?
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_$tag | grep -v -A6 rocclr | grep -v "^--"
exit $rc
