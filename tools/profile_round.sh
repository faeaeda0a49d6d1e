#!/bin/bash
# Run on the GPU box from the repo root (via gpurun): the round's rocprofv3 evidence -> gpurun_out/<tag>_*
# usage: tools/profile_round.sh r01
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# (1) kernel trace + stats of the default bench command
rocprofv3 --kernel-trace --stats -d $O/${tag}_stats --output-format csv -- python3 $R/bench.py > $O/${tag}_bench_under_rocprof.json 2> $O/${tag}_bench_under_rocprof.err
# (2) HBM traffic: separate passes
B="python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${tag}_pmc_fetch --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${tag}_pmc_write --output-format csv -- $B > /dev/null 2>&1
# (3) issue / wait breakdown
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $O/${tag}_pmc_sq1 --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES -d $O/${tag}_pmc_sq2 --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU -d $O/${tag}_pmc_sq3 --output-format csv -- $B > /dev/null 2>&1
cd $R
python3 tools/profile_summary.py stats $O/${tag}_stats $O/${tag}_bench_kernel_stats.csv
python3 tools/profile_summary.py phases $O/${tag}_stats 543 200 $O/${tag}_bench_kernel_phases.txt
python3 tools/profile_summary.py hbm c3 $O/${tag}_pmc_fetch $O/${tag}_pmc_write $O/${tag}_pmc_hbm.json
python3 tools/profile_summary.py counters $O/${tag}_pmc_sq.txt $O/${tag}_pmc_sq1 $O/${tag}_pmc_sq2 $O/${tag}_pmc_sq3
cat $O/${tag}_bench_kernel_stats.csv; cat $O/${tag}_bench_kernel_phases.txt; cat $O/${tag}_pmc_hbm.json; cat $O/${tag}_bench_under_rocprof.json
