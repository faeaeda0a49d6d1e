#!/bin/bash
# Run on the GPU box from the repo root (via gpurun): the round's rocprofv3 evidence -> gpurun_out/<tag>_*
# usage: tools/profile_round.sh r02 [workload]   (copy the summaries you want judged into profiles/)
# Counter passes are separate runs with --kernel-trace only (never combined with sys/hip traces); every rocprofv3 run is
# wrapped in timeout: a pass with TA_* counters hung in round 2.
set -e
tag=$1
W=${2:-c3}
pre=$tag; [ "$W" != c3 ] && pre=${tag}_$W
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
RAW=/tmp/bbr_prof_raw   # the raw rocprofv3 directories stay on the box (gpurun merges at most 64 MiB back); only the summaries travel
mkdir -p $RAW $O
cd /tmp && export TMPDIR=/tmp
# (1) kernel trace + stats of the default bench command
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $RAW/${pre}_stats --output-format csv -- python3 $R/bench.py --workload $W --no-also > $O/${pre}_bench_under_rocprof.json 2> $O/${pre}_bench_under_rocprof.err
# (2) HBM traffic: separate passes
B="python3 $R/bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline --no-also"   # (one workload's launches only: the passes average per kernel)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $RAW/${pre}_pmc_fetch --output-format csv -- $B > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $RAW/${pre}_pmc_write --output-format csv -- $B > /dev/null 2>&1
# (3) executed instructions by class, issue / wait breakdown
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $RAW/${pre}_pmc_sq1 --output-format csv -- $B > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES -d $RAW/${pre}_pmc_sq2 --output-format csv -- $B > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 -d $RAW/${pre}_pmc_sq3 --output-format csv -- $B > /dev/null 2>&1
cd $R
python3 tools/profile_summary.py stats $RAW/${pre}_stats $O/${pre}_bench_kernel_stats.csv
python3 tools/profile_summary.py phases $RAW/${pre}_stats $O/${pre}_bench_under_rocprof.json $O/${pre}_bench_kernel_phases.txt
python3 tools/profile_summary.py hbm $W $RAW/${pre}_pmc_fetch $RAW/${pre}_pmc_write $O/${pre}_pmc_hbm.json
python3 tools/profile_summary.py counters $O/${pre}_pmc_sq.txt $RAW/${pre}_pmc_sq1 $RAW/${pre}_pmc_sq2 $RAW/${pre}_pmc_sq3
python3 tools/profile_summary.py counters_json $W $O/${pre}_pmc_sq.json $RAW/${pre}_pmc_sq1 $RAW/${pre}_pmc_sq2 $RAW/${pre}_pmc_sq3
cat $O/${pre}_bench_kernel_stats.csv; cat $O/${pre}_bench_kernel_phases.txt; cat $O/${pre}_pmc_hbm.json; cat $O/${pre}_bench_under_rocprof.json
