#!/bin/bash
# (round 4) where the per-frame kernels wait: SQ activity / wait cycles, in-flight levels, L1 (TCP) requests, latency and stalls,
# texture-data-unit occupancy -- separate --pmc passes (more than seven counters, or the TD_*_WAVEFRONT ones, abort the profiler),
# kernels serialised by the profiler.  -> gpurun_out/r4/<tag>_memctr.txt (the reading: profiles/r04_memory_pipeline_counters.txt)
# usage (GPU box, via gpurun): tools/_gpu_memctr.sh <tag> [workload]
tag=$1; W=${2:-c3}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; RAW=/tmp/bbr_memctr; mkdir -p $O $RAW
cd /tmp && export TMPDIR=/tmp
B="python3 $R/tools/prof_frame.py --workload $W --frames 60 --opt frames_in_flight=3"
i=0; dirs=""
for P in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVES" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TAGRAM0_REQ_sum"; do
  i=$((i+1))
  if timeout -k 10 240 rocprofv3 --kernel-trace --pmc $P -d $RAW/${tag}_p$i --output-format csv -- $B > $O/${tag}_memctr_p$i.log 2>&1; then dirs="$dirs $RAW/${tag}_p$i"; else echo "pass $i failed"; fi
done
cd $R
python3 tools/profile_summary.py counters $O/${tag}_memctr.txt $dirs
grep -E "^k_shade  |^k_raster  |^k_geometry " $O/${tag}_memctr.txt
