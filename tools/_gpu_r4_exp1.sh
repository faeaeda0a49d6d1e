#!/bin/bash
# round 4, experiment 1: where the frame's chain and the host's turnaround cost throughput (options tail_mode, run_ahead, stage_serial)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
python3 tools/_gpu_rate.py --reps 3 \
  c3 c3::tail_mode=1 c3::tail_mode=2 c3::tail_mode=3 c3::run_ahead=1 c3::run_ahead=1:tail_mode=1 \
  c3::run_ahead=1:tail_mode=1:stage_serial=1 c3::run_ahead=1:tail_mode=1:stage_serial=2 c3::run_ahead=1:tail_mode=1:stage_serial=3 \
  c3::stream_layout=1 c3::stream_layout=0 c3::fif=2 c3::fif=2:run_ahead=1:tail_mode=1 c3::fif=4 c3::fif=4:run_ahead=1 c3 \
  c5 c5::tail_mode=1 c5::run_ahead=1 c5::run_ahead=1:tail_mode=1 c5::run_ahead=1:tail_mode=1:stage_serial=1 \
  c5::run_ahead=1:tail_mode=1:stage_serial=3 c5::stream_layout=1 c5::stream_layout=0 c5::fif=2 c5::fif=2:run_ahead=1:tail_mode=1 c5 \
  c2 c2::run_ahead=1 c2::fif=3:run_ahead=1 c2 > $O/exp1.txt 2>&1
cat $O/exp1.txt
