#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 900 python3 tools/_gpu_rate.py --reps 5 c2::fif=4:no_tail_items=0:heavy_tiles=0 c2::fif=4:no_tail_items=0:heavy_tiles=16 c2::fif=4:no_tail_items=0:heavy_tiles=32 c2::fif=4:no_tail_items=0:heavy_tiles=64 c2::fif=4:no_tail_items=0:heavy_tiles=0 c2::fif=4:no_tail_items=0:heavy_tiles=32 c2::fif=1:no_tail_items=0:heavy_tiles=0 c2::fif=1:no_tail_items=0:heavy_tiles=32 c2::fif=1 2>&1 | tee $O/c2heavy_rate.txt
timeout -k 10 200 python3 tools/_gpu_variants.py --workload c2 --opt no_tail_items=0 --opt heavy_tiles=0 bibim_renderer_amd/libbibim_hip.so
timeout -k 10 200 python3 tools/_gpu_variants.py --workload c2 --opt no_tail_items=0 --opt heavy_tiles=32 bibim_renderer_amd/libbibim_hip.so
timeout -k 10 200 python3 tools/_gpu_variants.py --workload c2 --opt no_tail_items=0 --opt heavy_tiles=16 bibim_renderer_amd/libbibim_hip.so
