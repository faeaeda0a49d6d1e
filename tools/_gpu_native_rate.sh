#!/bin/bash
# Experiment (GPU box): frame rate of the native C++ loop (examples/shaderball_demo.cpp: bb::drawFrame, no Python).
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
g++ -std=c++17 -O2 -Iinclude examples/shaderball_demo.cpp -Lbibim_renderer_amd -lbibim_hip -Wl,-rpath,$R/bibim_renderer_amd -o /tmp/shaderball_demo
python3 - <<'PY'
import numpy as np
z = np.load("bibim_renderer_amd/data/shaderball_vertices.npz")
z[z.files[0]].tofile("/tmp/ball.bin")
PY
for args in "--size 1920 1080 --grid 1" "--size 3840 2160 --grid 4"; do
  /tmp/shaderball_demo --vertices-bin /tmp/ball.bin $args --frames 4000 --frames-in-flight 3 --out /tmp/o.ppm | head -1
done
