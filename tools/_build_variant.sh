#!/bin/bash
# diagnostic: build the library with extra flags into tools/_keep/variants/<name>.so   usage: _build_variant.sh name [srcdir] [flags...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=bibim_renderer_amd/csrc
if [ -d "$1" ]; then src=$1; shift; fi
mkdir -p tools/_keep/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -shared -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 \
  -Iinclude "$@" $src/bibim_hip.hip $src/bb_scene.cpp $src/bb_assets.cpp -o tools/_keep/variants/$name.so -lz
