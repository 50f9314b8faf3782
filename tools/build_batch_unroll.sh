#!/bin/bash
# timing experiment: the whole library with the batched kernel's tile at 2 / 8 lane-vectors per lane
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ablate
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 -Iinclude -Imct_quantizers_amd/csrc"
SRC=$(ls mct_quantizers_amd/csrc/mctq_*.hip)
for u in 2 8; do
  ( hipcc $F -DMCTQ_BATCH_U=$u -shared -o tools/ablate/libmctq_U$u.so $SRC ) &
done
wait
ls -la tools/ablate/libmctq_U*.so
