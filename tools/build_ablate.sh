#!/bin/bash
# builds tools/libfanlin_gpu_ablate_<mask>.so for each mask given (experiment variants of the kernels)
set -e
cd "$(dirname "$0")/../fanlin-rs_amd/csrc"
mkdir -p /tmp/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
for f in fl_context.cpp fl_tables.cpp fl_query.cpp; do /opt/rocm/bin/hipcc $FLAGS -x hip -c $f -o /tmp/abl/$f.o & done
for m in "$@"; do /opt/rocm/bin/hipcc $FLAGS -DFL_ABLATE=$m ${EXTRA} -x hip -c fl_kernels.hip -o /tmp/abl/k_$m.o & done
wait
for m in "$@"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libfanlin_gpu_ablate_$m.so /tmp/abl/k_$m.o /tmp/abl/fl_context.cpp.o /tmp/abl/fl_tables.cpp.o /tmp/abl/fl_query.cpp.o; done
