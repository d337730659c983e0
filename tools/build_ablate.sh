#!/bin/bash
# builds tools/libfanlin_gpu_ablate_<name>.so for each "name[:mask[:extra compiler flags]]" given: experiment variants of
# one kernel file (ABL_FILE, default fl_kernels.hip; FL_ABLATE masks are documented in the file itself)
set -e
cd "$(dirname "$0")/../fanlin-rs_amd/csrc"
mkdir -p /tmp/abl
KFILE=${ABL_FILE:-fl_kernels.hip}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
ALL="fl_kernels.hip fl_mfma.hip fl_wtile.hip fl_mfma_tables.cpp fl_context.cpp fl_batch.cpp fl_queue.cpp fl_cmyk_ctx.cpp fl_tables.cpp fl_query.cpp fl_cmyk.cpp fl_jpeghuff.cpp fl_jpeg.hip fl_jpegdec.hip fl_jpeghuff_dev.hip"
OTHERS=""; for f in $ALL; do [ "$f" = "$KFILE" ] || OTHERS="$OTHERS $f"; done
make -s fl_buildinfo.gen.cpp >/dev/null 2>&1 || true
for f in $OTHERS fl_buildinfo.gen.cpp; do /opt/rocm/bin/hipcc $FLAGS -x hip -c $f -o /tmp/abl/$f.o & done
for v in "$@"; do
  IFS=: read -r name mask extra <<< "$v"
  /opt/rocm/bin/hipcc $FLAGS -DFL_ABLATE=${mask:-0} ${extra} ${EXTRA} -x hip -c $KFILE -o /tmp/abl/k_$name.o &
done
wait
true
for v in "$@"; do
  name=${v%%:*}
  objs=""; for f in $OTHERS fl_buildinfo.gen.cpp; do objs="$objs /tmp/abl/$f.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libfanlin_gpu_ablate_$name.so /tmp/abl/k_$name.o $objs -ldl
done
ls -la ../../tools/*.so
