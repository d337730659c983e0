#!/bin/bash
# The request path (tools/latency/latency_probe, 64 callers) against the queue's shape, lanes x flush size, for the three kinds of request
# the bench's latency legs send: JPEG files as sources, pixel buffers (pageable), pixel buffers with a blur.
#   bash tools/experiments/jh_lanes.sh [threads] [requests]
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-64}; N=${2:-4096}
cd $R && python3 -c "
import bench, shutil, os
fs = bench.synthetic_jpeg_files()
os.makedirs('/tmp/jhfiles', exist_ok=True)
for i, f in enumerate(fs): shutil.copy(f, '/tmp/jhfiles/src%d.jpg' % i)
"
F="/tmp/jhfiles/src0.jpg /tmp/jhfiles/src1.jpg /tmp/jhfiles/src2.jpg /tmp/jhfiles/src3.jpg"
SHAPES=${SHAPES:-"3,32 4,32 4,24 4,16 5,16 4,12 5,12"}
for shape in $SHAPES; do shape=${shape/,/ }
  set -- $shape
  echo "lanes $1 max_batch $2"
  echo "  files : $($R/tools/latency/latency_probe $T $N 1920 1080 'w=300&h=200' 3 $1 $2 0 $F 2>&1 | tail -1 | cut -c1-130)"
  echo "  pixels: $($R/tools/latency/latency_probe $T $N 1920 1080 'w=300&h=200' 3 $1 $2 0 2>&1 | tail -1 | cut -c1-130)"
  echo "  blur10: $($R/tools/latency/latency_probe $T $N 1920 1080 'w=300&h=200&blur=10' 3 $1 $2 0 $F 2>&1 | tail -1 | cut -c1-130)"
done
