#!/bin/bash
# The JPEG-source request path (tools/latency/latency_probe, 64 callers, files as sources) against the queue's shape: lanes x flush size.
#   bash tools/experiments/jh_lanes.sh [threads] [requests]
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-64}; N=${2:-4096}
cd $R && python3 -c "
import bench, shutil, os
fs = bench.synthetic_jpeg_files()
os.makedirs('/tmp/jhfiles', exist_ok=True)
for i, f in enumerate(fs): shutil.copy(f, '/tmp/jhfiles/src%d.jpg' % i)
"
for shape in "3 32" "4 32" "6 32" "4 16" "6 16" "8 16"; do
  set -- $shape
  echo "lanes $1 max_batch $2: $($R/tools/latency/latency_probe $T $N 1920 1080 'w=300&h=200' 3 $1 $2 0 /tmp/jhfiles/src0.jpg /tmp/jhfiles/src1.jpg /tmp/jhfiles/src2.jpg /tmp/jhfiles/src3.jpg 2>&1 | tail -1 | cut -c1-330)"
done
