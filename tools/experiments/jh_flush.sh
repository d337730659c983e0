#!/bin/bash
# The queue's flush timer against the JPEG-source leg (tools/latency/latency_probe, FL_PROBE_FLUSH_US).   bash tools/experiments/jh_flush.sh [threads] [requests]
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-64}; N=${2:-4096}
cd $R && python3 -c "
import bench, shutil, os
fs = bench.synthetic_jpeg_files()
os.makedirs('/tmp/jhfiles', exist_ok=True)
for i, f in enumerate(fs): shutil.copy(f, '/tmp/jhfiles/src%d.jpg' % i)
"
F="/tmp/jhfiles/src0.jpg /tmp/jhfiles/src1.jpg /tmp/jhfiles/src2.jpg /tmp/jhfiles/src3.jpg"
for us in ${TIMERS:-200 25 50 100 200 300 400 800}; do
  for rep in 1 2; do
    echo "flush $us us: $(FL_PROBE_FLUSH_US=$us $R/tools/latency/latency_probe $T $N 1920 1080 'w=300&h=200' 3 0 0 0 $F 2>&1 | tail -1 | cut -c1-132)"
  done
done
