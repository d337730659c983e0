#!/bin/bash
# The JPEG-source request path at 128 callers against the queue shape: is the ceiling the lanes or the device?   bash tools/experiments/jh_capacity.sh
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && python3 -c "
import bench, shutil, os
fs = bench.synthetic_jpeg_files()
os.makedirs('/tmp/jhfiles', exist_ok=True)
for i, f in enumerate(fs): shutil.copy(f, '/tmp/jhfiles/src%d.jpg' % i)
"
F="/tmp/jhfiles/src0.jpg /tmp/jhfiles/src1.jpg /tmp/jhfiles/src2.jpg /tmp/jhfiles/src3.jpg"
for shape in "4 16" "6 16" "8 16" "4 32" "6 24" "8 12"; do set -- $shape
  echo "128 callers, lanes $1 x $2: $($R/tools/latency/latency_probe 128 8192 1920 1080 'w=300&h=200' 3 $1 $2 0 $F 2>&1 | tail -1 | cut -c1-125)"
done
