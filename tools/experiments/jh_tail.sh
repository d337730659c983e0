#!/bin/bash
# Where the slow requests of the JPEG-source leg are: latencies in request order (tools/latency/latency_probe with FL_PROBE_DUMP), the slow ones listed
# with their position in the run.   bash tools/experiments/jh_tail.sh [threads] [requests]
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-64}; N=${2:-4096}
cd $R && python3 -c "
import bench, shutil, os
fs = bench.synthetic_jpeg_files()
os.makedirs('/tmp/jhfiles', exist_ok=True)
for i, f in enumerate(fs): shutil.copy(f, '/tmp/jhfiles/src%d.jpg' % i)
"
F="/tmp/jhfiles/src0.jpg /tmp/jhfiles/src1.jpg /tmp/jhfiles/src2.jpg /tmp/jhfiles/src3.jpg"
for run in 1 2 3; do
FL_PROBE_DUMP=/tmp/lat.txt $R/tools/latency/latency_probe $T $N 1920 1080 'w=300&h=200' 3 0 0 0 $F 2>&1 | tail -1 | cut -c1-140
python3 - <<PY
import numpy as np
l = np.loadtxt('/tmp/lat.txt')
n = len(l)
print('  mean %.2f ms; share of the summed latency in requests above 5 ms: %.2f' % (l.mean(), l[l > 5].sum() / l.sum()))
slow = np.nonzero(l > 5)[0]
print('  %d requests above 5 ms; by tenth of the run:' % len(slow), np.histogram(slow, bins=10, range=(0, n))[0].tolist())
# clusters: runs of slow requests whose indices are within 64 of each other
if len(slow):
    cl = np.split(slow, np.nonzero(np.diff(slow) > 64)[0] + 1)
    print('  clusters (first index, members, worst ms):', [(int(c[0]), len(c), round(float(l[c].max()), 1)) for c in cl][:12])
PY
done
