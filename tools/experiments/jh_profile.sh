#!/bin/bash
# Kernel statistics of the JPEG-source request path (tools/latency/latency_probe with files as sources) under rocprofv3:
# which kernels a batch of file requests spends its device time in.   bash tools/experiments/jh_profile.sh [threads] [requests]
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-64}; N=${2:-2048}
cd $R && python3 -c "
import bench, shutil, os
fs = bench.synthetic_jpeg_files()
os.makedirs('/tmp/jhfiles', exist_ok=True)
for i, f in enumerate(fs): shutil.copy(f, '/tmp/jhfiles/src%d.jpg' % i)
"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/jh_prof -o jh -- $R/tools/latency/latency_probe $T $N 1920 1080 "w=300&h=200" 3 0 0 0 /tmp/jhfiles/src0.jpg /tmp/jhfiles/src1.jpg /tmp/jhfiles/src2.jpg /tmp/jhfiles/src3.jpg 2>&1 | tail -2
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/jh_prof/**/jh_kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print("%-70s calls %6s  total %10.3f ms  avg %9.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
