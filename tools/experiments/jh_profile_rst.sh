#!/bin/bash
# The same probe as jh_profile.sh with the bench's JPEG sources written again with a restart interval of one MCU row (120 MCUs): the kernels that look for
# interval starts (RST instantiations), and how many rounds the chain of states needs when every row begins with a known state.
#   bash tools/experiments/jh_profile_rst.sh [threads] [requests]
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-64}; N=${2:-2048}
cd $R && python3 -c "
import bench, os, io
from PIL import Image
os.makedirs('/tmp/jhfiles', exist_ok=True)
for i, f in enumerate(bench.synthetic_jpeg_files()):
    im = Image.open(f)
    im.save('/tmp/jhfiles/rst%d.jpg' % i, 'JPEG', quality=85, subsampling=2, restart_marker_blocks=120)
    print(os.path.getsize('/tmp/jhfiles/rst%d.jpg' % i))
"
cd /tmp && export TMPDIR=/tmp
FLGPU_DEVICE_HUFFMAN_ALWAYS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/jh_prof_rst -o jh -- $R/tools/latency/latency_probe $T $N 1920 1080 "w=300&h=200" 3 0 0 0 /tmp/jhfiles/rst0.jpg /tmp/jhfiles/rst1.jpg /tmp/jhfiles/rst2.jpg /tmp/jhfiles/rst3.jpg 2>&1 | tail -1 | cut -c1-400
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/jh_prof_rst/**/jh_kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("%-70s calls %6s  total %10.3f ms  avg %9.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
