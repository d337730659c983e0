#!/bin/bash
# Counters of the JPEG decode kernels (the probe of jh_profile.sh under rocprofv3 --pmc, one pass per counter group): per-dispatch averages per kernel.
#   bash tools/experiments/jh_pmc.sh
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && python3 -c "
import bench, shutil, os
fs = bench.synthetic_jpeg_files()
os.makedirs('/tmp/jhfiles', exist_ok=True)
for i, f in enumerate(fs): shutil.copy(f, '/tmp/jhfiles/src%d.jpg' % i)
"
cd /tmp && export TMPDIR=/tmp
export FLGPU_DEVICE_HUFFMAN_ALWAYS=1
P="$R/tools/latency/latency_probe 64 1024 1920 1080 w=300&h=200 3 0 0 0 /tmp/jhfiles/src0.jpg /tmp/jhfiles/src1.jpg /tmp/jhfiles/src2.jpg /tmp/jhfiles/src3.jpg"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SMEM"; do
  d=$R/gpurun_out/jh_pmc/$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o p -- $P > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/jh_pmc/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not any(x in k for x in ("jpeg_color", "jpeg_idct", "jh_write", "jh_sync", "jh_init")): continue
    print("==", k)
    for c in sorted(acc[k]): v = acc[k][c]; print("   %-22s %14.1f  (%d dispatches)" % (c, sum(v) / len(v), len(v)))
PY
