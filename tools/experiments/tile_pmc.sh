#!/bin/bash
# PMC passes over one geometry of the tiled two-pass kernel (runs on the GPU box): bash tools/experiments/tile_pmc.sh W H N [SRC_H SRC_W]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/tile_pmc_$1x$2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/experiments/one_geometry.py $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $P > $O/stats.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/a -o p -- $P > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/b -o p -- $P > $O/b.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/[ab]/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "resample_tile" in r["Kernel_Name"] or "resample_wtile" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(f"{k:28s} {sum(acc[k]) / len(acc[k]):16.0f}  ({len(acc[k])} dispatches)")
for r in csv.DictReader(open(glob.glob("$O/stats/*kernel_stats.csv")[0])):
    if "resample_tile" in r["Name"] or "resample_wtile" in r["Name"]: print("average ns", r["AverageNs"], "calls", r["Calls"])
PY
