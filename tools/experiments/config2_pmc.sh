#!/bin/bash
# PMC passes over config 2's resample kernel (streaming kernel, grayscale pre-op): bash tools/experiments/config2_pmc.sh
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/config2_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $R/bench.py --cpu-images 0 --latency-requests 0 --extra-steps 0 --verify-images 0 --steps 3 --warmup 1 --blur 10 --grayscale --frontend none"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/a -o p -- $P > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/b -o p -- $P > $O/b.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/[ab]/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = "stream" if "resample_stream" in r["Kernel_Name"] else "blur" if "blur_tile" in r["Kernel_Name"] else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k)
    for c in sorted(acc[k]): print(f"  {c:28s} {sum(acc[k][c]) / len(acc[k][c]):16.0f}  ({len(acc[k][c])} dispatches)")
PY
