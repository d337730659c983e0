#!/bin/bash
# Which counter follows the placement mode?  N fresh processes of the flagship batch under rocprofv3 --pmc; per process: the resample
# kernel's mean duration and its counters (runs on the GPU box).   bash tools/experiments/placement_pmc.sh [N]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/placement_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
N=${1:-8}
for i in $(seq 1 $N); do
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE \
    --output-format csv -d $O/r$i -o p -- python3 $R/tools/experiments/one_geometry.py 300 200 1024 > $O/r$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/r*/")):
    acc = collections.defaultdict(list); dur = []
    for f in glob.glob(d + "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "resample_mfma" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "resample_mfma" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    dur = sorted(dur)[: max(1, len(dur) - 1)]  # (drop the slowest: the cold first launch)
    print(d.split("/")[-2], f"kernel {sum(dur) / len(dur):.4f} ms", " ".join(f"{k.replace('_sum', '')}={sum(v) / len(v):.4g}" for k, v in sorted(acc.items())))
PY
