#!/bin/bash
# Which runtime calls of the JPEG-source request path are slow, and how often buffers are (re)allocated in the steady state: the probe under
# rocprofv3 --hip-trace, per-call statistics and the slowest single calls.   bash tools/experiments/jh_hiptrace.sh [threads] [requests]
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-64}; N=${2:-4096}
cd $R && python3 -c "
import bench, shutil, os
fs = bench.synthetic_jpeg_files()
os.makedirs('/tmp/jhfiles', exist_ok=True)
for i, f in enumerate(fs): shutil.copy(f, '/tmp/jhfiles/src%d.jpg' % i)
"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/jh_hip
FL_PROBE_DUMP=/tmp/lat.txt rocprofv3 --hip-trace --stats --output-format csv -d $R/gpurun_out/jh_hip -o jh -- $R/tools/latency/latency_probe $T $N 1920 1080 "w=300&h=200" 3 0 0 0 /tmp/jhfiles/src0.jpg /tmp/jhfiles/src1.jpg /tmp/jhfiles/src2.jpg /tmp/jhfiles/src3.jpg 2>&1 | tail -1 | cut -c1-150
python3 -c "
import numpy as np
l = np.loadtxt('/tmp/lat.txt'); slow = np.nonzero(l > 5)[0]
print('requests above 5 ms:', len(slow), 'first indices', slow[:6].tolist(), 'worst %.1f ms' % l.max())"
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/jh_hip/**/jh_hip_api_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-40s calls %7s  total %10.3f ms  avg %9.1f us  max %10.1f us" % (r["Name"][:40], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
for f in glob.glob("$R/gpurun_out/jh_hip/**/jh_hip_api_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    names = ("hipMalloc", "hipFree", "hipHostMalloc", "hipHostFree")
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    at = lambda r: (int(r["Start_Timestamp"]) - t0) / 1e6
    end = max(at(r) for r in rows)
    print("process %.0f ms; calls other than allocations that took more than 3 ms (start ms, duration ms):" % end)
    for r in sorted((r for r in rows if r["Function"] not in names and dur(r) > 3.0), key=at):
        print("  %9.1f  %8.2f  %-28s tid %s" % (at(r), dur(r), r["Function"], r["Thread_Id"]))
    # one lane's cycle in the middle of the run: every call of the thread that launches most kernels, between two of its stream synchronisations
    from collections import Counter
    lane = Counter(r["Thread_Id"] for r in rows if r["Function"] == "hipLaunchKernel").most_common(1)[0][0]
    mine = sorted((r for r in rows if r["Thread_Id"] == lane), key=at)
    syncs = [i for i, r in enumerate(mine) if r["Function"] == "hipStreamSynchronize"]
    if len(syncs) > 40:
        a, b = syncs[len(syncs) // 2], syncs[len(syncs) // 2 + 2]
        base = at(mine[a]) + dur(mine[a])
        print("lane thread %s, two cycles from the middle of the run (ms after the first one's start, duration ms):" % lane)
        prev = None
        for r in mine[a:b + 1]:
            # runs of the same call are folded
            if prev and prev[0] == r["Function"] and r["Function"] != "hipStreamSynchronize": prev[2] += 1; prev[3] = at(r) + dur(r) - base; continue
            if prev: print("  %8.3f .. %8.3f  %-24s x %d" % (prev[1], prev[3], prev[0], prev[2]))
            prev = [r["Function"], at(r) - base, 1, at(r) + dur(r) - base]
        if prev: print("  %8.3f .. %8.3f  %-24s x %d" % (prev[1], prev[3], prev[0], prev[2]))
    alloc = [r for r in rows if r["Function"] in names]
    hist = {}
    for r in alloc: hist[int(at(r) // 50) * 50] = hist.get(int(at(r) // 50) * 50, 0) + 1
    print("allocation calls by 50 ms of the process:", sorted(hist.items()))
    print("allocations slower than 3 ms outside the first 600 ms:", [(round(at(r)), round(dur(r), 1), r["Function"]) for r in alloc if at(r) > 600 and dur(r) > 3.0][:20])
PY
rm -f $R/gpurun_out/jh_hip/*/*trace.csv   # (large; the statistics stay)
