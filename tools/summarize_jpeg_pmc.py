#!/usr/bin/env python3
"""Condenses gpurun_out/jpeg_pmc (tools/collect_jpeg_pmc.sh) into one text file: per-dispatch averages of every counter for
the JPEG encode kernels (and the resample kernel beside them) + the kernel-trace statistics.   python tools/summarize_jpeg_pmc.py [out.txt]"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "jpeg_pmc")
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(O, "summary.txt")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(collections.Counter)
names = ("jpeg_dct_quant_kernel", "jpeg_pack_kernel", "jpeg_scan_kernel", "jpeg_emit_kernel", "jpeg_stuff_kernel", "resample_mfma_kernel")
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for name in names:
            if name in r["Kernel_Name"]:
                acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
                n[name][r["Counter_Name"]] += 1
with open(out_path, "w") as out:
    out.write("rocprofv3 --kernel-trace --pmc <one group per pass> -- python3 bench.py --steps 3 --warmup 1 (config 1 + JPEG encode, 1024 pictures per launch); per-dispatch averages\n\n")
    for name in names:
        if name not in acc:
            continue
        out.write(name + "\n")
        for c in sorted(acc[name]):
            out.write(f"  {c:28s} {acc[name][c] / n[name][c]:18.1f}  ({n[name][c]} dispatches)\n")
        out.write("\n")
    for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
        out.write("kernel trace statistics (" + os.path.basename(f) + ")\n")
        for line in open(f):
            if "fl::" in line or line.startswith('"Name"'):
                out.write(line)
print(open(out_path).read())
