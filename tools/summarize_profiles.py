#!/usr/bin/env python3
"""Turns gpurun_out/profiles_raw/ (tools/collect_profiles.sh, run on the GPU box) into the committed summaries
under profiles/: per-config rocprofv3 kernel statistics, the PMC counter averages of the dominant kernel, the
measured HBM traffic bench.py reports as roofline.traffic, and the bench lines.   python tools/summarize_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", "profiles_raw")
OUT = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def kernel_stats(sub, prefix, dst):
    src = os.path.join(RAW, sub, f"{prefix}_kernel_stats.csv")
    rows = list(csv.DictReader(open(src)))
    keep = [r for r in rows if "fl::" in r["Name"]]
    with open(os.path.join(OUT, dst), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)
    return keep


def counters(sub, kernel):
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(os.path.join(RAW, sub, "p_counter_collection.csv"))):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
    # one row per dispatch and counter: average over the dispatches
    return {k: acc[k] / n[k] for k in acc}, (max(n.values()) if n else 0)


def main():
    os.makedirs(OUT, exist_ok=True)
    kernel_stats("config1", "c1", f"{tag}_config1_kernel_stats.csv")
    kernel_stats("config2", "c2", f"{tag}_config2_kernel_stats.csv")
    kernel_stats("jpeg", "jp", f"{tag}_config1_resize_only_kernel_stats.csv")
    try:
        kernel_stats("config1_packed", "c1p", f"{tag}_config1_packed_arithmetic_kernel_stats.csv")
    except OSError:
        pass
    try:  # roctx ranges of the runtime (FLGPU_ROCTX=1) as rocprofv3 --marker-trace reports them
        import shutil
        for f in glob.glob(os.path.join(RAW, "markers", "*marker*stats*.csv")) + glob.glob(os.path.join(RAW, "markers", "*/*marker*stats*.csv")):
            shutil.copy(f, os.path.join(OUT, f"{tag}_roctx_marker_stats.csv"))
            break
    except Exception:
        pass
    bench = json.load(open(os.path.join(RAW, "bench_config1.json")))
    kernel = bench["roofline"].get("kernel", "resample_stream_kernel")   # the dominant kernel of config 1, as bench.py names it
    vals, nd = {}, 0
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_mfma"):
        try:
            v, n = counters(sub, kernel)
        except OSError:
            continue          # (an optional pass whose counters this rocprofv3 does not know)
        vals.update(v)
        nd = max(nd, n)
    workload = bench["config"]["workload"]
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half of wide coalesced reads -> FETCH x 2
    hbm = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    k_ms = bench["roofline"]["kernel_ms"]
    with open(os.path.join(OUT, f"{tag}_config1_pmc.txt"), "w") as f:
        f.write("rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-images 0 --latency-requests 0\n")
        f.write("separate passes: {FETCH_SIZE} {WRITE_SIZE} {SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY} "
                "{SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE} {SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES}\n")
        f.write(f"per-dispatch averages of fl::{kernel} ({workload}), {nd} dispatches each\n\n")
        for k in sorted(vals):
            f.write(f"{k:28s} {vals[k]:16.1f}\n")
        f.write(f"\nHBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB = {hbm / 1e9:.3f} GB (FETCH_SIZE doubled: gfx950 reports half of wide coalesced reads)\n")
        f.write(f"algorithmic bytes per launch = {alg / 1e9:.3f} GB -> traffic / algorithmic = {hbm / alg:.3f}\n")
        if "GRBM_GUI_ACTIVE" in vals and "SQ_INSTS_VALU" in vals:
            clk = vals["GRBM_GUI_ACTIVE"] / 8.0 / (k_ms * 1e-3) / 1e9
            f.write(f"effective clock = GRBM_GUI_ACTIVE / 8 / kernel time = {clk:.2f} GHz; VALU issue = "
                    f"{vals['SQ_INSTS_VALU'] / (k_ms * 1e-3 * clk * 1e9 * 256):.2f} wave-instr/clk/CU\n")
    try:
        traffic_all = json.load(open(os.path.join(OUT, "traffic.json")))
    except (OSError, ValueError):
        traffic_all = {}
    traffic = {workload: {"kernel": kernel, "FETCH_SIZE_KB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KB_raw": vals["WRITE_SIZE"],
                          "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads), WRITE_SIZE as is; separate --pmc passes",
                          "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
                          "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-images 0 --latency-requests 0",
                          "round": int(tag[1:])}}
    # the resample kernel's traffic does not depend on what follows it: the resize-only workload string gets the same record
    traffic_all.update(traffic)
    base = workload.split(" + baseline JPEG encode")[0]
    if base != workload:
        traffic_all[base] = traffic[workload]
    json.dump(traffic_all, open(os.path.join(OUT, "traffic.json"), "w"), indent=1)
    for p in glob.glob(os.path.join(RAW, "bench_*.json")):
        line = open(p).read().strip().splitlines()[-1]
        json.loads(line)
        open(os.path.join(OUT, f"{tag}_" + os.path.basename(p)), "w").write(line + "\n")
    with open(os.path.join(OUT, f"{tag}_commands.txt"), "w") as f:
        f.write("""How the files of this round were produced (tools/collect_profiles.sh on a 1 x MI355X box, then tools/summarize_profiles.py):

(B = python3 bench.py --cpu-images 0 --latency-requests 0 --extra-steps 0 --verify-images 0; the default workload is resize + letterbox + JPEG encode)
{tag}_config1_kernel_stats.csv              rocprofv3 --kernel-trace --stats --output-format csv -- B --steps 20 --warmup 3
                                            (24 launches of each kernel: 1 planning run + 3 warm-up + 20 timed, so the average includes the cold first launch)
{tag}_config2_kernel_stats.csv              ... -- B --steps 5 --warmup 2 --blur 10 --grayscale --frontend none
{tag}_config1_resize_only_kernel_stats.csv  ... -- B --steps 5 --warmup 2 --frontend none
{tag}_roctx_marker_stats.csv                FLGPU_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats -- B --steps 3 --warmup 1
{tag}_config1_pmc.txt, traffic.json         rocprofv3 --kernel-trace --pmc <one counter group per pass> --output-format csv -- B --steps 2 --warmup 1
{tag}_bench_*.json                          python3 bench.py [--frontend none | --crop | --blur 10 --grayscale | --frontend jfif444|webp420] (config1: plain defaults with
                                            verification, extras, the three latency probes and the CPU baseline)
(only rows of this repository's kernels, fl::*, are kept in the kernel statistics)
""".replace("{tag}", tag))
        # files other tools wrote for this round: listed only when they exist, so the list never names a file the round does not have
        others = [
            ("latency_sweep.txt", 'tools/experiments/latency_sweep.sh + tools/latency/latency_probe <threads> 4096 1920 1080 "w=300&h=200"'),
            ("microbench_valu_rate.txt", "tools/microbench/valu_rate.hip"),
            ("resample_sweep.txt", "tools/experiments/resample_sweep.py 1024 (matrix-pipe vs streaming kernel over target sizes)"),
            ("generic_sweep.txt", "tools/experiments/generic_sweep.py (requests neither fused kernel takes: tiled two-pass kernel vs the form through HBM)"),
            ("mfma_model_rate.txt", "tools/experiments/mfma_model_rate.py (device vs the numpy model of the matrix-pipe arithmetic vs the reference arithmetic)"),
            ("jpeg_pmc_before.txt", "tools/collect_jpeg_pmc.sh + tools/summarize_jpeg_pmc.py, encoder as round 2 left it"),
            ("jpeg_pmc.txt", "tools/collect_jpeg_pmc.sh + tools/summarize_jpeg_pmc.py, encoder of this round"),
            ("kernel_experiments.txt", "tools/experiments/ab_inproc.py over the ablation builds of tools/build_ablate.sh (same buffers, same process)"),
            ("power_probe.txt", "tools/experiments/power_probe.sh (rocm-smi power / clock samples while one kernel runs back to back)"),
            ("store_probe.txt", "tools/microbench/store_probe.hip"),
            ("halfblock_probe.txt", "tools/microbench/halfblock_probe.hip"),
            ("placement_probes.txt", "tools/experiments/placement_probe2.py ... placement_probe5.py"),
            ("xcd_and_placement_counters.txt", "tools/experiments/wgtime_runs.sh (-DFL_MFMA_TIMING build) and tools/experiments/placement_pmc.sh"),
            ("config2_pmc.txt", "tools/experiments/config2_pmc.sh"),
            ("tile_kernel_pmc.txt", "tools/experiments/tile_pmc.sh 2000 1000 128"),
            ("jpeg_decoder_ab.txt", "tools/experiments/jpeg_source_rate.py with the library swapped between runs"),
            ("config4_mixed.txt", "tools/experiments/config4_mixed.py"),
            ("latency_jpeg_sources.txt", "tools/experiments/jpeg_source_rate.py + tools/microbench/jpegdec/decode_bench.cpp"),
            ("generic_sweep_notile.txt", "FLGPU_NO_TILE=1 tools/experiments/generic_sweep.py (the two-kernel form through HBM)"),
        ]
        for name, how in others:
            if os.path.exists(os.path.join(OUT, f"{tag}_{name}")):
                f.write(f"{tag}_{name:34s} {how}\n")
        for p in sorted(glob.glob(os.path.join(OUT, f"{tag}_microbench_*_probe.txt"))):
            f.write(f"{os.path.basename(p):38s} tools/microbench/{os.path.basename(p)[len(tag) + 12:-4]}.hip\n")
    print("profiles/ updated:", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
