"""Development check of the matrix-pipe kernel's two arithmetics (csrc/fl_mfma.h: full width = default, packed =
FLGPU_MFMA_ARITH=packed) and the streaming kernel (FLGPU_NO_MFMA=1) on the GPU box:
   python tools/experiments/fw_check.py [--no-time] [--n 1024]
  1. a few geometries against the oracle's reference arithmetic: max difference and the rate of off-by-one bytes per kernel;
  2. the flagship batch (n x 1080p Rgb8 -> w=300&h=200) timed in ONE process on the same buffers, the three kernels interleaved
     (run-to-run placement moves this kernel by +-5 %, profiles/r03_placement_probes.txt)."""
import argparse
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib

fl = importlib.import_module("fanlin-rs_amd")
import oracle_lib

ap = argparse.ArgumentParser()
ap.add_argument("--no-time", action="store_true")
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--launches", type=int, default=30)
ap.add_argument("--w", type=int, default=300)
ap.add_argument("--h", type=int, default=200)
ap.add_argument("--channels", type=int, default=3)
ap.add_argument("--modes", default="full,packed,stream")
a = ap.parse_args()

MODES = {"full": {}, "packed": {"FLGPU_MFMA_ARITH": "packed"}, "stream": {"FLGPU_NO_MFMA": "1"}}


def set_mode(m):
    for k in ("FLGPU_MFMA_ARITH", "FLGPU_NO_MFMA"):
        os.environ.pop(k, None)
    os.environ.update(MODES[m])


bad = 0
if not a.no_check:
    oracle = oracle_lib.load()
    rng = np.random.default_rng(7)
    cases = [(1080, 1920, 3, 300, 200, False), (1080, 1920, 3, 300, 200, True), (720, 1280, 3, 160, 90, False), (1080, 1920, 3, 300, 169, False),
             (2160, 3840, 3, 640, 360, False), (600, 800, 3, 100, 100, False), (1080, 1920, 3, 480, 270, False), (333, 1024, 3, 90, 30, False),
             (1080, 1920, 1, 300, 200, False), (1080, 1920, 4, 300, 200, False), (1080, 1920, 2, 300, 169, False), (1080, 1920, 3, 640, 360, False)]
    with fl.State() as st:
        for (h, w, c, ow, oh, crop) in cases:
            img = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
            p = fl.make_params(w=ow, h=oh, crop=crop)
            want = oracle.process_pixels(img, ow, oh, crop=crop)
            line = f"{w}x{h}x{c} -> {ow}x{oh} crop={int(crop)}:"
            for m in a.modes.split(","):
                set_mode(m)
                before = st.stats()["mfma_launches"]
                got = st.process_pixels(img, p)
                used = st.stats()["mfma_launches"] > before
                d = np.abs(got.astype(np.int32) - want.astype(np.int32))
                again = st.process_pixels(img, p)
                line += f"  {m}{'' if used or m == 'stream' else '(not mfma)'}: max {d.max()} off-by-one {1e6 * (d > 0).mean():.0f} ppm{'' if np.array_equal(got, again) else ' NOT REPEATABLE'}"
                if d.max() > 1:
                    bad += 1
                    ys, xs, cs = np.nonzero(d > 1)
                    line += f" BAD rows {sorted(set(ys))[:12]} cols {sorted(set(xs))[:12]}"
            print(line, flush=True)
    set_mode("full")

if not a.no_time:
    import torch
    n, H, W, C = a.n, 1080, 1920, a.channels
    src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    with fl.State(device=0, profile=True) as st:
        p = fl.make_params(a.w, a.h)
        plan = fl.plan_output(p, W, H, C)
        stride = (int(plan.out_bytes) + 255) // 256 * 256
        dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
        run = st.prepared_batch([src.data_ptr() + k * H * W * C for k in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + k * stride for k in range(n)], [stride] * n)
        modes = a.modes.split(",")
        times = {m: [] for m in modes}
        for m in modes:
            set_mode(m)
            for _ in range(3):
                run(stream)
        torch.cuda.synchronize()
        for r in range(a.rounds):
            for m in modes:
                set_mode(m)
                st.reset_stats()
                for _ in range(a.launches):
                    run(stream)
                torch.cuda.synchronize()
                st.batch_results()
                s = st.stats()
                times[m].append(s["resample_ms"] / max(s["resample_launches"], 1))
        alg = n * (H * W * C + plan.out_bytes)
        for m in modes:
            t = times[m]
            med = statistics.median(t)
            print(f"{m:8s} median {med:.4f} ms = {alg / med / 1e9:.2f} TB/s = {alg / med / 8e9:.3f} of 8 TB/s   min {min(t):.4f}  max {max(t):.4f}   ({' '.join(f'{x:.3f}' for x in t)})", flush=True)
sys.exit(1 if bad else 0)
