"""Development check of the window-tile matrix-pipe kernel (csrc/fl_wtile.h) on the GPU box:
   python tools/experiments/wt_check.py [--no-time]
  1. geometries the kernel serves (mild down-scales, up-scales, letterboxed thumbnails, blurs) against the oracle's reference
     arithmetic: max difference and off-by-one rate, beside the f32 vector kernels (FLGPU_NO_WTILE=1);
  2. the request-space sweep's cases timed both ways in one process."""
import argparse
import os
import sys
import time

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
ap.add_argument("--only", type=int, default=-1)
a = ap.parse_args()


def set_mode(m, st=None):
    """The window-tile kernel on or off: through the context's switch when there is one, else through the variable flgpu_create reads."""
    os.environ.pop("FLGPU_NO_WTILE", None)
    if m == "vector":
        os.environ["FLGPU_NO_WTILE"] = "1"
    if st is not None:
        st.debug_set("no_wtile", int(m == "vector"))


bad = 0
if not a.no_check:
    oracle = oracle_lib.load()
    rng = np.random.default_rng(11)
    # (h, w, c) -> request
    cases = [((120, 160, 3), dict(w=300, h=200)),
             ((540, 960, 3), dict(w=600, h=400)),
             ((1080, 1920, 3), dict(w=1000, h=562)),
             ((200, 320, 3), dict(w=640, h=400)),
             ((300, 400, 1), dict(w=500, h=333)),
             ((300, 400, 4), dict(w=333, h=250)),
             ((301, 403, 3), dict(w=350, h=260, crop=True)),
             ((200, 300, 3), dict(blur_sigma=10.0)),
             ((256, 384, 3), dict(blur_sigma=20.0)),
             ((250, 330, 4), dict(blur_sigma=3.0)),
             ((1080, 1920, 3), dict(w=300, h=200, blur_sigma=8.0))]
    with fl.State() as st:
        for idx, (shape, kw) in enumerate(cases):
            if a.only >= 0 and idx != a.only:
                continue
            img = rng.integers(0, 256, size=shape, dtype=np.uint8)
            p = fl.make_params(**kw)
            want = oracle.process_pixels(img, kw.get("w"), kw.get("h"), crop=kw.get("crop", False), blur_sigma=kw.get("blur_sigma", 0.0))
            line = f"{shape[1]}x{shape[0]}x{shape[2]} {kw}:"
            for m in ("wtile", "vector"):
                set_mode(m, st)
                before = st.stats()["wtile_launches"]
                got = st.process_pixels(img, p)
                used = st.stats()["wtile_launches"] - before
                d = np.abs(got.astype(np.int32) - want.astype(np.int32))
                again = st.process_pixels(img, p)
                line += f"  {m}({used} launches): max {d.max()} off-by-one {1e6 * (d > 0).mean():.0f} ppm{'' if np.array_equal(got, again) else ' NOT REPEATABLE'}"
                if d.max() > 1:
                    bad += 1
                    nz = np.nonzero(d > 1)
                    line += f" BAD {int((d > 1).sum())} bytes, rows {sorted(set(nz[0].tolist()))[:10]} cols {sorted(set(nz[1].tolist()))[:10]}"
            print(line, flush=True)
    set_mode("wtile")

if not a.no_time:
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    CASES = [("1080p -> w=1000&h=562", 256, (1080, 1920, 3), dict(w=1000, h=562)),
             ("1080p -> w=2000&h=1000", 128, (1080, 1920, 3), dict(w=2000, h=1000)),
             ("160x120 -> w=300&h=200", 8192, (120, 160, 3), dict(w=300, h=200)),
             ("2000x1000 blur=20", 256, (1000, 2000, 3), dict(blur_sigma=20.0)),
             ("1080p -> w=300&h=200 + blur=20", 1024, (1080, 1920, 3), dict(w=300, h=200, blur_sigma=20.0)),
             ("1080p -> w=300&h=200 gray + blur=10 (config 2)", 1024, (1080, 1920, 3), dict(w=300, h=200, blur_sigma=10.0, grayscale=True))]
    for name, n, (H, W, C), kw in CASES:
        src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
        line = f"{name:48s} n={n:5d}"
        for m in ("wtile", "vector"):
            set_mode(m)
            with fl.State(device=0, profile=True) as st:
                p = fl.make_params(**kw)
                plan = fl.plan_output(p, W, H, C)
                stride = (int(plan.out_bytes) + 255) // 256 * 256
                dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
                run = st.prepared_batch([src.data_ptr() + i * H * W * C for i in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
                for _ in range(2):
                    run(stream)
                torch.cuda.synchronize()
                st.reset_stats()
                steps = 5
                t0 = time.perf_counter()
                for _ in range(steps):
                    run(stream)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3 / steps
                s = st.stats()
                alg = n * (H * W * C + int(plan.pixel_bytes))
                line += f"   {m}: {ms:7.3f} ms ({alg / (ms * 1e-3) / 8e12:5.3f} of peak; resample {s['resample_ms'] / steps:.3f} blur {s['blur_ms'] / steps:.3f}; wtile launches {s['wtile_launches']})"
            del dst
        print(line, flush=True)
        del src
        torch.cuda.empty_cache()
    set_mode("wtile")
sys.exit(1 if bad else 0)
