"""Request-space sweep of the paths beside the flagship (query.rs:20-21 allows w 20..2000, h 20..1000): mild down-scales, up-scales
and the large blur -- which kernel serves each, the step time (wall clock between device synchronisations, whole batch call) and the algorithmic
bytes (source + destination) over it as a fraction of the 8 TB/s HBM peak.   python tools/experiments/generic_sweep.py"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fl = importlib.import_module("fanlin-rs_amd")
stream = torch.cuda.current_stream().cuda_stream
CASES = [
    # name, n, source (h, w, c), request kwargs
    ("1080p -> w=300&h=200 (flagship)", 1024, (1080, 1920, 3), dict(w=300, h=200)),
    ("1080p -> w=1000&h=562 (ratio 1.92)", 256, (1080, 1920, 3), dict(w=1000, h=562)),
    ("1080p -> w=2000&h=1000 (mild up-scale to the largest allowed target)", 128, (1080, 1920, 3), dict(w=2000, h=1000)),
    ("1080p -> w=640&h=360 (ratio 3)", 512, (1080, 1920, 3), dict(w=640, h=360)),
    ("160x120 -> w=300&h=200 (up-scale: config 4's thumbnails)", 8192, (120, 160, 3), dict(w=300, h=200)),
    ("4K -> w=300&h=200 (ratio 12.8)", 256, (2160, 3840, 3), dict(w=300, h=200)),
    ("2000x1000 blur=20 only", 256, (1000, 2000, 3), dict(blur_sigma=20.0)),
    ("1080p -> w=300&h=200 + blur=20", 1024, (1080, 1920, 3), dict(w=300, h=200, blur_sigma=20.0)),
    ("1080p grayscale only (no resize)", 256, (1080, 1920, 3), dict(grayscale=True)),
]
out = []
for name, n, (H, W, C), kw in CASES:
    src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
    with fl.State(device=0, profile=True) as st:
        p = fl.make_params(**kw)
        plan = fl.plan_output(p, W, H, C)
        stride = (int(plan.out_bytes) + 255) // 256 * 256
        dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
        run = st.prepared_batch([src.data_ptr() + i * H * W * C for i in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
        for _ in range(3):
            run(stream)
        torch.cuda.synchronize()
        st.reset_stats()
        steps = 10
        t0 = time.perf_counter()
        for _ in range(steps):
            run(stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        s = st.stats()
        kern = "window-tile matrix" if s.get("wtile_launches") and s["wtile_launches"] == s["mfma_launches"] else "matrix-pipe (+ window-tile blur)" if s.get("wtile_launches") else "matrix-pipe" if s["mfma_launches"] else ("generic two-pass" if s["generic_launches"] else ("streaming" if s["resample_launches"] else "place / blur only"))
        alg = n * (H * W * C + int(plan.pixel_bytes))
        line = f"{name:72s} n={n:5d}  {kern:32s} {ms:8.3f} ms/step  {n / ms:9.1f} k images/s  algorithmic {alg / 1e9:6.3f} GB -> {alg / (ms * 1e-3) / 1e12:5.2f} TB/s = {alg / (ms * 1e-3) / 8e12:5.3f} of peak" \
               f"  [resample {s['resample_ms'] / steps:.3f} ms, blur {s['blur_ms'] / steps:.3f} ms]"
    print(line, flush=True)
    out.append(line)
    del src, dst
    torch.cuda.empty_cache()
if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
    open(os.path.join(ROOT, "gpurun_out", "generic_sweep.txt"), "w").write("\n".join(out) + "\n")
