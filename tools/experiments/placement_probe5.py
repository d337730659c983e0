"""Inside ONE source allocation: does the distance between consecutive pictures (the batch's image stride) move the resample kernel's
level?  (Workgroups that run at the same time read the same row of ~85 consecutive pictures.)   python tools/experiments/placement_probe5.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fl = importlib.import_module("fanlin-rs_amd")
n, H, W, C = 1024, 1080, 1920, 3
img = H * W * C
stream = torch.cuda.current_stream().cuda_stream
with fl.State(device=0, profile=True) as st:
    p = fl.make_params(300, 200)
    plan = fl.plan_output(p, W, H, C)
    dstride = (int(plan.out_bytes) + 255) // 256 * 256
    pool = torch.empty(9 << 30, dtype=torch.uint8, device="cuda")
    pool[: 8 << 30].random_(0, 256)
    dst = torch.zeros(n * dstride, dtype=torch.uint8, device="cuda")
    print(f"pool at {pool.data_ptr():#x}", flush=True)
    for rnd in range(2):
        for stride in (img, img + 256, img + 1024, img + 4096, img + 4096 + 256, img + 65536, img + 70912, 6 << 20, (6 << 20) + 256, (6 << 20) + 4352, 8 << 20, (8 << 20) - 4096 - 256):
            if stride * (n - 1) + img > (9 << 30):
                continue
            run = st.prepared_batch([pool.data_ptr() + k * stride for k in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + k * dstride for k in range(n)], [dstride] * n)
            ts = []
            for _ in range(3):
                st.reset_stats()
                for _ in range(60):
                    run(stream)
                torch.cuda.synchronize()
                s = st.stats()
                ts.append(s["resample_ms"] / max(s["resample_launches"], 1))
            print(f"  picture stride {stride:>9} (= picture + {stride - img:>8}): " + " ".join(f"{x:.4f}" for x in ts), flush=True)
