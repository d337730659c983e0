"""Which buffer's placement moves the resample kernel's time?  One process, one context, the flagship batch.
  A  source fixed, destination allocated anew (spacers in between)      B  destination fixed, source allocated anew
  C  both fixed, the destination pointers shifted inside one pool       D  both fixed, the per-image destination stride varied
python tools/experiments/placement_probe2.py"""
import importlib
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fl = importlib.import_module("fanlin-rs_amd")
n, H, W, C = 1024, 1080, 1920, 3
stream = torch.cuda.current_stream().cuda_stream
random.seed(2)


def timed(st, run, reps=3, launches=60):
    ts = []
    for _ in range(reps):
        st.reset_stats()
        for _ in range(launches):
            run(stream)
        torch.cuda.synchronize()
        s = st.stats()
        ts.append(s["resample_ms"] / max(s["resample_launches"], 1))
    return " ".join(f"{x:.4f}" for x in ts)


with fl.State(device=0, profile=True) as st:
    p = fl.make_params(300, 200)
    plan = fl.plan_output(p, W, H, C)
    stride0 = (int(plan.out_bytes) + 255) // 256 * 256

    def prepared(src, dst_ptr, stride):
        return st.prepared_batch([src.data_ptr() + k * H * W * C for k in range(n)], [(H, W, C)] * n, p, [dst_ptr + k * stride for k in range(n)], [stride] * n)

    src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
    pool = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
    base = (pool.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    print(f"src {src.data_ptr():#x} pool {pool.data_ptr():#x} stride {stride0}", flush=True)
    print("C: destination shifted inside the pool", flush=True)
    for shift in (0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 64 << 20, (64 << 20) + 65536, 300 << 20):
        print(f"  shift {shift:>10}: {timed(st, prepared(src, base + shift, stride0))}", flush=True)
    print("D: per-image destination stride", flush=True)
    for stride in (stride0, stride0 + 256, stride0 + 4096, 245760, 262144, 262144 + 256, 300000 // 256 * 256, 524288):
        print(f"  stride {stride:>8}: {timed(st, prepared(src, base, stride))}", flush=True)
    print("A: destination allocated anew", flush=True)
    spacers = []
    for trial in range(6):
        dst = torch.zeros((n, stride0), dtype=torch.uint8, device="cuda")
        print(f"  dst {dst.data_ptr():#x}: {timed(st, prepared(src, dst.data_ptr(), stride0))}", flush=True)
        del dst
        torch.cuda.empty_cache()
        spacers.append(torch.empty(random.randrange(1, 64) << 20, dtype=torch.uint8, device="cuda"))
    print("B: source allocated anew (destination = the pool)", flush=True)
    del src
    torch.cuda.empty_cache()
    for trial in range(5):
        src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
        print(f"  src {src.data_ptr():#x}: {timed(st, prepared(src, base, stride0))}", flush=True)
        del src
        torch.cuda.empty_cache()
        spacers.append(torch.empty(random.randrange(1, 64) << 20, dtype=torch.uint8, device="cuda"))
