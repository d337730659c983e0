"""One fresh process per run: the flagship batch with its source inside an allocation of SIZE MiB (0 = exactly the batch) and its
destination inside one of DSIZE MiB (0 = exactly the results).   python tools/experiments/placement_probe4.py SIZE DSIZE [reps]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fl = importlib.import_module("fanlin-rs_amd")
n, H, W, C = 1024, 1080, 1920, 3
size, dsize = int(sys.argv[1]) << 20, int(sys.argv[2]) << 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
stream = torch.cuda.current_stream().cuda_stream
with fl.State(device=0, profile=True) as st:
    p = fl.make_params(300, 200)
    plan = fl.plan_output(p, W, H, C)
    stride = (int(plan.out_bytes) + 255) // 256 * 256
    src = torch.empty(max(size, n * H * W * C), dtype=torch.uint8, device="cuda")
    src[: n * H * W * C].random_(0, 256)
    dst = torch.zeros(max(dsize, n * stride), dtype=torch.uint8, device="cuda")
    run = st.prepared_batch([src.data_ptr() + k * H * W * C for k in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + k * stride for k in range(n)], [stride] * n)
    ts = []
    for _ in range(reps):
        st.reset_stats()
        for _ in range(60):
            run(stream)
        torch.cuda.synchronize()
        s = st.stats()
        ts.append(s["resample_ms"] / max(s["resample_launches"], 1))
    print(f"src alloc {sys.argv[1]:>6} MiB at {src.data_ptr():#x}, dst alloc {sys.argv[2]:>5} MiB: " + " ".join(f"{x:.4f}" for x in ts), flush=True)
