"""Does the SIZE of the allocation a buffer lives in decide the placement mode?  (tools/experiments/placement_probe2.py: the same
batch runs in 1.39 ms with the destination in a 1 GiB pool and in 1.43 / 1.51 ms, alternating, in fresh 245 MB allocations.)
python tools/experiments/placement_probe3.py"""
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
random.seed(3)


def timed(st, run, reps=4, launches=60):
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
    stride = (int(plan.out_bytes) + 255) // 256 * 256

    def prepared(src_ptr, dst_ptr):
        return st.prepared_batch([src_ptr + k * H * W * C for k in range(n)], [(H, W, C)] * n, p, [dst_ptr + k * stride for k in range(n)], [stride] * n)

    spacers = []
    src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
    print("E: destination in fresh allocations of different sizes (source fixed)", flush=True)
    for size in (n * stride, 256 << 20, 512 << 20, 1 << 30, 2 << 30):
        for trial in range(4):
            dst = torch.zeros(size, dtype=torch.uint8, device="cuda")
            print(f"  dst alloc {size >> 20:5d} MiB at {dst.data_ptr():#x}: {timed(st, prepared(src.data_ptr(), dst.data_ptr()))}", flush=True)
            del dst
            torch.cuda.empty_cache()
            spacers.append(torch.empty(random.randrange(1, 64) << 20, dtype=torch.uint8, device="cuda"))
    del src
    torch.cuda.empty_cache()
    pool = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
    print("F: source in fresh allocations of different sizes (destination: one 1 GiB pool)", flush=True)
    for size in (n * H * W * C, 8 << 30, 16 << 30):
        for trial in range(4):
            src = torch.empty(size, dtype=torch.uint8, device="cuda")
            src[: n * H * W * C].random_(0, 256)
            print(f"  src alloc {size >> 20:6d} MiB at {src.data_ptr():#x}: {timed(st, prepared(src.data_ptr(), pool.data_ptr()))}", flush=True)
            del src
            torch.cuda.empty_cache()
            spacers.append(torch.empty(random.randrange(1, 64) << 20, dtype=torch.uint8, device="cuda"))
