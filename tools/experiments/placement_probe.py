"""Does the resample kernel's time depend on WHERE its buffers are?  One process, one library, the batch allocated again and
again (a few spacer allocations of random size in between): if the level moves with the allocation and not with time, the
5 % process-to-process spread of this kernel is memory placement, not thermal state.   python tools/experiments/placement_probe.py [lib.so]"""
import importlib
import os
import sys
import random
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    os.environ["FLGPU_LIB"] = os.path.abspath(sys.argv[1])
fl = importlib.import_module("fanlin-rs_amd")
n, H, W, C = 1024, 1080, 1920, 3
stream = torch.cuda.current_stream().cuda_stream
random.seed(1)
t00 = time.time()
with fl.State(device=0, profile=True) as st:
    p = fl.make_params(300, 200)
    plan = fl.plan_output(p, W, H, C)
    stride = (int(plan.out_bytes) + 255) // 256 * 256
    spacers = []
    for trial in range(8):
        src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
        dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
        run = st.prepared_batch([src.data_ptr() + k * H * W * C for k in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + k * stride for k in range(n)], [stride] * n)
        ts = []
        for rep in range(4):
            st.reset_stats()
            for _ in range(60):
                run(stream)
            torch.cuda.synchronize()
            s = st.stats()
            ts.append(s["resample_ms"] / max(s["resample_launches"], 1))
        print(f"t={time.time()-t00:5.1f}s trial {trial}: src {src.data_ptr():#x} (mod 2 MiB {src.data_ptr() % (2<<20):#x}) dst {dst.data_ptr():#x}: " + " ".join(f"{x:.4f}" for x in ts), flush=True)
        del run, src, dst
        torch.cuda.empty_cache()
        spacers.append(torch.empty(random.randrange(1, 64) << 20, dtype=torch.uint8, device="cuda"))  # shifts the next allocation
