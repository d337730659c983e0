"""Experiment: how much of the streaming kernel's time is memory latency / bandwidth?  Same launch as bench.py
config 1, but every descriptor points at the SAME source image (or at K distinct ones), so reads hit L2 / MALL."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import load_package
fl = load_package()
n, H, W, C = 1024, 1080, 1920, 3
dev = torch.device("cuda", 0)
params = fl.make_params(300, 200)
plan = fl.plan_output(params, W, H, C)
stride = (int(plan.out_bytes) + 255) // 256 * 256
dst = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
for distinct in (1024, 64, 8, 1):
    src = torch.randint(0, 256, (distinct, H, W, C), dtype=torch.uint8, device=dev)
    st = fl.State(device=0, profile=True)
    run = st.prepared_batch([src.data_ptr() + (i % distinct) * H * W * C for i in range(n)], [(H, W, C)] * n, params,
                            [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3): run(s)
    torch.cuda.synchronize(); st.reset_stats()
    for _ in range(10): run(s)
    torch.cuda.synchronize()
    print(distinct, "distinct sources: kernel ms", round(st.stats()["resample_ms"] / 10, 3))
    st.close(); del src
