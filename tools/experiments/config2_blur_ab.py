"""Blur of BASELINE config 2 (grey, one channel filtered) and of its colour twin on the vector blur kernel and on the window-tile
matrix-pipe kernel (FLGPU_WTILE_BLUR_ALWAYS=1): where the routing rule `all but the one-channel shortcut` comes from.
   python tools/experiments/config2_blur_ab.py"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fl = importlib.import_module("fanlin-rs_amd")
n, H, W, C = 1024, 1080, 1920, 3
src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for env in ({}, {"FLGPU_WTILE_BLUR_ALWAYS": "1"}):
    os.environ.pop("FLGPU_WTILE_BLUR_ALWAYS", None); os.environ.update(env)
    for kw in (dict(w=300, h=200, blur_sigma=10.0, grayscale=True), dict(w=300, h=200, blur_sigma=10.0)):
        with fl.State(device=0, profile=True) as st:
            p = fl.make_params(**kw)
            plan = fl.plan_output(p, W, H, C)
            stride = (int(plan.out_bytes) + 255) // 256 * 256
            dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
            run = st.prepared_batch([src.data_ptr() + i * H * W * C for i in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
            for _ in range(3): run(stream)
            torch.cuda.synchronize(); st.reset_stats()
            for _ in range(10): run(stream)
            torch.cuda.synchronize()
            s = st.stats()
            print(env, kw, "resample %.3f blur %.3f ms; wtile launches %d" % (s["resample_ms"] / 10, s["blur_ms"] / 10, s["wtile_launches"]), flush=True)
