"""Runs one geometry of the sweep a few times (for rocprofv3 --kernel-trace --stats): python tools/experiments/one_geometry.py W H [n] [src_h src_w] [blur sigma]
(W = 0: a blur-only request)"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fl = importlib.import_module("fanlin-rs_amd")
w, h = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
H, W = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1080, 1920)
C = 3
src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
with fl.State(device=0) as st:
    sigma = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
    p = fl.make_params(w, h, blur_sigma=sigma) if w else fl.make_params(blur_sigma=sigma)
    plan = fl.plan_output(p, W, H, C)
    stride = (int(plan.out_bytes) + 255) // 256 * 256
    dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
    run = st.prepared_batch([src.data_ptr() + i * H * W * C for i in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
    for _ in range(5):
        run(0)
    torch.cuda.synchronize()
