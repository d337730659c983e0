"""The flagship step (resize + letterbox + JPEG encode, 1024 x 1080p) on a context WITHOUT stage profiling, for a kernel trace:
are the gaps between the kernels the profiling events' or the dispatch's?   rocprofv3 --kernel-trace ... -- python3 tools/experiments/gaps_noprofile.py [0|1]"""
import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fl = importlib.import_module("fanlin-rs_amd")
n, H, W, C = 1024, 1080, 1920, 3
prof = len(sys.argv) > 1 and sys.argv[1] == "1"
src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
with fl.State(device=0, profile=prof) as st:
    p = fl.make_params(300, 200, front_end=fl.FE_JPEG, quality=75)
    plan = fl.plan_output(p, W, H, C)
    stride = (int(plan.max_out_bytes) + 255) // 256 * 256
    dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
    run = st.prepared_batch([src.data_ptr() + i * H * W * C for i in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
    for _ in range(40):
        run(0)
    torch.cuda.synchronize()
