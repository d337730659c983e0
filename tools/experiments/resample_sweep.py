"""Kernel time of the resample stage for a list of target sizes on a 1920x1080 Rgb8 batch (device resident), with the
matrix-pipe kernel and with the streaming kernel (FLGPU_NO_MFMA=1): which geometries gain, which plans keep their
horizontal operands in LDS.   python tools/experiments/resample_sweep.py [n_images]"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
fl = importlib.import_module("fanlin-rs_amd")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H, W, C = 1080, 1920, 3
src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
targets = [(300, 200, False), (300, 200, True), (317, 200, False), (320, 240, False), (333, 222, True), (400, 300, False), (480, 270, False),
           (256, 144, False), (200, 200, False), (150, 100, False), (96, 54, False), (512, 288, False)]
stream = torch.cuda.current_stream().cuda_stream
for (w, h, crop) in targets:
    line = f"w={w}&h={h}{'&crop' if crop else ''}:"
    for no_mfma in ("0", "1"):
        os.environ["FLGPU_NO_MFMA"] = no_mfma
        with fl.State(device=0, profile=True) as st:
            p = fl.make_params(w, h, crop=crop)
            plan = fl.plan_output(p, W, H, C)
            stride = (int(plan.out_bytes) + 255) // 256 * 256
            dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
            run = st.prepared_batch([src.data_ptr() + i * H * W * C for i in range(n)], [(H, W, C)] * n, p,
                                    [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
            for _ in range(3):
                run(stream)
            torch.cuda.synchronize()
            st.reset_stats()
            for _ in range(20):
                run(stream)
            torch.cuda.synchronize()
            s = st.stats()
            ms = s["resample_ms"] / max(s["resample_launches"], 1)
            gbs = (s["resample_src_bytes"] + s["resample_dst_bytes"]) / max(s["resample_launches"], 1) / (ms * 1e-3) / 1e9
            line += f"  {'mfma' if s['mfma_launches'] else 'stream'} {ms:.3f} ms ({gbs / 8000:.3f} of peak)"
    print(line, flush=True)
    open(os.path.join(ROOT, "gpurun_out", "resample_sweep.txt"), "a").write(line + "\n") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None
