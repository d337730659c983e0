import importlib, os, sys, torch
sys.path.insert(0, '/root/repo')
fl = importlib.import_module("fanlin-rs_amd")
n=1024; H,W,C=1080,1920,3
src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for (w,h) in [(256,144),(200,200),(96,54),(512,288),(640,360),(1000,562)]:
    line=f"w={w}&h={h}:"
    with fl.State(device=0, profile=True) as st:
        p = fl.make_params(w, h)
        plan = fl.plan_output(p, W, H, C)
        stride = (int(plan.out_bytes) + 255) // 256 * 256
        dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
        run = st.prepared_batch([src.data_ptr() + i * H * W * C for i in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
        for _ in range(3): run(stream)
        torch.cuda.synchronize(); st.reset_stats()
        for _ in range(20): run(stream)
        torch.cuda.synchronize(); s = st.stats()
        ms = s["resample_ms"] / max(s["resample_launches"], 1)
        gbs = (s["resample_src_bytes"] + s["resample_dst_bytes"]) / max(s["resample_launches"], 1) / (ms * 1e-3) / 1e9
        line += f"  {'mfma' if s['mfma_launches'] else ('generic' if s['generic_launches'] else 'stream')} {ms:.3f} ms ({gbs / 8000:.3f} of peak)"
    print(line, flush=True)
