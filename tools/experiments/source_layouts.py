"""Experiment: streaming-kernel time for the other source layouts (Luma8, LumaA8, Rgba8, unaligned Rgb8 rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import load_package
fl = load_package()
dev = torch.device("cuda", 0)
n = 512
for (H, W, C, kw) in ((1080, 1920, 3, {}), (1080, 1920, 4, {}), (1080, 1920, 1, {}), (1080, 1920, 2, {}), (1080, 1919, 3, {}),
                      (1080, 1920, 4, dict(grayscale=True)), (1080, 1920, 3, dict(inverse=True)), (2160, 3840, 3, {})):
    if H == 2160: n = 128
    params = fl.make_params(300, 200, **kw)
    plan = fl.plan_output(params, W, H, C)
    stride = (int(plan.out_bytes) + 255) // 256 * 256
    src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device=dev)
    dst = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    st = fl.State(device=0, profile=True)
    run = st.prepared_batch([src.data_ptr() + i * H * W * C for i in range(n)], [(H, W, C)] * n, params,
                            [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3): run(s)
    torch.cuda.synchronize(); st.reset_stats()
    for _ in range(10): run(s)
    torch.cuda.synchronize()
    stt = st.stats()
    ms = stt["resample_ms"] / 10
    alg = n * (H * W * C + plan.out_bytes)
    print(f"{W}x{H}x{C} {kw}: kernel {ms:.3f} ms / {n} images, {alg / ms / 1e9:.2f} TB/s = {alg / ms / 8e9 * 100:.1f}% of HBM peak, stream launches {stt['resample_launches']}, generic {stt['generic_launches']}")
    st.close(); del src, dst
