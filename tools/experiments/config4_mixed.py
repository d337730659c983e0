"""BASELINE config 4 on one GPU: a mixed-size batch (3840x2160 / 1920x1080 / 160x120 in ratio 1:6:3, seed-shuffled),
`w=300&h=200&webp=true&quality=85`: resize + letterbox + libwebp-style YUV420 front end, inputs resident in HBM."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import load_package
fl = load_package()
dev = torch.device("cuda", 0)
n = 1000
kinds = [(2160, 3840)] * (n // 10) + [(1080, 1920)] * (6 * n // 10) + [(120, 160)] * (3 * n // 10)
rng = np.random.default_rng(4); rng.shuffle(kinds)
q = fl.Query.parse("w=300&h=200&webp=true&quality=85")
params, fmt = q.to_params(fl.Format.from_accept_header("image/webp"), input_is_jpeg=True)
srcs = {k: torch.randint(0, 256, (sum(1 for x in kinds if x == k), k[0], k[1], 3), dtype=torch.uint8, device=dev) for k in set(kinds)}
idx = {k: 0 for k in srcs}
ptrs, shapes = [], []
for k in kinds:
    ptrs.append(srcs[k].data_ptr() + idx[k] * k[0] * k[1] * 3); idx[k] += 1; shapes.append((k[0], k[1], 3))
plan = fl.plan_output(params, 1920, 1080, 3)
stride = (int(plan.out_bytes) + 255) // 256 * 256
dst = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
st = fl.State(device=0, profile=True)
run = st.prepared_batch(ptrs, shapes, params, [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3): run(s)
torch.cuda.synchronize(); st.reset_stats()
t0 = time.perf_counter()
for _ in range(10): run(s)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
stt = st.stats()
src_bytes = sum(h * w * 3 for h, w in kinds)
print(f"config 4 (1 GPU): {n} mixed images per step, {dt * 1e3:.3f} ms/step = {n / dt:,.0f} images/s; source {src_bytes / 1e9:.2f} GB/step -> {src_bytes / dt / 1e12:.2f} TB/s algorithmic read;",
      f"stream launches/step {stt['resample_launches'] / 10:.0f}, generic {stt['generic_launches'] / 10:.0f}, resample {stt['resample_ms'] / 10:.3f} ms, front end {stt['frontend_ms'] / 10:.3f} ms")
