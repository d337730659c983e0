"""A/B of library builds inside ONE process, on the SAME source and destination buffers: run-to-run differences of this kernel
(1.40 - 1.54 ms for identical code, process to process) come from where a process's buffers land in physical memory, so
builds are compared on one allocation, interleaved.
    python tools/experiments/ab_inproc.py a.so b.so ... [--rounds 8] [--launches 40] [--w 300 --h 200] [--crop] [--gray] [--channels 3]"""
import argparse
import importlib.util
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_binding(lib_path, tag):
    os.environ["FLGPU_LIB"] = os.path.abspath(lib_path)
    pkg = os.path.join(ROOT, "fanlin-rs_amd")
    spec = importlib.util.spec_from_file_location("fl_" + tag, os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["fl_" + tag] = mod
    spec.loader.exec_module(mod)
    mod.load_library()
    return mod


ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--launches", type=int, default=40)
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--w", type=int, default=300)
ap.add_argument("--h", type=int, default=200)
ap.add_argument("--crop", action="store_true")
ap.add_argument("--gray", action="store_true")
ap.add_argument("--channels", type=int, default=3)
a = ap.parse_args()
n, H, W, C = a.n, 1080, 1920, a.channels
src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
runs = []
dst = None
for i, lib in enumerate(a.libs):
    fl = load_binding(lib, str(i))
    st = fl.State(device=0, profile=True)
    st.__enter__()
    p = fl.make_params(a.w, a.h, crop=a.crop, grayscale=a.gray)
    plan = fl.plan_output(p, W, H, C)
    stride = (int(plan.out_bytes) + 255) // 256 * 256
    if dst is None:
        dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
    run = st.prepared_batch([src.data_ptr() + k * H * W * C for k in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + k * stride for k in range(n)], [stride] * n)
    runs.append((os.path.basename(lib), st, run))
times = {name: [] for name, _, _ in runs}
for name, st, run in runs:
    for _ in range(3):
        run(stream)
torch.cuda.synchronize()
for r in range(a.rounds):
    for name, st, run in runs:
        st.reset_stats()
        for _ in range(a.launches):
            run(stream)
        torch.cuda.synchronize()
        s = st.stats()
        times[name].append(s["resample_ms"] / max(s["resample_launches"], 1))
for name, _, _ in runs:
    t = times[name]
    print(f"{name:44s} median {statistics.median(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}   ({' '.join(f'{x:.3f}' for x in t)})", flush=True)
