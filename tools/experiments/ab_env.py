"""A/B of library builds (and of FLGPU_* switches) inside one process on the same buffers, for the flagship batch:
    python tools/experiments/ab_env.py lib1.so[:ENV=VAL,...] lib2.so ... [--rounds 5] [--launches 30] [--channels 3] [--w 300 --h 200]
A bare name 'default' means the library in the tree.  The library reads the environment only in flgpu_create (round 5), so a variant's
FLGPU_* values are in the environment while ITS context is created and nowhere else."""
import argparse, importlib.util, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--launches", type=int, default=30)
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--w", type=int, default=300)
ap.add_argument("--h", type=int, default=200)
ap.add_argument("--crop", action="store_true")
ap.add_argument("--gray", action="store_true")
ap.add_argument("--channels", type=int, default=3)
ap.add_argument("--srcw", type=int, default=1920)
ap.add_argument("--srch", type=int, default=1080)
ap.add_argument("--blur", type=float, default=0.0)
ap.add_argument("--stat", default="", help="resample | blur | both: which stage time to print (default: blur for a blur-only request, else resample)")
a = ap.parse_args()
n, H, W, C = a.n, a.srch, a.srcw, a.channels
src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
runs, dst = [], None
for i, v in enumerate(a.variants):
    lib, _, envs = v.partition(":")
    env = dict(e.split("=", 1) for e in envs.split(",") if e)
    path = os.path.join(ROOT, "fanlin-rs_amd", "libfanlin_gpu.so") if lib == "default" else os.path.abspath(lib)
    os.environ["FLGPU_LIB"] = path
    pkg = os.path.join(ROOT, "fanlin-rs_amd")
    spec = importlib.util.spec_from_file_location("fl_%d" % i, os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
    fl = importlib.util.module_from_spec(spec); sys.modules["fl_%d" % i] = fl; spec.loader.exec_module(fl); fl.load_library()
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    st = fl.State(device=0, profile=True); st.__enter__()
    for k, v in old.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    p = fl.make_params(a.w, a.h, crop=a.crop, grayscale=a.gray, blur_sigma=a.blur) if a.w else fl.make_params(blur_sigma=a.blur)
    plan = fl.plan_output(p, W, H, C)
    stride = (int(plan.out_bytes) + 255) // 256 * 256
    if dst is None:
        dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
    run = st.prepared_batch([src.data_ptr() + k * H * W * C for k in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + k * stride for k in range(n)], [stride] * n)
    runs.append((v, st, run, env))
KEYS = ("FLGPU_NO_SMALL", "FLGPU_MFMA_ARITH", "FLGPU_NO_MFMA", "FLGPU_FORCE_BANDS", "FLGPU_NO_WTILE", "FLGPU_WTILE_ALWAYS", "FLGPU_WTILE_FIRST", "FLGPU_WTILE_NO_OVERLAP")
def setenv(env):   # (libraries of earlier rounds read these per batch)
    for k in KEYS: os.environ.pop(k, None)
    os.environ.update(env)
times = {v: [] for v, _, _, _ in runs}
for v, st, run, env in runs:
    setenv(env)
    for _ in range(3): run(stream)
torch.cuda.synchronize()
for r in range(a.rounds):
    for v, st, run, env in runs:
        setenv(env)
        st.reset_stats()
        for _ in range(a.launches): run(stream)
        torch.cuda.synchronize()
        s = st.stats()
        tb, tr = s["blur_ms"] / max(s["blur_launches"], 1), s["resample_ms"] / max(s["resample_launches"], 1)
        which = a.stat or ("blur" if a.blur > 0 and not a.w else "resample")
        times[v].append(tb if which == "blur" else tr if which == "resample" else tb + tr)
for v, _, _, _ in runs:
    t = times[v]
    print(f"{os.path.basename(v):56s} median {statistics.median(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}   ({' '.join(f'{x:.3f}' for x in t)})", flush=True)
