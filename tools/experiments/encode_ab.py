"""A/B of library builds on the encoder stage of config 1 (1024 x 1080p -> 300x200 -> JPEG q 75), one process:
   python tools/experiments/encode_ab.py lib1.so lib2.so ...   ('default' = the library in the tree)"""
import importlib.util, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n, H, W, C = 1024, 1080, 1920, 3
src = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
runs = []
for i, lib in enumerate(sys.argv[1:]):
    os.environ["FLGPU_LIB"] = os.path.join(ROOT, "fanlin-rs_amd", "libfanlin_gpu.so") if lib == "default" else os.path.abspath(lib)
    pkg = os.path.join(ROOT, "fanlin-rs_amd")
    spec = importlib.util.spec_from_file_location("fl_%d" % i, os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
    fl = importlib.util.module_from_spec(spec); sys.modules["fl_%d" % i] = fl; spec.loader.exec_module(fl); fl.load_library()
    st = fl.State(device=0, profile=True); st.__enter__()
    p = fl.make_params(300, 200, quality=75, front_end=fl.FE_JPEG)
    plan = fl.plan_output(p, W, H, C); stride = (int(plan.max_out_bytes) + 255) // 256 * 256
    dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
    run = st.prepared_batch([src.data_ptr() + k * H * W * C for k in range(n)], [(H, W, C)] * n, p, [dst.data_ptr() + k * stride for k in range(n)], [stride] * n)
    for _ in range(3): run(stream)
    runs.append((lib, st, run, dst))
torch.cuda.synchronize()
ref = None
for r in range(3):
    for lib, st, run, dst in runs:
        st.reset_stats()
        for _ in range(20): run(stream)
        torch.cuda.synchronize(); s = st.stats()
        same = "" if ref is None else (" streams identical to the first build" if torch.equal(dst, ref) else " STREAMS DIFFER")
        if ref is None: ref = dst.clone()
        print(f"{os.path.basename(lib):40s} frontend {s['frontend_ms'] / 20:.4f} ms{same}", flush=True)
