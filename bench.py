#!/usr/bin/env python3
"""bench.py -- throughput of the fanlin-rs image hot path on MI355X.

Workload (BASELINE.json configs[1], the metric's "1080p -> 300x200 resize+encode"): a batch of 1024 synthetic
1920x1080 RGB8 images, HBM-resident, each resized to w=300&h=200 with Lanczos3 (300x169), letterboxed onto the fill
colour (300x200 RGBA8) and encoded to a baseline JPEG stream on the device (quality 75, the reference's default,
src/query.rs:18) -- one call of flgpu_transform_batch_device per step.  After the timed region a sample of the results
is copied back and compared with the CPU oracle; a mismatch fails the run.  The resize-only figure (--frontend none)
and BASELINE config 2 (grayscale + blur) are measured next to it as extra keys.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Images are independent, so ranks never exchange pixels (weak
scaling: every rank owns its own 1024-image batch); the only collective is a one-off RCCL
broadcast of the read-only table blob from rank 0.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "fanlin-rs_amd")

SRC_W, SRC_H, SRC_C = 1920, 1080, 3
REQ_W, REQ_H = 300, 200
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def load_package():
    if "fanlin_rs_amd" in sys.modules:
        return sys.modules["fanlin_rs_amd"]
    spec = importlib.util.spec_from_file_location("fanlin_rs_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["fanlin_rs_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def shard_batches(global_images: int, world: int):
    """Weak-scaling shard map: rank r owns images [r*per, (r+1)*per) of the global batch."""
    per = global_images // world
    return [(r * per, (r + 1) * per) for r in range(world)]


KERNEL_DTYPE_FULL = ("u8 (exact, as f16 subnormals) x the reference's f32 weights as three f16 terms -> f32 sums (vertical, v_mfma_f32_16x16x32_f16); "
                     "24-bit fixed-point intermediate (2^-14 steps, three i8 planes) x 24-bit fixed-point weights (2^-24 steps, three i8 digits), "
                     "eight of the nine digit products -> exact i32 (horizontal, v_mfma_i32_16x16x64_i8; lowest plane x lowest digit, < 2^-18 of a pixel step, not computed): "
                     "no operand narrower than f32's 24 bits")
KERNEL_DTYPE_PACKED = ("u8 x f16-pair weights (22 bit) -> f32 acc (vertical, MFMA); i16 (1/64 steps) x 17-bit fixed weights -> i32 exact (horizontal, MFMA) "
                       "[switch mfma_arith=1: rounds 2-3's arithmetic, narrower than the reference's]")
KERNEL_DTYPE_STREAM = "f32 (one fused multiply-add per tap, vertical then horizontal)"


def kernel_dtype(stats, st=None) -> str:
    """The arithmetic the dominant kernel really computes in (not a precision claim: every byte is checked to lie within 1 LSB of
    the reference's f32 arithmetic, see verified_against)."""
    if stats.get("mfma_launches", 0) <= stats.get("wtile_launches", 0):   # (the resample ran on the f32 streaming kernel; a blur on the window-tile kernel also counts as a matrix-pipe launch)
        return KERNEL_DTYPE_STREAM
    return KERNEL_DTYPE_PACKED if (st is not None and st.debug_get("mfma_arith") == 1) else KERNEL_DTYPE_FULL


def cpu_baseline(n_images: int, workload: dict, max_threads: int = 16):
    """Times the CPU oracle (reference arithmetic, one image per thread, like one request per
    tokio worker) on a bounded sample of the same workload.  Reported, never a target."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import synth
    oracle = oracle_lib.load()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, max_threads))  # a 1-GPU box's CPU share is 16 cores, whatever the host exposes
    imgs = [synth.uniform(SRC_H, SRC_W, SRC_C, index=1000 + i) for i in range(min(n_images, 8))]

    def one(i):
        px = oracle.process_pixels(imgs[i % len(imgs)], REQ_W, REQ_H, blur_sigma=workload["blur_sigma"],
                                   grayscale=workload["grayscale"], crop=workload["crop"], arith=oracle_lib.ARITH_REF)
        return oracle.jpeg_encode(px, workload.get("quality", 75)) if workload.get("jpeg") else px

    one(0)  # page in
    t0 = time.perf_counter()
    one(0)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(one, range(n_images)))
    dt = time.perf_counter() - t0
    return {"value": n_images / dt, "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "host_cores_visible": avail, "single_thread_ms_per_image": t1 * 1e3,
            "sample": f"{n_images} synthetic 1920x1080 RGB8 images (uniform bytes), oracle/libfanlin_oracle.so "
                      f"(C restatement of image 0.25.6, reference arithmetic), resize + letterbox"
                      + (" + JPEG encode (q 75)" if workload.get("jpeg") else "") + f", one image per thread on {cores} threads"}


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def config0(fl, st, runs: int = 200):
    """BASELINE.json configs[0]: the reference's own demo request, images/lenna.jpg (512x512) -> w=300&h=200 (200x200 Lanczos3,
    letterboxed to 300x200, JPEG quality 75), once on the CPU path and once through the library, from the FILE bytes on
    (decode + resize + letterbox + encode), one request at a time, p50 over `runs`.  The reference's README quotes 18.06 ms
    p50 for this request INCLUDING its HTTP stack and file fetch on an i7-13700HX (/root/reference/README.md:111-114)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle = oracle_lib.load()
    path = os.path.join(ROOT, "tests", "golden", "lenna_reference.jpg")
    try:
        data = open(path, "rb").read()
    except OSError:
        return None

    def cpu_once():
        px = oracle.process_pixels(oracle.jpeg_decode(data), REQ_W, REQ_H, arith=oracle_lib.ARITH_REF)
        return oracle.jpeg_encode(px, 75)

    def p50(fn):
        fn()
        ts = []
        for _ in range(runs):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        a = np.sort(np.array(ts))
        return float(a[len(a) // 2]), float(a[min(len(a) - 1, int(len(a) * 0.99))])

    out = {"request": "tests/golden/lenna_reference.jpg (the reference's images/lenna.jpg, 512x512, 343,160 B) -> w=300&h=200, JPEG q 75, from the file bytes on",
           "runs": runs, "reference_readme_p50_ms": 18.06,
           "reference_readme_note": "whole HTTP path incl. local-file fetch, i7-13700HX, README.md:111-114 -- context only"}
    c50, c99 = p50(cpu_once)
    out["cpu_oracle"] = {"p50_ms": c50, "p99_ms": c99, "threads": 1, "kind": "port", "cpu_model": cpu_model(),
                         "path": "oracle: zune-jpeg-style decode + Lanczos3 (reference arithmetic) + letterbox + JPEG encode"}
    try:
        q = f"w={REQ_W}&h={REQ_H}"
        body = st.process_jpeg(data, q)[2]
        g50, g99 = p50(lambda: st.process_jpeg(data, q))
        out["gpu_single_request"] = {"p50_ms": g50, "p99_ms": g99, "stream_bytes": len(body),
                                     "path": "flgpu_process_jpeg: host Huffman decode on the caller's thread, IDCT + colour + resize + letterbox + JPEG encode on the device, one request in flight"}
    except Exception as e:
        out["gpu_single_request"] = {"skipped": repr(e)[:160]}
    return out


def measured_traffic(workload: str):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    bench.py cannot run under the profiler and time itself at once, so the committed summary of the last
    profiled run of this workload is reported, or null if there is none."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    return rec.get(workload, {}).get("hbm_bytes_per_launch")


def baseline_metric() -> str:
    """BASELINE.json's metric string, verbatim (`value` is its images/s half; the p50 half is in `latency`)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "images/sec whole-node (1080p\u2192300\u00d7200 resize+encode); p50 per-image ms"


def broadcast_icc_lut(fl, st, rank, dev, cdev):
    """the read-only CMYK -> sRGB device-link table (17^4 x 3 u16 = 501 KB) is baked once, on rank 0, and
    reaches the other GPUs by one broadcast (RCCL over xGMI with --backend nccl); every rank then converts the same
    CMYK pixels and the results are compared.  Untimed; any failure is reported in the JSON line, never raised."""
    import numpy as np
    import torch
    import torch.distributed as dist
    try:
        lut = torch.zeros(17 ** 4 * 3 * 2, dtype=torch.uint8, device=cdev)   # u16 table as bytes (every backend moves uint8)
        baked = 0
        if rank == 0:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import synth_icc
                st.set_cmyk_profile(synth_icc.cmyk_profile())          # needs liblcms2 on the host
                lut = torch.from_numpy(st.get_cmyk_clut().view(np.uint8).reshape(-1).copy()).to(cdev)
                baked = 1
            except Exception:
                baked = 0
        # from here on every rank issues the same collectives whatever happens locally (a rank that raised between
        # two collectives would leave the others waiting)
        flag = torch.tensor([baked], dtype=torch.int64, device=cdev)
        dist.broadcast(flag, src=0)
        if int(flag.item()) == 0:
            return {"ok": False, "reason": "rank 0 could not bake a table (liblcms2 missing?)"}
        dist.broadcast(lut, src=0)
        digest = torch.zeros(2, dtype=torch.int64, device=cdev)
        local_ok, n_px = 1, 1 << 16
        try:
            st.set_cmyk_clut(lut.cpu().numpy().view(np.uint16))
            px = np.random.default_rng(7).integers(0, 256, (n_px, 4), dtype=np.uint8)
            rgb = st.cmyk_to_rgb(px).astype(np.uint64)
            digest = torch.tensor([int(rgb.sum()), int((rgb * (np.arange(rgb.size, dtype=np.uint64).reshape(rgb.shape) % 251 + 1)).sum() % (1 << 62))],
                                  dtype=torch.int64, device=cdev)
        except Exception:
            local_ok = 0
        ref = digest.clone()
        dist.broadcast(ref, src=0)
        same = torch.tensor([int(local_ok and bool((ref == digest).all()))], dtype=torch.int64, device=cdev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        return {"ok": bool(same.item()), "bytes": int(lut.numel()), "pixels_checked_per_rank": n_px}
    except Exception as e:  # the bench line must survive
        return {"ok": False, "reason": repr(e)[:200]}


def verify_sample(fl, st, src, dst, out_stride, results, params, plan, fe_name, count=16):
    """Checks `count` images of the batch just processed against the CPU oracle (the checker here, never the thing
    measured).  Pixels: the same request is sent through flgpu_transform for the picked sources -- the kernel a request
    gets depends on its geometry only, so these are the pixels the timed batch produced -- and every byte must lie within
    1 LSB of the oracle's reference arithmetic (the north-star tolerance for resampling; the rate of off-by-one bytes is
    reported).  With the JPEG front end the stream the TIMED batch wrote must equal, byte for byte, the oracle encoder's
    stream of those device pixels; without a front end the timed batch's pixels must equal them.
    Returns (n_verified, error string or None, fraction of bytes off by one)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle = oracle_lib.load()
    n = src.shape[0]
    picks = sorted(set(int(i) for i in np.linspace(0, n - 1, count)))
    px_params = fl.make_params(REQ_W, REQ_H, crop=bool(params.crop), blur_sigma=float(params.blur_sigma), grayscale=bool(params.grayscale),
                               front_end=fl.FE_NONE)
    off, total = 0, 0
    for i in picks:
        img = src[i].cpu().numpy()
        kw = dict(blur_sigma=float(params.blur_sigma), grayscale=bool(params.grayscale), crop=bool(params.crop))
        want_ref = oracle.process_pixels(img, REQ_W, REQ_H, arith=oracle_lib.ARITH_REF, **kw)
        dev_px = st.process_pixels(img, px_params)
        if dev_px.shape != want_ref.shape:
            return 0, f"image {i}: device pixels have shape {dev_px.shape}, the oracle's {want_ref.shape}", 0.0
        d = np.abs(dev_px.astype(np.int16) - want_ref.astype(np.int16))
        if int(d.max()) > 1:
            return 0, f"pixels of image {i} differ from the oracle's reference arithmetic by {int(d.max())} LSB", 0.0
        off += int((d > 0).sum())
        total += d.size
        raw = dst[i].cpu().numpy()
        if fe_name == "jpeg":
            nb = results[i][1]
            if raw[:nb].tobytes() != oracle.jpeg_encode(dev_px, int(params.quality)):
                return 0, f"JPEG stream of image {i} ({nb} bytes) differs from the oracle encoder's stream of the same pixels", 0.0
        elif fe_name == "none":
            if not np.array_equal(raw[: plan.pixel_bytes].reshape(dev_px.shape), dev_px):
                return 0, f"pixels of image {i} in the timed batch differ from the same request sent alone", 0.0
        else:
            return 0, None, 0.0  # plane front ends are checked by the test-suite only
    return len(picks), None, off / max(total, 1)


def synthetic_jpeg_files(count=4):
    """Photo-like 1920x1080 baseline JPEGs (quality 85, 4:2:0 -- what a web origin typically serves) for the JPEG-source
    latency probe; written to a temporary directory."""
    import tempfile
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    d = tempfile.mkdtemp(prefix="flgpu_jpeg_")
    paths = []
    for i in range(count):
        p = os.path.join(d, f"src{i}.jpg")
        Image.fromarray(synth.photo(SRC_H, SRC_W, 3, index=3000 + i)).save(p, "JPEG", quality=85, subsampling=2)
        paths.append(p)
    return paths


def latency_probe_c(args, query: str, front_end: int, pinned: int = 0, jpeg_files=()):
    """The same probe from a plain C program (tools/latency/latency_probe.c, built by __graft_entry__.build()):
    pthread callers instead of Python threads, so the interpreter lock is not part of the measurement."""
    exe = os.path.join(ROOT, "tools", "latency", "latency_probe")
    if not os.path.exists(exe):
        return None
    import subprocess
    try:
        r = subprocess.run([exe, str(args.latency_threads), str(args.latency_requests), str(SRC_W), str(SRC_H), query, str(front_end),
                            str(args.queue_lanes), str(args.queue_max_batch), str(pinned)] + list(jpeg_files), capture_output=True, text=True, timeout=300)
        return json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else None
    except Exception:
        return None


def config4_one_gpu(fl, st, dev, stream, steps):
    import numpy as np
    import torch
    n = 1000
    kinds = [(2160, 3840)] * (n // 10) + [(1080, 1920)] * (6 * n // 10) + [(120, 160)] * (3 * n // 10)
    rng = np.random.default_rng(4)
    rng.shuffle(kinds)
    q = fl.Query.parse("w=300&h=200&webp=true&quality=85")
    params, _ = q.to_params(fl.Format.from_accept_header("image/webp"), input_is_jpeg=True)
    srcs = {k: torch.randint(0, 256, (sum(1 for x in kinds if x == k), k[0], k[1], 3), dtype=torch.uint8, device=dev) for k in set(kinds)}
    idx = {k: 0 for k in srcs}
    ptrs, shapes = [], []
    for k in kinds:
        ptrs.append(srcs[k].data_ptr() + idx[k] * k[0] * k[1] * 3)
        idx[k] += 1
        shapes.append((k[0], k[1], 3))
    plan = fl.plan_output(params, 1920, 1080, 3)
    stride = (int(plan.out_bytes) + 255) // 256 * 256
    dst = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    run = st.prepared_batch(ptrs, shapes, params, [dst.data_ptr() + i * stride for i in range(n)], [stride] * n)
    for _ in range(2):
        run(stream)
    torch.cuda.synchronize()
    st.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        run(stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    s2 = st.stats()
    alg = sum(h * w * 3 for h, w in kinds) + n * int(plan.pixel_bytes)
    k_ms = s2["resample_ms"] / steps
    out = {"images_per_s": n / dt, "ms_per_step": dt * 1e3, "images_per_step": n,
           "stage_ms_per_step": {"resample": k_ms, "frontend": s2["frontend_ms"] / steps},
           "resample_launches_per_step": s2["resample_launches"] / steps, "matrix_pipe_launches_per_step": s2["mfma_launches"] / steps,
           "roofline": {"bound": "hbm", "kernel": "all resample kernels of the step (matrix-pipe: 4K and 1080p; window-tile: thumbnails)", "kernel_ms": k_ms,
                        "achieved": alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else 0.0, "algorithmic_bytes_per_step": alg}}
    del run, dst, srcs
    torch.cuda.empty_cache()
    return out


def config3_share(fl, dev, stream, n):
    """8,192 x 1080p = 51 GB of sources resident in HBM, one flgpu_transform_batch_device call with an `rgb=` fill; pictures from all
    over the batch are compared with the same request sent alone."""
    import numpy as np
    import torch
    free, _ = torch.cuda.mem_get_info()
    need = n * SRC_W * SRC_H * SRC_C + n * 240128 + (4 << 30)
    if free < need:
        return {"skipped": f"{free / 2**30:.0f} GiB free, {need / 2**30:.0f} GiB needed"}
    p = fl.make_params(REQ_W, REQ_H, fill=(200, 16, 99), front_end=fl.FE_NONE)
    plan = fl.plan_output(p, SRC_W, SRC_H, SRC_C)
    stride = (int(plan.out_bytes) + 255) // 256 * 256
    src = torch.empty((n, SRC_H, SRC_W, SRC_C), dtype=torch.uint8, device=dev)
    for k in range(0, n, 512):   # (filled in pieces: randint's temporaries are int64)
        src[k:k + 512].random_(0, 256)
    dst = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    img = SRC_W * SRC_H * SRC_C
    out = {}
    for label, kw in (("one device context", dict(device=0)), ("two shards on this device (devices = [0, 0])", dict(devices=[0, 0]))):
        with fl.State(profile=True, **kw) as st:
            run = st.prepared_batch([src.data_ptr() + k * img for k in range(n)], [(SRC_H, SRC_W, SRC_C)] * n, p, [dst.data_ptr() + k * stride for k in range(n)], [stride] * n)
            run(stream)
            torch.cuda.synchronize()
            st.reset_stats()
            steps = 3
            t0 = time.perf_counter()
            for _ in range(steps):
                run(stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            s2 = st.stats()
            st.batch_results()
            k_ms = s2["resample_ms"] / steps
            alg = n * (img + int(plan.pixel_bytes))
            bad = 0
            for k in (0, 1, n // 3, n // 2 + 7, n - 2, n - 1):
                alone = st.process_pixels(src[k].cpu().numpy(), p)
                bad += int(not np.array_equal(dst[k, :alone.size].cpu().numpy().reshape(alone.shape), alone))
            out[label] = {"images_per_s": n / dt, "ms_per_step": dt * 1e3, "kernel_ms_per_step": k_ms, "pictures_differing_from_the_request_sent_alone": bad,
                          "source_bytes": n * img,
                          "roofline": {"bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else 0.0}}
            del run
    del src, dst
    torch.cuda.empty_cache()
    return out


def latency_probe(fl, st, params, n_requests: int, n_threads: int):
    """Per-image latency of the drop-in entry point: concurrent callers of flgpu_transform with HOST buffers
    (PCIe in both directions included), packed into shared launches by the library's request queue."""
    import threading
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    imgs = [synth.uniform(SRC_H, SRC_W, SRC_C, index=2000 + i) for i in range(8)]
    lat = [0.0] * n_requests
    nxt = [0]
    lock = threading.Lock()
    st.process_pixels(imgs[0], params)  # warm up (tables, staging buffers)
    before = st.stats()

    def worker():
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= n_requests:
                return
            t0 = time.perf_counter()
            st.process_pixels(imgs[i % len(imgs)], params)
            lat[i] = (time.perf_counter() - t0) * 1e3

    t0 = time.perf_counter()
    ts = [threading.Thread(target=worker) for _ in range(n_threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    wall = time.perf_counter() - t0
    after = st.stats()
    a = np.sort(np.array(lat))
    return {"p50_ms": float(a[len(a) // 2]), "p99_ms": float(a[min(len(a) - 1, int(len(a) * 0.99))]),
            "requests": n_requests, "caller_threads": n_threads, "images_per_s": n_requests / wall,
            "queue_flushes": int(after["queue_flushes"] - before["queue_flushes"]),
            "path": "flgpu_transform, host buffers (H2D + kernels + D2H), request-batching queue"}


def timed_loop(run, stream, steps, warmup, st, world, dist, cdev):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; max over ranks."""
    import torch
    for _ in range(warmup):
        run(stream)
    torch.cuda.synchronize()
    st.reset_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, st.stats()


def main_one_context(args):
    """--one-context: ONE process and ONE flgpu context over N GPUs (flgpu_config.devices[], the drop-in's mode: one Arc<State>
    for all workers, src/main.rs:108-112).  The batch of N x --batch images is cut by flgpu_plan_shards (contiguous shards balanced
    by algorithmic bytes), shard k is resident on device k, and one flgpu_transform_batch_device call per step runs all shards
    concurrently and returns when every one is done.  No torch.distributed, no collective (the library hands the CMYK table
    round with one ncclBroadcast of its own when a profile is set).  Same JSON line as the per-rank mode."""
    import numpy as np
    import torch
    fl = load_package()
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    N = args.gpus
    devices = [k % ndev for k in range(N)]  # fewer GPUs than asked for: shards share them (rehearsal on a 1-GPU box)
    FE = {"none": fl.FE_NONE, "jfif444": fl.FE_JFIF444, "webp420": fl.FE_WEBP420, "jpeg": fl.FE_JPEG}
    fe = FE[args.frontend]
    params = fl.make_params(REQ_W, REQ_H, crop=args.crop, blur_sigma=args.blur, grayscale=args.grayscale, front_end=fe, quality=args.quality)
    plan = fl.plan_output(params, SRC_W, SRC_H, SRC_C)
    n = args.batch * N
    src_bytes = SRC_W * SRC_H * SRC_C
    out_stride = (int(plan.max_out_bytes) + 255) // 256 * 256
    shapes = [(SRC_H, SRC_W, SRC_C)] * n
    shard_of, shard_bytes = fl.plan_shards(N, shapes, params)
    srcp, dstp, keep = [0] * n, [0] * n, []
    for k in range(N):
        idx = [i for i in range(n) if int(shard_of[i]) == k]
        dev = torch.device("cuda", devices[k])
        gen = torch.Generator(device=dev)
        gen.manual_seed(0xFA171200 + k)
        s_k = torch.empty((len(idx), SRC_H, SRC_W, SRC_C), dtype=torch.uint8, device=dev)
        for i in range(0, len(idx), 64):
            s_k[i:i + 64] = torch.randint(0, 256, (min(64, len(idx) - i), SRC_H, SRC_W, SRC_C), dtype=torch.uint8, device=dev, generator=gen)
        d_k = torch.zeros((len(idx), out_stride), dtype=torch.uint8, device=dev)
        keep.append((s_k, d_k))
        for j, i in enumerate(idx):
            srcp[i] = s_k.data_ptr() + j * src_bytes
            dstp[i] = d_k.data_ptr() + j * out_stride
    st = fl.State(devices=devices, profile=True) if N > 1 else fl.State(device=devices[0], profile=True)
    run = st.prepared_batch(srcp, shapes, params, dstp, [out_stride] * n)

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    for _ in range(max(args.warmup, 1)):
        run(0)
    sync_all()
    st.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run(0)
    sync_all()
    elapsed = time.perf_counter() - t0
    stats = st.stats()
    ok, err = 1, None
    srcs_c, dsts_c, _ = run._keep   # (collected whatever the front end: a device-side failure of the batch comes back here)
    fl._check(st._lib.flgpu_batch_results(st._ctx, n, dsts_c), st._ctx)
    if fe == fl.FE_JPEG and any(d.bytes == 0 for d in dsts_c):
        ok, err = 0, "an encoded stream did not fit its destination"
    launches = max(int(stats["resample_launches"]), 1)
    k_ms = stats["resample_ms"] / launches
    alg_bytes = (stats["resample_src_bytes"] + stats["resample_dst_bytes"]) / launches
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    line = {"metric": baseline_metric(), "value": n * args.steps / elapsed, "unit": "images/s", "n_gpus": N, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": kernel_dtype(stats, st), "data": "synthetic",
            "config": {"workload": f"{n} x {SRC_W}x{SRC_H} RGB8 (uniform bytes, HBM-resident, shard k on device k) -> w={REQ_W}&h={REQ_H} Lanczos3 + letterbox RGBA8"
                                   + (f" + baseline JPEG encode (q {args.quality}) on the device" if fe == fl.FE_JPEG else ""),
                       "mode": "one process, one context over the node's GPUs (flgpu_config.devices[]), shards by flgpu_plan_shards",
                       "devices": devices, "shard_images": [int((shard_of == k).sum()) for k in range(N)], "images_per_gpu_per_step": args.batch},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "resample_mfma_kernel" if stats.get("mfma_launches", 0) > stats.get("wtile_launches", 0) else "resample_stream_kernel", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": alg_bytes, "note": "per launch and device: average over the shards' launches"},
            "stage_ms_per_step": {"resample": stats["resample_ms"] / args.steps, "blur": stats["blur_ms"] / args.steps, "frontend": stats["frontend_ms"] / args.steps},
            "cpu_baseline": None}
    # the same two objects as the per-rank mode: committed PMC traffic of the per-GPU workload, and the CPU port on a bounded sample
    line["roofline"]["traffic"] = measured_traffic(f"{args.batch} x {SRC_W}x{SRC_H} RGB8 (uniform bytes, HBM-resident) -> w={REQ_W}&h={REQ_H} Lanczos3"
                                                   + (" crop" if args.crop else " + letterbox RGBA8")
                                                   + (f" + baseline JPEG encode (q {args.quality}) on the device" if fe == fl.FE_JPEG else ""))
    if args.cpu_images > 0 and ok:
        line["cpu_baseline"] = cpu_baseline(args.cpu_images, {"blur_sigma": args.blur, "grayscale": args.grayscale, "crop": args.crop,
                                                              "jpeg": fe == fl.FE_JPEG, "quality": args.quality}, args.cpu_threads)
    if err:
        line["error"] = err
    print(json.dumps(line), flush=True)
    st.close()
    if not ok:
        raise SystemExit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--one-context", action="store_true",
                    help="one process, one flgpu context over --gpus devices (the drop-in's mode) instead of one process per GPU; do not launch under torch.distributed.run")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="timed steps (the default keeps the timed region above one second)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU per step")
    ap.add_argument("--blur", type=float, default=0.0, help="blur sigma (config 2 uses 10)")
    ap.add_argument("--grayscale", action="store_true")
    ap.add_argument("--crop", action="store_true")
    ap.add_argument("--quality", type=int, default=75, help="JPEG quality (reference default, src/query.rs:18)")
    ap.add_argument("--frontend", choices=["none", "jfif444", "webp420", "jpeg"], default="jpeg",
                    help="what follows the pixel pipeline; jpeg = the complete baseline JPEG encode on the device (the metric's resize+encode)")
    ap.add_argument("--extra-steps", type=int, default=40, help="steps of the secondary measurements (resize only, config 2); 0 = skip")
    ap.add_argument("--verify-images", type=int, default=16, help="images compared with the CPU oracle after the timed region (0 = skip)")
    ap.add_argument("--cpu-images", type=int, default=1024, help="CPU-baseline sample size (0 = skip); ~16 ms of CPU work per image")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the CPU baseline (one image per thread)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (= RCCL, the real multi-GPU path) or gloo (rehearsal: several ranks may share one GPU)")
    ap.add_argument("--latency-requests", type=int, default=4096,
                    help="requests of the per-image latency probe through flgpu_transform (0 = skip)")
    ap.add_argument("--latency-threads", type=int, default=64, help="concurrent caller threads of the latency probe")
    ap.add_argument("--config0-runs", type=int, default=200, help="runs of BASELINE config 0 (lenna.jpg, CPU oracle and one GPU request at a time); 0 = skip")
    ap.add_argument("--config4", type=int, default=1, help="1 = BASELINE config 4's mixed batch on this GPU as an `extra` entry")
    ap.add_argument("--config3-share", type=int, default=8192, help="pictures of BASELINE config 3's per-GPU share run as one device batch (an `extra` entry; 0 = skip)")
    ap.add_argument("--queue-lanes", type=int, default=0, help="batches the request queue keeps in flight (0 = library default)")
    ap.add_argument("--queue-max-batch", type=int, default=0, help="largest batch the request queue forms (0 = library default)")
    args = ap.parse_args()
    if args.one_context:
        return main_one_context(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    ndev = torch.cuda.device_count()
    local_dev = local_rank % max(ndev, 1)  # gloo rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where collectives run

    fl = load_package()
    FE = {"none": fl.FE_NONE, "jfif444": fl.FE_JFIF444, "webp420": fl.FE_WEBP420, "jpeg": fl.FE_JPEG}
    fe = FE[args.frontend]
    params = fl.make_params(REQ_W, REQ_H, crop=args.crop, blur_sigma=args.blur, grayscale=args.grayscale, front_end=fe, quality=args.quality)
    plan = fl.plan_output(params, SRC_W, SRC_H, SRC_C)
    n = args.batch
    src_bytes = SRC_W * SRC_H * SRC_C

    # synthetic, HBM-resident input: uniform random bytes, seeded per rank (rank r = images [r*n, (r+1)*n))
    gen = torch.Generator(device=dev)
    gen.manual_seed(0xFA171200 + rank)
    src = torch.empty((n, SRC_H, SRC_W, SRC_C), dtype=torch.uint8, device=dev)
    for i in range(0, n, 64):
        src[i:i + 64] = torch.randint(0, 256, (min(64, n - i), SRC_H, SRC_W, SRC_C), dtype=torch.uint8, device=dev, generator=gen)
    # uniform noise is the least compressible input there is: give every stream the format's worst case so none can overflow
    out_stride = (int(plan.max_out_bytes) + 255) // 256 * 256
    dst = torch.zeros((n, out_stride), dtype=torch.uint8, device=dev)

    st = fl.State(device=local_dev, profile=True, queue_lanes=args.queue_lanes, max_batch=args.queue_max_batch)
    srcp = [src.data_ptr() + i * src_bytes for i in range(n)]
    dstp = [dst.data_ptr() + i * out_stride for i in range(n)]
    shapes = [(SRC_H, SRC_W, SRC_C)] * n
    run = st.prepared_batch(srcp, shapes, params, dstp, [out_stride] * n)
    stream = torch.cuda.current_stream().cuda_stream

    # plan + build tables (untimed), then make every rank use rank 0's table blob (RCCL broadcast over xGMI)
    run(stream)
    torch.cuda.synchronize()
    if world > 1:
        blob = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
        nbytes = st.copy_tables(blob.data_ptr(), blob.numel())
        size_t = torch.tensor([nbytes], dtype=torch.int64, device=cdev)
        dist.broadcast(size_t, src=0)
        # every rank learns whether ALL ranks planned the same tables before anybody moves on: a rank that raised between
        # two collectives would leave the others waiting for the RCCL timeout
        same = torch.tensor([int(int(size_t.item()) == nbytes)], dtype=torch.int64, device=cdev)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        if int(same.item()) == 0:
            if rank == 0:
                print(json.dumps({"error": "ranks planned different tables"}), flush=True)
            dist.destroy_process_group()
            raise SystemExit(1)
        payload = blob[:nbytes].to(cdev)           # nccl: stays on the device (xGMI); gloo rehearsal: staged through the host
        dist.broadcast(payload, src=0)
        blob[:nbytes] = payload.to(dev)
        torch.cuda.synchronize()
        st.import_tables(blob.data_ptr(), nbytes)
        del blob, payload
        icc_note = broadcast_icc_lut(fl, st, rank, dev, cdev)
    else:
        icc_note = None

    elapsed, stats = timed_loop(run, stream, args.steps, args.warmup, st, world, dist, cdev)

    # ---- check what the timed region produced ----
    verified, verr, off_by_one = 0, None, 0.0
    results = None
    # (collected whatever the front end: flgpu_batch_results is also where a device-side failure of the batch -- the matrix-pipe
    # kernel's bounded LDS waits -- comes back as FLGPU_ERR_DEVICE)
    srcs_c, dsts_c, _ = run._keep
    fl._check(st._lib.flgpu_batch_results(st._ctx, n, dsts_c), st._ctx)
    if fe == fl.FE_JPEG:
        results = [(d.flags, d.bytes) for d in dsts_c]
        if any(b == 0 for _, b in results):
            verr = "an encoded stream did not fit its destination"
    if args.verify_images > 0 and verr is None:
        verified, verr, off_by_one = verify_sample(fl, st, src, dst, out_stride, results, params, plan, args.frontend, args.verify_images)
    ok_flag = 0 if verr else 1
    if world > 1:
        okt = torch.tensor([ok_flag], dtype=torch.int64, device=cdev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok_flag = int(okt.item())

    # ---- secondary measurements on the same batch (single-GPU runs only; untimed for `value`) ----
    extra = {}
    if world == 1 and args.extra_steps > 0 and ok_flag:
        def measure(p):
            r = st.prepared_batch(srcp, shapes, p, dstp, [out_stride] * n)
            r(stream)
            torch.cuda.synchronize()
            el, s2 = timed_loop(r, stream, args.extra_steps, 2, st, 1, dist, cdev)
            return {"images_per_s": n * args.extra_steps / el, "ms_per_step": el / args.extra_steps * 1e3,
                    "stage_ms_per_step": {"resample": s2["resample_ms"] / args.extra_steps, "blur": s2["blur_ms"] / args.extra_steps,
                                          "frontend": s2["frontend_ms"] / args.extra_steps}}
        if fe != fl.FE_NONE:
            extra["resize_only (config 1 without the encode)"] = measure(
                fl.make_params(REQ_W, REQ_H, crop=args.crop, blur_sigma=args.blur, grayscale=args.grayscale, front_end=fl.FE_NONE))
        # the same resize with the reference's own arithmetic width: the streaming kernel (f32 accumulation, bit-exact against the
        # oracle's fused-order mode) serves the request when the matrix-pipe kernel is switched off
        with st.switches(no_mfma=1):
            f32m = measure(fl.make_params(REQ_W, REQ_H, crop=args.crop, blur_sigma=args.blur, grayscale=args.grayscale, front_end=fl.FE_NONE))
        alg = SRC_W * SRC_H * SRC_C * n + int(fl.plan_output(fl.make_params(REQ_W, REQ_H, crop=args.crop, blur_sigma=args.blur, grayscale=args.grayscale), SRC_W, SRC_H, SRC_C).pixel_bytes) * n
        f32m["roofline"] = {"bound": "hbm", "kernel": "resample_stream_kernel", "kernel_ms": f32m["stage_ms_per_step"]["resample"],
                            "achieved": alg / (f32m["stage_ms_per_step"]["resample"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": alg / (f32m["stage_ms_per_step"]["resample"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        extra["f32 streaming kernel (switch no_mfma), resize only"] = f32m
        # rounds 2-3's packed arithmetic (22-bit vertical weights, 1/64-step intermediate, 17-bit horizontal weights): narrower than
        # the reference's f32, kept selectable; the whole metric (resize + letterbox + encode) and the resize alone
        with st.switches(mfma_arith=1):
            pk = measure(params)
            pk_r = measure(fl.make_params(REQ_W, REQ_H, crop=args.crop, blur_sigma=args.blur, grayscale=args.grayscale, front_end=fl.FE_NONE))
        pk["dtype"] = KERNEL_DTYPE_PACKED
        pk["resize_only"] = pk_r
        pk["roofline"] = {"bound": "hbm", "kernel": "resample_mfma_kernel<packed>", "kernel_ms": pk["stage_ms_per_step"]["resample"],
                          "achieved": alg / (pk["stage_ms_per_step"]["resample"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": alg / (pk["stage_ms_per_step"]["resample"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        extra["packed arithmetic (switch mfma_arith=1), same workload as `value`"] = pk
        # round 4: the window-tile matrix-pipe kernel (csrc/fl_wtile.h) -- a sigma-20 blur behind the flagship resample, and a mild
        # down-scale (ratio 1.92) of the first 256 pictures of the same batch
        if not args.blur:
            extra["config1 + blur sigma 20 (blur on the window-tile matrix-pipe kernel), pixels out"] = measure(
                fl.make_params(REQ_W, REQ_H, crop=args.crop, blur_sigma=20.0, front_end=fl.FE_NONE))
        if SRC_W == 1920 and SRC_H == 1080 and n >= 256:
            pm = fl.make_params(1000, 562, front_end=fl.FE_NONE)
            plm = fl.plan_output(pm, SRC_W, SRC_H, SRC_C)
            strm = (int(plm.out_bytes) + 255) // 256 * 256
            dstm = torch.zeros((256, strm), dtype=torch.uint8, device=dev)
            rm = st.prepared_batch(srcp[:256], shapes[:256], pm, [dstm.data_ptr() + k * strm for k in range(256)], [strm] * 256)
            rm(stream)
            torch.cuda.synchronize()
            steps_m = max(4, args.extra_steps // 4)
            el, s2 = timed_loop(rm, stream, steps_m, 2, st, 1, dist, cdev)
            algm = 256 * (SRC_W * SRC_H * SRC_C + int(plm.pixel_bytes))
            k_ms = s2["resample_ms"] / steps_m
            extra["256 x 1080p -> w=1000&h=562 (ratio 1.92, window-tile matrix-pipe kernel), pixels out"] = {
                "images_per_s": 256 * steps_m / el, "ms_per_step": el / steps_m * 1e3, "stage_ms_per_step": {"resample": k_ms},
                "wtile_launches": int(s2.get("wtile_launches", 0)),
                "roofline": {"bound": "hbm", "kernel": "resample_wtile_kernel", "kernel_ms": k_ms, "achieved": algm / (k_ms * 1e-3) / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algm / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
            del rm, dstm
        if not (args.grayscale and args.blur):
            extra["config2 (grayscale + blur sigma 10, pixels out)"] = measure(
                fl.make_params(REQ_W, REQ_H, blur_sigma=10.0, grayscale=True, front_end=fl.FE_NONE))
            extra["config2 + JPEG encode"] = measure(
                fl.make_params(REQ_W, REQ_H, blur_sigma=10.0, grayscale=True, front_end=fl.FE_JPEG, quality=args.quality))
        # BASELINE config 4 on ONE GPU: a mixed-size batch (3840x2160 : 1920x1080 : 160x120 = 1 : 6 : 3, seed-shuffled),
        # `w=300&h=200&webp=true&quality=85`: resize + letterbox + libwebp-style YUV420 front end, inputs resident in HBM
        if args.config4:
            try:
                extra["config4 on one GPU (4K : 1080p : thumbnail = 1 : 6 : 3, WebP 4:2:0 front end)"] = config4_one_gpu(fl, st, dev, stream, max(4, args.extra_steps // 4))
            except Exception as e:
                extra["config4 on one GPU (4K : 1080p : thumbnail = 1 : 6 : 3, WebP 4:2:0 front end)"] = {"skipped": repr(e)[:200]}
        # BASELINE config 3's per-GPU share: 65,536 pictures over 8 GPUs = 8,192 x 1080p (51 GB) resident on THIS GPU, one
        # flgpu_transform_batch_device call, `rgb=` fill -- and the same over a two-shard context on this one device
        if args.config3_share > 0:
            del run
            try:
                extra[f"config3 per-GPU share ({args.config3_share} x 1080p resident, rgb= fill, one device batch)"] = config3_share(fl, dev, stream, args.config3_share)
            except Exception as e:
                extra[f"config3 per-GPU share ({args.config3_share} x 1080p resident, rgb= fill, one device batch)"] = {"skipped": repr(e)[:200]}

    if rank == 0:
        total_images = n * world * args.steps
        value = total_images / elapsed
        # roofline of the dominant kernel (fused streaming resample), from HIP events recorded by the
        # library on the launching stream around every launch of the timed region
        launches = max(int(stats["resample_launches"]), 1)
        k_ms = stats["resample_ms"] / launches
        alg_bytes = (stats["resample_src_bytes"] + stats["resample_dst_bytes"]) / launches
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        workload = {"blur_sigma": args.blur, "grayscale": args.grayscale, "crop": args.crop, "jpeg": fe == fl.FE_JPEG, "quality": args.quality}
        line = {
            "metric": baseline_metric(),
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": kernel_dtype(stats, st), "data": "synthetic",
            "config": {"workload": f"{n} x {SRC_W}x{SRC_H} RGB8 (uniform bytes, HBM-resident) -> w={REQ_W}&h={REQ_H} Lanczos3"
                                   + (" crop" if args.crop else " + letterbox RGBA8")
                                   + (" + grayscale" if args.grayscale else "") + (f" + blur sigma {args.blur:g}" if args.blur else "")
                                   + (f" + baseline JPEG encode (q {args.quality}) on the device" if fe == fl.FE_JPEG else (f" + {args.frontend} front end" if fe else "")),
                       "stages_in_value": ["resize", "letterbox"] + (["blur"] if args.blur else []) + ([args.frontend] if fe else []),
                       "images_per_gpu_per_step": n, "sharding": "one independent batch per rank, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "resample_mfma_kernel" if stats.get("mfma_launches", 0) > stats.get("wtile_launches", 0) else "resample_stream_kernel",
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes},
            "per_image_us_kernel": k_ms * 1e3 / n if n else None,
            "stage_ms_per_step": {"resample": stats["resample_ms"] / args.steps, "blur": stats["blur_ms"] / args.steps,
                                  "frontend": stats["frontend_ms"] / args.steps},
            "verified_images": verified if ok_flag else 0,
            "verified_against": "oracle: every pixel byte within 1 LSB of the reference arithmetic (ARITH_REF)"
                                + ("; JPEG streams byte-identical to the oracle encoder's stream of the device pixels" if fe == fl.FE_JPEG else ""),
            "off_by_one_fraction": off_by_one,
        }
        if verr:
            line["error"] = verr
        if results is not None:
            line["mean_stream_bytes"] = sum(b for _, b in results) / max(len(results), 1)
        line["roofline"]["traffic"] = measured_traffic(line["config"]["workload"])
        if extra:
            line["extra"] = extra
        if icc_note is not None:
            line["icc_lut_broadcast"] = icc_note
        if args.latency_requests > 0 and world == 1 and ok_flag:
            query = f"w={REQ_W}&h={REQ_H}" + ("&crop=true" if args.crop else "") + (f"&blur={int(args.blur)}" if args.blur else "") \
                + ("&grayscale=true" if args.grayscale else "") + f"&quality={args.quality}"
            line["latency"] = latency_probe_c(args, query, fe, 0) or latency_probe(fl, st, params, args.latency_requests, args.latency_threads)
            pinned = latency_probe_c(args, query, fe, 1)
            if pinned:
                line["latency_pinned"] = pinned
            # the same requests with JPEG FILES as sources: Huffman decoding on the caller threads, IDCT + colour on the device
            try:
                jp = latency_probe_c(args, query, fe, 0, synthetic_jpeg_files())
                if jp:
                    jp["path"] = ("flgpu_transform with FLGPU_IMG_JPEG_SOURCE (1920x1080 q85 4:2:0 files): entropy decoding on the caller's thread while a CPU is "
                                  "idle, on the device (fl_jpeghuff_dev.hip) beyond that; IDCT + colour + pipeline + encode on the device")
                    # the same probe with every file Huffman-decoded on the host (round 3's path), for comparison
                    os.environ["FLGPU_HOST_HUFFMAN"] = "1"
                    try:
                        jh = latency_probe_c(args, query, fe, 0, synthetic_jpeg_files())
                    finally:
                        del os.environ["FLGPU_HOST_HUFFMAN"]
                    if jh:
                        jh["path"] = "the same with FLGPU_HOST_HUFFMAN=1: every file entropy-decoded on its caller's thread"
                        line["latency_jpeg_sources_host_huffman"] = jh
                    line["latency_jpeg_sources"] = jp
                    # north_star's own end-to-end wording -- "resize+blur on 1080p -> 300x200", >= 50 k images/s on 8 GPUs: the same
                    # 64 callers with `blur=10` in the query, JPEG files in, JPEG streams out
                    if not args.blur:
                        jb = latency_probe_c(args, query + "&blur=10", fe, 0, synthetic_jpeg_files())
                        if jb:
                            jb["path"] = "the same files, query w=300&h=200&blur=10: decode + resize + letterbox + blur sigma 10 + JPEG encode"
                            line["latency_jpeg_sources_blur10"] = jb
                    # what a node of 8 such GPUs needs for north_star's 50 k images/s: 6.25 k per GPU, and host CPU for the callers
                    best = max((x.get("images_per_s", 0.0) for x in (jp, line.get("latency_jpeg_sources_blur10") or {}) if x), default=0.0)
                    cpu_ms = jp.get("host_cpu_ms_per_request")
                    line["node_budget"] = {
                        "north_star_images_per_s_at_8_gpus": 50000, "needed_per_gpu": 6250,
                        "measured_per_gpu_jpeg_files_in_jpeg_out": jp.get("images_per_s"),
                        "measured_per_gpu_with_blur10": (line.get("latency_jpeg_sources_blur10") or {}).get("images_per_s"),
                        "host_cpu_ms_per_request": cpu_ms,
                        "host_cores_busy_at_50k_per_s": (cpu_ms * 50.0) if cpu_ms else None,   # ms per request x 50,000 requests / 1000 ms
                        "host_cores_visible": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
                        "note": "one GPU measured; the 8-GPU figure is 8 x the per-GPU rate only if every GPU keeps this CPU share",
                    }
                    _ = best
            except Exception as e:  # Pillow missing: the probe is optional
                line["latency_jpeg_sources"] = {"skipped": repr(e)[:120]}
        if args.config0_runs > 0 and args.cpu_images > 0 and world == 1 and ok_flag:  # (--cpu-images 0 switches every CPU leg off)
            line["config0"] = config0(fl, st, args.config0_runs)
        if args.cpu_images > 0 and world == 1 and ok_flag:
            line["cpu_baseline"] = cpu_baseline(args.cpu_images, workload, args.cpu_threads)
            # "all host cores of the GPU box (core count stated)": the same sample once more on every core of the
            # affinity mask (on a 1-GPU share that is the same 16; on a whole host, all of them)
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            if avail > line["cpu_baseline"]["cores"]:
                line["cpu_baseline_all_cores"] = cpu_baseline(max(args.cpu_images, 8 * avail), workload, avail)
            else:
                line["cpu_baseline_all_cores"] = dict(line["cpu_baseline"], note="the affinity mask has no more cores than the 16-thread figure used")
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    st.close()
    if world > 1:
        dist.destroy_process_group()
    if not ok_flag:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
