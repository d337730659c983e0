"""One-off: random query strings through flgpu_process_image (the one-call entry) against the oracle driven by the
Query accessors -- checks the planning glue (as_is, size gate, negotiation, front-end choice), not the kernels again.
    python tests/tools/fuzz_process_image.py <cases> <seed>"""
import sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, parity, synth
from bench import load_package
fl = load_package(); oracle = oracle_lib.load()
n, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
st = fl.State(device=0)
bad = 0
for i in range(n):
    parts = []
    if rng.integers(0, 4): parts.append(f"w={int(rng.choice([10, 20, 64, 300, 555, 2000, 2001]))}")
    if rng.integers(0, 4): parts.append(f"h={int(rng.choice([10, 20, 48, 200, 333, 1000, 1001]))}")
    if rng.integers(0, 3) == 0: parts.append("rgb=" + ",".join(str(int(x)) for x in rng.integers(0, 256, int(rng.integers(1, 5)))))
    if rng.integers(0, 3) == 0: parts.append(f"quality={int(rng.integers(0, 120))}")
    for k in ("crop", "grayscale", "inverse", "avif", "webp"):
        if rng.integers(0, 4) == 0: parts.append(f"{k}={'true' if rng.integers(0, 2) else 'false'}")
    if rng.integers(0, 5) == 0: parts.append(f"blur={int(rng.integers(0, 40))}")
    rng.shuffle(parts)
    query = "&".join(parts)
    accept = fl.Format()
    if rng.integers(0, 2): accept.accept_webp()
    if rng.integers(0, 2): accept.accept_avif()
    fmt = int(rng.choice([fl.IN_JPEG, fl.IN_PNG, fl.IN_WEBP, fl.IN_GIF_FRAME]))
    c = 4 if fmt == fl.IN_GIF_FRAME else int(rng.choice([1, 3, 4]))
    img = synth.photo(int(rng.integers(20, 300)), int(rng.integers(20, 400)), c, index=i)
    orient = int(rng.choice([1, 1, 3, 6]))
    q = fl.Query.parse(query)
    try:
        mime, kind, payload = st.process_image(img, query, accept, fmt, orientation=orient)
        err = None
    except fl.FanlinError as e:
        err = e.status
    ok = True
    if q.unsupported_scale_size():
        ok = err == fl.ERR_PARSE
    elif q.as_is():
        ok = err is None and kind == fl.RESULT_AS_IS
    elif err is not None:
        ok = False
    else:
        gif = fmt == fl.IN_GIF_FRAME
        dims = q.dimensions()
        kw = dict(w=dims[0] if dims else None, h=dims[1] if dims else None, fill=q.fill_color(), crop=q.cropping(), grayscale=q.grayscale(),
                  inverse=q.inverse(), blur_sigma=0.0 if gif else q.blur(), orientation=0 if gif else orient,
                  filter=oracle_lib.FILTER_NEAREST if gif else oracle_lib.FILTER_LANCZOS3)
        if gif:
            px = oracle.process_pixels(img, arith=oracle_lib.ARITH_FMA, **kw)
        else:
            # the pixels this request gives when sent alone, after they have cleared the bar of the kernel that served them
            # (tests/parity.py: bit-exact for the streaming / tiled kernels, <= 1 LSB for the matrix-pipe kernel)
            try:
                px = parity.expected_pixels(fl, st, oracle, img, **{k: v for k, v in kw.items() if k != "filter"})
            except AssertionError as e:
                print("BAR", str(e)[:200]); px = None
        webp = (not gif) and q.use_webp() and accept.webp_accepted()
        avif = (not gif) and (not webp) and q.use_avif() and accept.avif_accepted()
        qual = min(max(q.quality(), 1), 100)
        want_mime = "image/webp" if webp else "image/avif" if avif else fl.MIME[fmt]
        if px is None or mime != want_mime: ok = False
        elif (webp or (fmt == fl.IN_WEBP and not avif)) and qual < 100:
            rgba = px if px.shape[2] == 4 else (np.concatenate([px[:, :, :1]] * 3 + [np.full(px.shape[:2] + (1,), 255, np.uint8)], 2) if px.shape[2] == 1 else
                                                np.concatenate([px[:, :, :1]] * 3 + [px[:, :, 1:]], 2) if px.shape[2] == 2 else
                                                np.concatenate([px, np.full(px.shape[:2] + (1,), 255, np.uint8)], 2))
            y, u, v, _ = oracle.webp_yuv420(np.ascontiguousarray(rgba))
            ok = kind == fl.RESULT_WEBP_PLANES and np.array_equal(payload.y, y) and np.array_equal(payload.u, u) and np.array_equal(payload.v, v)
        elif fmt == fl.IN_JPEG and not webp and not avif:
            ok = kind == fl.RESULT_JPEG_STREAM and payload == oracle.jpeg_encode(px, qual)
        else:
            ok = kind == fl.RESULT_PIXELS and np.array_equal(payload, px)
    if not ok:
        bad += 1
        print("MISMATCH", i, repr(query), "fmt", fmt, "accept", accept.flags, "orient", orient, "err", err, flush=True)
print("cases", n, "bad", bad)
