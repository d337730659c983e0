"""One-off: the extremes of the request space (size gate 20..2000 x 20..1000, 8K sources) against the oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/tools/ -> repository root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, synth
from bench import load_package
fl = load_package(); oracle = oracle_lib.load()
st = fl.State(device=0)
for (sh, sw, c, kw) in ((4320, 7680, 3, dict(w=2000, h=1000)), (4320, 7680, 3, dict(w=2000, h=1000, crop=True)), (4320, 7680, 4, dict(w=20, h=20)),
                        (1000, 2000, 3, dict(w=2000, h=1000, blur_sigma=20.0)), (300, 400, 3, dict(w=2000, h=1000)), (6000, 31, 1, dict(w=20, h=1000)),
                        (1080, 1920, 3, dict(w=2000, h=1000, grayscale=True, crop=True))):
    img = synth.uniform(sh, sw, c, index=sh)
    t0 = time.time(); got = st.process_pixels(img, fl.make_params(**kw)); t1 = time.time()
    want = oracle.process_pixels(img, arith=oracle_lib.ARITH_FMA, **kw); t2 = time.time()
    ref = oracle.process_pixels(img, arith=oracle_lib.ARITH_REF, **kw)
    print((sh, sw, c), kw, got.shape, "exact" if np.array_equal(got, want) else "MISMATCH", "maxdiff vs ref", int(np.abs(got.astype(int) - ref.astype(int)).max()),
          f"gpu {1e3 * (t1 - t0):.1f} ms oracle {1e3 * (t2 - t1):.0f} ms", flush=True)
    if kw.get("w") == 2000 and not kw.get("blur_sigma"):
        j = st.process_pixels(img, fl.make_params(quality=90, front_end=fl.FE_JPEG, **kw), capacity=2000 * 1000 * 8)
        print("   JPEG 2000x1000 q90:", len(j), "bytes,", "equal" if j == oracle.jpeg_encode(want, 90) else "MISMATCH", flush=True)
