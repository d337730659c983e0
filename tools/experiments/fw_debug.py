"""Development aid: where does the full-width matrix-pipe kernel differ from the oracle?  (error map by 16-row tile and 20-column block)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib
fl = importlib.import_module("fanlin-rs_amd")
import oracle_lib
oracle = oracle_lib.load()
rng = np.random.default_rng(7)
h, w, c, ow, oh = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (1080, 1920, 3, 300, 200))]
img = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
with fl.State() as st:
    for bands in ("1", None):
        st.debug_set("force_bands", int(bands) if bands else 0)
        got = st.process_pixels(img, fl.make_params(w=ow, h=oh))
        want = oracle.process_pixels(img, ow, oh)
        d = np.abs(got.astype(np.int32) - want.astype(np.int32)).max(axis=2)
        print("bands", bands, "shape", got.shape, "max", d.max())
        ys = np.nonzero(d.max(axis=1) > 1)[0]; xs = np.nonzero(d.max(axis=0) > 1)[0]
        print(" bad rows", ys.min() if len(ys) else None, ys.max() if len(ys) else None, len(ys), " bad cols", xs.min() if len(xs) else None, xs.max() if len(xs) else None, len(xs))
        for y0 in range(0, d.shape[0], 16):
            print("  rows %3d: " % y0 + " ".join("%3d" % d[y0:y0 + 16, x0:x0 + 20].max() for x0 in range(0, d.shape[1], 20)))
        if len(ys):
            y = ys[0]
            print(" row", y, "got ", got[y, :12].reshape(-1)[:36]); print(" row", y, "want", want[y, :12].reshape(-1)[:36])
