"""Development check of the matrix-pipe resample kernel against the oracle (reference arithmetic), on the GPU box.
   python tools/experiments/mfma_check.py            : a few geometries, max difference and rate of off-by-one bytes"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib

fl = importlib.import_module("fanlin-rs_amd")
import oracle_lib

oracle = oracle_lib.load()
rng = np.random.default_rng(7)
cases = [(1080, 1920, 300, 200, False), (1080, 1920, 300, 200, True), (720, 1280, 160, 90, False), (1080, 1920, 300, 169, False),
         (2160, 3840, 640, 360, False), (600, 800, 100, 100, False), (1080, 1920, 480, 270, False), (333, 1024, 90, 30, False)]
bad = 0
with fl.State() as st:
    for (h, w, ow, oh, crop) in cases:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        p = fl.make_params(w=ow, h=oh, crop=crop)
        st.reset_stats()
        got = st.process_pixels(img, p)
        want = oracle.process_pixels(img, ow, oh, crop=crop)
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        again = st.process_pixels(img, p)
        print(f"{w}x{h} -> {ow}x{oh} crop={crop}: out {got.shape}, max diff {d.max()}, off-by-one {100.0 * (d > 0).mean():.3f} %, "
              f"repeatable {np.array_equal(got, again)}, launches {st.stats()['resample_launches']}", flush=True)
        if d.max() > 1:
            bad += 1
            ys, xs, cs = np.nonzero(d > 1)
            print("   first bad:", list(zip(ys[:8], xs[:8], cs[:8])), "rows", sorted(set(ys))[:20], "cols", sorted(set(xs))[:20])
sys.exit(1 if bad else 0)
