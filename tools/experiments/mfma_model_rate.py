"""How many bytes of the device result differ from the numpy model of the matrix-pipe kernel's arithmetic (tests/parity.py)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
fl = importlib.import_module("fanlin-rs_amd")
import oracle_lib, parity, synth
oracle = oracle_lib.load()
with fl.State() as st:
    for (h, w, c, rw, rh) in [(1080, 1920, 3, 300, 169), (1080, 1920, 3, 352, 198), (1080, 1920, 3, 320, 180), (1080, 1920, 4, 300, 169), (2160, 3840, 1, 640, 360)]:
        for kind in ("uniform", "photo"):
            img = getattr(synth, kind)(h, w, c, index=3)
            got, used = parity.device_pixels(fl, st, img, w=rw, h=rh)
            model = parity.mfma_model(fl, img, rw, rh)
            ref = oracle.process_pixels(img, rw, rh, arith=oracle_lib.ARITH_REF)
            dm = np.abs(got.astype(int) - model.astype(int)); dr = np.abs(got.astype(int) - ref.astype(int)); mr = np.abs(model.astype(int) - ref.astype(int))
            print(f"{w}x{h}x{c} -> {rw}x{rh} {kind}: matrix-pipe {used}; device vs model: max {dm.max()}, {100 * (dm > 0).mean():.4f} % differ; device vs reference arithmetic: max {dr.max()}, {100 * (dr > 0).mean():.4f} %; model vs reference: {100 * (mr > 0).mean():.4f} %", flush=True)
