"""Small and odd-shaped sources through the matrix-pipe kernel (development check): narrow pictures (one strip, rows
shorter than a wave's columns), few rows (one K-block), huge ratios."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
fl = importlib.import_module("fanlin-rs_amd")
import oracle_lib, parity, synth
oracle = oracle_lib.load()
bad = used_n = 0
cases = [(64, 64, 3, 8, 8), (48, 64, 3, 12, 9), (33, 80, 3, 10, 4), (200, 32, 3, 4, 25), (17, 1600, 3, 100, 1), (1000, 16, 3, 2, 125),
         (40, 48, 4, 6, 5), (31, 64, 1, 8, 4), (32, 64, 1, 16, 8), (65, 128, 2, 16, 8), (2000, 3008, 3, 47, 31), (3000, 4000, 3, 25, 19),
         (96, 128, 3, 32, 24), (128, 256, 3, 60, 30), (1080, 1920, 3, 30, 17), (1080, 1920, 3, 10, 6)]
with fl.State() as st:
    for (h, w, c, ow, oh) in cases:
        img = synth.uniform(h, w, c, index=h + w)
        try:
            got, used = parity.device_pixels(fl, st, img, w=ow, h=oh)
            parity.check_pixels(oracle, got, img, used, w=ow, h=oh)
            used_n += int(used)
            print((h, w, c), "->", (ow, oh), "matrix-pipe" if used else "streaming/generic", "ok", flush=True)
        except AssertionError as e:
            bad += 1
            print((h, w, c), "->", (ow, oh), "MISMATCH", str(e)[:120], flush=True)
print("bad", bad, "matrix-pipe", used_n)
sys.exit(1 if bad else 0)
