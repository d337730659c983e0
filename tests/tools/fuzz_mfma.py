"""One-off fuzzing of the matrix-pipe resample kernel against the oracle (not part of the test-suite): Rgb8 sources whose rows
are 16-byte aligned (1-4 channels), down-scale targets, crops, fills, band splits; every result must clear tests/parity.py's bars for the
kernel that served it.   python tests/tools/fuzz_mfma.py <cases> <seed>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, parity, synth
from bench import load_package

fl = load_package()
oracle = oracle_lib.load()
n, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
st = fl.State(device=0)
bad = used_n = 0
for i in range(n):
    c = int(rng.choice([3, 3, 3, 4, 1, 2]))
    sw = 16 * int(rng.integers(20, 260))                      # 320 .. 4144 pixels: c * sw is a multiple of 16
    sh = int(rng.integers(120, 2400))
    ratio = float(rng.uniform(3.0, 22.0))
    kw = dict(w=max(1, int(sw / ratio) + int(rng.integers(-3, 4))), h=max(1, int(sh / ratio) + int(rng.integers(-3, 4))),
              crop=bool(rng.integers(0, 2)), fill=tuple(int(x) for x in rng.integers(0, 256, 3)))
    if rng.integers(0, 4) == 0:
        kw["h"] = max(1, kw["h"] * int(rng.integers(2, 4)))   # letterboxed top and bottom (or a narrow crop)
    img = synth.uniform(sh, sw, c, index=i) if i % 3 else synth.photo(sh, sw, c, index=i)
    bands = str(int(rng.integers(1, 9)))
    st.debug_set("force_bands", int(bands))
    try:
        got, used = parity.device_pixels(fl, st, img, **kw)
        parity.check_pixels(oracle, got, img, used, **parity.oracle_kwargs(kw))
        used_n += int(used)
    except AssertionError as e:
        bad += 1
        print("MISMATCH", i, (sh, sw, c), kw, "bands", bands, str(e)[:160], flush=True)
    except Exception as e:
        bad += 1
        print("EXC", i, (sh, sw), kw, repr(e)[:200], flush=True)
print("cases", n, "matrix-pipe kernel", used_n, "bad", bad, flush=True)
sys.exit(1 if bad else 0)
