"""One-off fuzzing of the window-tile matrix-pipe kernel (csrc/fl_wtile.h) against the oracle (not part of the test-suite): sources
of 1-4 channels with ANY width (odd pitches, unaligned rows), mild down-scales (and, with FLGPU_WTILE_ALWAYS=1, up-scales), crops,
fills, band splits, blurs of sigma 0.3 .. 20 alone and behind a resize; every result must clear tests/parity.py's bars for the kernel
that served it, and equal itself when run again.   python tests/tools/fuzz_wtile.py <cases> <seed>"""
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
    sw = int(rng.integers(8, 1400))
    sh = int(rng.integers(8, 1000))
    kind = int(rng.integers(0, 4))
    kw = {}
    if kind != 1:                                             # a resize (ratio 0.6 .. 3.3: the kernel's range and a bit on either side)
        ratio = float(rng.uniform(0.6, 3.3))
        kw.update(w=int(min(2000, max(20, sw / ratio + rng.integers(-3, 4)))), h=int(min(1000, max(20, sh / ratio + rng.integers(-3, 4)))),
                  crop=bool(rng.integers(0, 2)), fill=tuple(int(x) for x in rng.integers(0, 256, 3)))
    if kind != 1 and rng.integers(0, 5) == 0:
        kw["inverse"] = True
    if kind in (1, 2):                                        # a blur, alone or behind the resize
        kw["blur_sigma"] = float(rng.choice([0.3, 0.8, 1.5, 3.0, 7.0, 10.0, 14.5, 20.0]))
    img = synth.uniform(sh, sw, c, index=i) if i % 3 else synth.photo(sh, sw, c, index=i)
    bands = str(int(rng.integers(1, 9)))
    st.debug_set("force_bands", int(bands))
    try:
        before = st.stats()["wtile_launches"]
        got, used = parity.device_pixels(fl, st, img, **kw)
        parity.check_pixels(oracle, got, img, used, **parity.oracle_kwargs(kw))
        used_n += int(st.stats()["wtile_launches"] > before)
        again, _ = parity.device_pixels(fl, st, img, **kw)
        assert np.array_equal(got, again), "not repeatable"
    except AssertionError as e:
        bad += 1
        print("MISMATCH", i, (sh, sw, c), kw, "bands", bands, str(e)[:160], flush=True)
    except Exception as e:
        bad += 1
        print("EXC", i, (sh, sw, c), kw, repr(e)[:200], flush=True)
print("cases", n, "window-tile kernel", used_n, "bad", bad, flush=True)
sys.exit(1 if bad else 0)
