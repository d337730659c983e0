"""One-off fuzzing of the device path against the oracle over a wide request space (not part of the test-suite):
    python tests/tools/fuzz_parity.py <cases> <seed>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/tools/ -> repository root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, parity, synth
from bench import load_package
fl = load_package()
oracle = oracle_lib.load()
n, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
st = fl.State(device=0)
bad = n_mfma = 0
for i in range(n):
    mode = rng.integers(0, 5)
    if mode == 0:   sh, sw = int(rng.integers(1, 40)), int(rng.integers(1, 40))            # tiny
    elif mode == 1: sh, sw = int(rng.integers(1, 12)), int(rng.integers(500, 3000))        # thin and wide
    elif mode == 2: sh, sw = int(rng.integers(500, 2500)), int(rng.integers(1, 12))        # thin and tall
    elif mode == 3: sh, sw = int(rng.integers(600, 1400)), int(rng.integers(800, 2200))    # photo sized
    else:           sh, sw = int(rng.integers(20, 700)), int(rng.integers(20, 900))
    c = int(rng.choice([1, 2, 3, 3, 4]))
    kw = dict(crop=bool(rng.integers(0, 2)), grayscale=bool(rng.integers(0, 4) == 0), inverse=bool(rng.integers(0, 4) == 0),
              fill=tuple(int(x) for x in rng.integers(0, 256, 3)), orientation=int(rng.choice([1, 1, 2, 4, 5, 7])))
    if rng.integers(0, 6):
        kw["w"], kw["h"] = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    if rng.integers(0, 5) == 0:
        kw["blur_sigma"] = float(rng.choice([10.0, 13.0, 20.0]))
    if rng.integers(0, 8) == 0:
        kw["filter"] = 1
    img = synth.uniform(sh, sw, c, index=i)
    okw = dict(kw)
    if okw.get("filter"): okw["filter"] = oracle_lib.FILTER_NEAREST
    try:
        fl.plan_output(fl.make_params(**kw), sw, sh, c)
    except fl.FanlinError as e:
        if e.status == fl.ERR_UNSUPPORTED:   # (a resize_to_fill whose covering size passes 2^31 bytes: refused by design, include/fanlin_gpu.h)
            continue
    try:
        # every result against the bars of the kernel that served it (tests/parity.py): bit-exact against the fused-order oracle and
        # <= 1 LSB from the reference arithmetic for the streaming / tiled / generic kernels, <= 1 LSB with a bounded rate for the
        # matrix-pipe kernel
        got, used_mfma = parity.device_pixels(fl, st, img, **kw)
        n_mfma += int(used_mfma)
        parity.check_pixels(oracle, got, img, used_mfma, **okw)
        ok = True
    except AssertionError as e:
        ok = False; print("BAR", str(e)[:200])
    except Exception as e:
        ok = False; print("EXC", repr(e)[:200])
    if not ok:
        bad += 1
        print("MISMATCH", i, (sh, sw, c), kw, flush=True)
print("cases", n, "served by the matrix-pipe kernel", n_mfma, "bad", bad)
