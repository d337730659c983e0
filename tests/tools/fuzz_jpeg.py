"""One-off fuzzing of the device JPEG encoder against the oracle encoder (byte equality):
    python tests/tools/fuzz_jpeg.py <cases> <seed>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/tools/ -> repository root
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, synth
from bench import load_package
fl = load_package()
oracle = oracle_lib.load()
n, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
st = fl.State(device=0)
bad = 0
for i in range(n):
    h, w = (int(rng.integers(1, 40)), int(rng.integers(1, 40))) if rng.integers(0, 3) == 0 else (int(rng.integers(8, 500)), int(rng.integers(8, 700)))
    if rng.integers(0, 12) == 0: h, w = int(rng.integers(900, 1100)), int(rng.integers(1800, 2000))   # > one LDS window of bit stream
    c = int(rng.choice([1, 2, 3, 4, 4]))
    q = int(rng.choice([1, 5, 25, 50, 75, 75, 90, 100]))
    kind = rng.integers(0, 4)
    img = synth.uniform(h, w, c, index=i) if kind == 0 else synth.photo(h, w, c, index=i)
    if kind == 2: img = (img // 64 * 64).astype(np.uint8)                 # flat regions: DC-only blocks, long zero runs
    if kind == 3: img[::2, ::2] = 255 - img[::2, ::2]                      # high frequency everywhere
    got = st.process_pixels(img, fl.make_params(quality=q, front_end=fl.FE_JPEG), capacity=h * w * 16 + 8192)
    want = oracle.jpeg_encode(img, q)
    if got != want:
        bad += 1
        print("MISMATCH", i, (h, w, c), q, kind, len(got), len(want), flush=True)
print("cases", n, "bad", bad)
