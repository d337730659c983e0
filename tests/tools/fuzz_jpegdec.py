"""One-off fuzzing of the device entropy decoder (csrc/fl_jpeghuff_dev.hip) against the oracle's decoder (not part of the test-suite): Pillow-written
baseline files of random sizes (8 .. 2600 pixels a side), 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0, quality 3 .. 98, with and without restart
intervals of 1 .. 500 MCUs, Annex K tables or optimised ones
(other code lengths: the 12-bit lookahead, the end-of-block fusing and the long-code search all see tables they were not tuned on), photographs, noise and
flat pictures.  Every file must come back as the oracle decoder's pixels, bit for bit; which files the device decoded and which ended in the host retry is reported.
   python tests/tools/fuzz_jpegdec.py <cases> <seed>"""
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from PIL import Image, ImageFile
ImageFile.MAXBLOCK = 1 << 25   # (optimize=True needs the whole file in one encoder buffer)
import oracle_lib, synth
from bench import load_package

fl = load_package()
oracle = oracle_lib.load()
n, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
os.environ["FLGPU_DEVICE_HUFFMAN_MIN_BYTES"] = "0"; os.environ["FLGPU_DEVICE_HUFFMAN_ALWAYS"] = "1"   # (flgpu_create seeds the switches from these; libraries of round 4 read them per batch)
if len(sys.argv) > 3: os.environ["FLGPU_DEBUG_JH"] = "1"                                                  # a third argument: the device's error words on stderr
st = fl.State(device=0)
bad = n_device = n_retried = 0
names = ("noise", "photo", "photo", "flat")
for i in range(n):
    c = int(rng.choice([3, 3, 3, 1]))
    h, w = (int(rng.integers(8, 2600)), int(rng.integers(8, 2600))) if rng.integers(0, 3) else (int(rng.integers(1, 64)), int(rng.integers(1, 64)))
    if h * w > 3_000_000: h = max(8, 3_000_000 // w)
    kind = int(rng.integers(0, 4))
    img = synth.uniform(h, w, c, index=i) if kind == 0 else synth.photo(h, w, c, index=i) if kind < 3 else np.full((h, w, c), int(rng.integers(0, 256)), np.uint8)
    kw = dict(quality=int(rng.integers(3, 99)), optimize=bool(rng.integers(0, 2)))
    if c == 3: kw["subsampling"] = int(rng.integers(0, 3))
    if rng.integers(0, 3) == 0: kw["restart_marker_blocks"] = int(rng.choice([1, 2, 3, 7, 20, 64, 120, 500]))   # restart intervals (round 5: on the device)
    buf = io.BytesIO()
    Image.fromarray(img[:, :, 0] if c == 1 else img).save(buf, "JPEG", **kw)
    data = buf.getvalue()
    if os.environ.get("FUZZ_ONLY") and str(i) not in os.environ["FUZZ_ONLY"].split(","): continue
    try:
        s0 = st.stats()
        got = st.decode_jpeg(data)
        s1 = st.stats()
        on_device = s1["jpeg_device_huffman"] - s0["jpeg_device_huffman"]
        retried = s1["jpeg_device_huffman_retries"] - s0["jpeg_device_huffman_retries"]
        n_device += on_device; n_retried += retried
        if retried: print("retried on the host:", i, names[kind], (h, w, c), kw, len(data), "bytes", flush=True)
        want = oracle.jpeg_decode(data)
        assert got.shape == want.shape and np.array_equal(got, want), "pixels differ: %d bytes" % int((got != want).sum())
    except AssertionError as e:
        bad += 1
        print("MISMATCH", i, names[kind], (h, w, c), kw, len(data), str(e)[:160], flush=True)
    except Exception as e:
        bad += 1
        print("EXC", i, (h, w, c), kw, repr(e)[:200], flush=True)
# (what ends in the host retry, as in round 4: noise at quality >= 92, whose blocks are longer than a subsequence; one-colour pictures and photographs below
# quality ~10 are left to the host decoder by the staging step: fewer than 7 bits per block)
print("cases", n, "entropy-decoded on the device", n_device, "of them retried on the host", n_retried, "bad", bad, flush=True)
sys.exit(1 if bad else 0)
