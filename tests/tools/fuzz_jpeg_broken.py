"""One-off fuzzing of the device entropy decoder on BROKEN files (not part of the test-suite): Pillow-written baseline files with and without restart
intervals, damaged by byte flips in the segment, damaged / removed / doubled restart markers, flipped bits in the byte in front of a marker (padding or the
interval's last code words) and cuts.  Which side decodes the entropy-coded segment depends on how busy the host is, so every file must come out the same
either way: the same error, or the same pixels.   python tests/tools/fuzz_jpeg_broken.py <cases> <seed>"""
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from PIL import Image
import synth
from bench import load_package

fl = load_package()
n, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
st = fl.State(device=0)
st.debug_set("device_huffman_min_bytes", 0)
st.debug_set("device_huffman_always", 1)


def outcome(blob):
    try:
        return 0, st.decode_jpeg(blob)
    except fl.FanlinError as e:
        return e.status, None


bad = pictures = 0
for i in range(n):
    h, w = int(rng.integers(40, 700)), int(rng.integers(40, 900))
    c = int(rng.choice([3, 3, 1]))
    kw = dict(quality=int(rng.integers(30, 96)))
    if c == 3: kw["subsampling"] = int(rng.integers(0, 3))
    rst = int(rng.choice([0, 1, 2, 4, 9, 40]))
    if rst: kw["restart_marker_blocks"] = rst
    img = synth.photo(h, w, c, index=i)
    buf = io.BytesIO()
    Image.fromarray(img[:, :, 0] if c == 1 else img).save(buf, "JPEG", **kw)
    data = bytearray(buf.getvalue())
    sos = data.find(b"\xff\xda") + (14 if c == 3 else 10)
    marks = [k for k in range(sos, len(data) - 1) if data[k] == 0xFF and 0xD0 <= data[k + 1] <= 0xD7]
    kind = int(rng.integers(0, 7 if marks else 3))
    d = bytearray(data)
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))): d[int(rng.integers(sos, len(d) - 2))] = int(rng.integers(0, 255))
    elif kind == 1: d = d[:int(rng.integers(sos + 8, len(d)))] + b"\xff\xd9"
    elif kind == 2: d[int(rng.integers(sos, len(d) - 2))] ^= 1 << int(rng.integers(0, 8))
    else:
        m = marks[int(rng.integers(0, len(marks)))]
        if kind == 3: del d[m:m + 2]
        elif kind == 4: d[m + 1] = 0xD0 + (d[m + 1] - 0xD0 + int(rng.integers(1, 8))) % 8
        elif kind == 5: d[m:m] = d[m:m + 2]
        else: d[m - 1] ^= 1 << int(rng.integers(0, 8))
    d = bytes(d)
    dev = outcome(d)
    with st.switches(host_huffman=1):
        host = outcome(d)
    pictures += dev[1] is not None
    if dev[0] != host[0] or (dev[1] is not None and not np.array_equal(dev[1], host[1])):
        bad += 1
        print("DIFFERENT", i, (h, w, c), kw, "kind", kind, "status", dev[0], host[0], flush=True)
print("cases", n, "came out as pictures", pictures, "bad", bad, "retries", st.stats()["jpeg_device_huffman_retries"], flush=True)
sys.exit(1 if bad else 0)
