"""One-off stress of the request queue with JPEG FILES as sources (not part of the test-suite): good files with and without restart intervals and damaged
ones, sent by many threads at once through flgpu_transform (device entropy decoding under load, host retries for the damaged ones, batches that mix all of
them), then the same requests one at a time.  Every request must come out the same both times: the same error, or the same bytes.
   python tests/tools/fuzz_jpeg_concurrent.py <requests> <threads> <seed> [always|- [query]]   (a query with a large target, e.g. w=1000&h=800,
sends streams longer than the 32 KB the queue fetches in front of its wait)"""
import io
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from PIL import Image
import synth
from bench import load_package

fl = load_package()
n, nthreads, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
files = []
for i in range(n):
    h, w = int(rng.integers(100, 900)), int(rng.integers(100, 1200))
    c = int(rng.choice([3, 3, 1]))
    kw = dict(quality=int(rng.integers(40, 95)))
    if c == 3: kw["subsampling"] = int(rng.integers(0, 3))
    rst = int(rng.choice([0, 0, 1, 3, 16, 75]))
    if rst: kw["restart_marker_blocks"] = rst
    img = synth.photo(h, w, c, index=i)
    buf = io.BytesIO()
    Image.fromarray(img[:, :, 0] if c == 1 else img).save(buf, "JPEG", **kw)
    d = bytearray(buf.getvalue())
    if rng.integers(0, 4) == 0:   # a quarter of the files damaged
        sos = d.find(b"\xff\xda") + 12
        for _ in range(int(rng.integers(1, 4))): d[int(rng.integers(sos, len(d) - 2))] = int(rng.integers(0, 255))
    files.append(bytes(d))

st = fl.State(device=0)
st.debug_set("device_huffman_min_bytes", 0)
QUERY = sys.argv[5] if len(sys.argv) > 5 else "w=120&h=90"
if len(sys.argv) > 4 and sys.argv[4] != "-": st.debug_set("device_huffman_always", 1)   # a fourth argument: every file the device takes goes to it, busy CPUs or not


def run(i):
    try:
        out = st.process_jpeg(files[i], QUERY)
        return (0, bytes(out[2]) if isinstance(out[2], (bytes, bytearray)) else np.asarray(out[2]).tobytes())
    except fl.FanlinError as e:
        return (e.status, b"")


res = [None] * n
def worker(t):
    for i in range(t, n, nthreads): res[i] = run(i)
ts = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
for t in ts: t.start()
for t in ts: t.join()
s = st.stats()
bad = 0
for i in range(n):
    alone = run(i)
    if alone != res[i]:
        bad += 1
        print("DIFFERENT", i, "concurrent status", res[i][0], "alone", alone[0], "bytes", len(res[i][1]), len(alone[1]), flush=True)
print("requests", n, "threads", nthreads, "on the device (concurrent phase)", s["jpeg_device_huffman"], "retried", s["jpeg_device_huffman_retries"], "errors", sum(1 for r in res if r[0]), "bad", bad, flush=True)
sys.exit(1 if bad else 0)
