import sys, os, io
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
os.environ["FLGPU_DEBUG_JH"] = "1"; os.environ["FLGPU_DEVICE_HUFFMAN_ALWAYS"] = "1"; os.environ["FLGPU_DEVICE_HUFFMAN_MIN_BYTES"] = "0"
import numpy as np
from bench import load_package
import oracle_lib, synth
from test_jpeg_decode import make_jpeg
fl = load_package(); oracle = oracle_lib.load()
st = fl.State(device=0)
for case in [(64, 96, 3, 85, 2, 5), (37, 53, 3, 70, 0, 2), (200, 301, 3, 5, 0, 3), (1080, 1920, 3, 85, 2, 120), (720, 1280, 3, 92, 0, 1), (1000, 1500, 1, 80, 0, 33)]:
    h, w, c, q, sub, rst = case
    data = make_jpeg(h, w, c, q, sub, rst, index=h + rst)
    s0 = st.stats(); got = st.decode_jpeg(data); s1 = st.stats()
    want = oracle.jpeg_decode(data)
    print(case, "bytes", len(data), "device", s1["jpeg_device_huffman"] - s0["jpeg_device_huffman"], "retries", s1["jpeg_device_huffman_retries"] - s0["jpeg_device_huffman_retries"], "equal", np.array_equal(got, want), flush=True)
data = open("" + os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + "/tests/golden/lenna_reference.jpg", "rb").read()
print(fl.jpeg_info(data))
for k in range(3):
    s0 = st.stats(); got = st.decode_jpeg(data); s1 = st.stats()
    want = oracle.jpeg_decode(data)
    print("lenna: device", s1["jpeg_device_huffman"] - s0["jpeg_device_huffman"], "retries", s1["jpeg_device_huffman_retries"] - s0["jpeg_device_huffman_retries"], "equal", np.array_equal(got, want), "diff bytes", int((got != want).sum()), flush=True)
