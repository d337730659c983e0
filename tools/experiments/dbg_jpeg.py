import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from conftest import load_package
import oracle_lib
from test_jpeg_decode import make_jpeg, CASES
fl = load_package(); o = oracle_lib.load()
st = fl.State(device=0)
data = make_jpeg(16, 16, 1, 95, 0, 0, index=32)
got = st.decode_jpeg(data).astype(int)[:, :, 0]; want = o.jpeg_decode(data).astype(int)[:, :, 0]
np.set_printoptions(linewidth=200)
print(got[:8, :16]); print(want[:8, :16])
