"""ctypes binding of oracle/libwtile_model.so: the window-tile matrix-pipe kernel's tables (built by the product's own table
builder) run on the host, operand for operand (oracle/wtile_model.cpp).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libwtile_model.so")
KEYS = ("m_tiles", "n_tiles", "strips", "hs", "nslot", "nkmax", "lds_bytes", "table_words")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libwtile_model.so"], check=True, capture_output=True)
        _lib = C.CDLL(LIB)
        _lib.wtile_model_run.restype = C.c_int
        _lib.wtile_model_run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_void_p, C.POINTER(C.c_uint32)]
    return _lib


def run(img=None, rw: int = 0, rh: int = 0, blur_sigma: float = 0.0, shape=None):
    """resize_exact of `img` (H x W x C uint8) to rw x rh, or its Gaussian blur, through the kernel's tables on the host: returns
    (pixels, info) or None if the geometry does not fit the kernel.  With img = None and shape = (H, W, C): (None, plan info)."""
    lib = load()
    info = (C.c_uint32 * 8)()
    if img is None:
        h, w, c = shape
        if not lib.wtile_model_run(None, w, h, c, rw, rh, blur_sigma, None, info):
            return None
        return None, dict(zip(KEYS, (int(x) for x in info)))
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    if blur_sigma > 0.0:
        rw, rh = w, h
    out = np.zeros((rh, rw, c), np.uint8)
    if not lib.wtile_model_run(img.ctypes.data, w, h, c, rw, rh, blur_sigma, out.ctypes.data, info):
        return None
    return out, dict(zip(KEYS, (int(x) for x in info)))
