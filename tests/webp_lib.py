"""ctypes access to the system's libwebp encoder -- the C library under the reference's `webp` crate
(webp 0.3.0 -> libwebp-sys, reference src/handler.rs:295-297).  Test infrastructure only.

Two ways to a lossy WebP file:
  encode_rgba(rgba, q)        what `webp::Encoder::from_image(&img).encode(q)` does: WebPPictureImportRGBA (use_argb = 1),
                              default config with `quality = q`, WebPEncode (which converts ARGB -> YUV420 itself);
  encode_planes(y, u, v, q)   the split this repository proposes: the colour front end ran elsewhere (HIP kernel or
                              oracle), libwebp receives a YUV420 picture (use_argb = 0) and only predicts / entropy-codes.
If the planes equal libwebp's own conversion the two files are byte-identical."""
import ctypes as C
import ctypes.util

import numpy as np


class WebPPicture(C.Structure):
    _fields_ = [
        ("use_argb", C.c_int), ("colorspace", C.c_int), ("width", C.c_int), ("height", C.c_int),
        ("y", C.POINTER(C.c_uint8)), ("u", C.POINTER(C.c_uint8)), ("v", C.POINTER(C.c_uint8)),
        ("y_stride", C.c_int), ("uv_stride", C.c_int),
        ("a", C.POINTER(C.c_uint8)), ("a_stride", C.c_int), ("pad1", C.c_uint32 * 2),
        ("argb", C.POINTER(C.c_uint32)), ("argb_stride", C.c_int), ("pad2", C.c_uint32 * 3),
        ("writer", C.c_void_p), ("custom_ptr", C.c_void_p), ("extra_info_type", C.c_int),
        ("extra_info", C.c_void_p), ("stats", C.c_void_p), ("error_code", C.c_int),
        ("progress_hook", C.c_void_p), ("user_data", C.c_void_p), ("pad3", C.c_uint32 * 3),
        ("pad4", C.c_void_p), ("pad5", C.c_void_p), ("pad6", C.c_uint32 * 8),
        ("memory_", C.c_void_p), ("memory_argb_", C.c_void_p), ("pad7", C.c_void_p * 2),
    ]


class WebPMemoryWriter(C.Structure):
    _fields_ = [("mem", C.POINTER(C.c_uint8)), ("size", C.c_size_t), ("max_size", C.c_size_t), ("pad", C.c_uint32 * 1)]


_lib, _abi = None, None


def load():
    global _lib, _abi
    if _lib is None:
        path = ctypes.util.find_library("webp") or "libwebp.so.7"
        try:
            lib = C.CDLL(path)
        except OSError:
            return None
        for cand in (0x020f, 0x020e, 0x0210, 0x0209):
            if lib.WebPPictureInitInternal(C.byref(WebPPicture()), cand) == 1:
                _abi = cand
                break
        if _abi is None:
            return None
        lib.WebPMemoryWriterInit.argtypes = [C.POINTER(WebPMemoryWriter)]
        lib.WebPMemoryWriterClear.argtypes = [C.POINTER(WebPMemoryWriter)]
        lib.WebPConfigInitInternal.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int]
        lib.WebPEncode.argtypes = [C.c_void_p, C.POINTER(WebPPicture)]
        _lib = lib
    return _lib


def _config(q):
    cfg = (C.c_uint8 * 512)()                      # WebPConfig, left opaque: only `quality` (second field) is touched
    assert _lib.WebPConfigInitInternal(cfg, 0, C.c_float(75.0), _abi) == 1   # WebPConfig::new(): default preset
    C.cast(cfg, C.POINTER(C.c_float))[1] = float(q)                          # config.quality = q; lossless stays 0
    return cfg


def _encode(pic, q):
    w = WebPMemoryWriter()
    _lib.WebPMemoryWriterInit(C.byref(w))
    pic.writer = C.cast(_lib.WebPMemoryWrite, C.c_void_p)
    pic.custom_ptr = C.cast(C.pointer(w), C.c_void_p)
    ok = _lib.WebPEncode(_config(q), C.byref(pic))
    assert ok == 1, f"WebPEncode failed, error {pic.error_code}"
    data = bytes(bytearray(w.mem[: w.size]))
    _lib.WebPMemoryWriterClear(C.byref(w))
    _lib.WebPPictureFree(C.byref(pic))
    return data


def encode_rgba(rgba, q):
    assert load() is not None
    h, w, c = rgba.shape
    assert c == 4
    pic = WebPPicture()
    assert _lib.WebPPictureInitInternal(C.byref(pic), _abi) == 1
    pic.use_argb, pic.width, pic.height = 1, w, h
    buf = np.ascontiguousarray(rgba)
    assert _lib.WebPPictureImportRGBA(C.byref(pic), buf.ctypes.data_as(C.POINTER(C.c_uint8)), w * 4) == 1
    return _encode(pic, q)


def encode_planes(y, u, v, q, a=None):
    """a: the alpha plane of a translucent picture (WEBP_YUV420A), None for opaque ones."""
    assert load() is not None
    h, w = y.shape
    pic = WebPPicture()
    assert _lib.WebPPictureInitInternal(C.byref(pic), _abi) == 1
    pic.use_argb, pic.width, pic.height = 0, w, h
    if a is not None:
        pic.colorspace = 4            # WEBP_YUV420A = WEBP_YUV420 | WEBP_CSP_ALPHA_BIT
    assert _lib.WebPPictureAlloc(C.byref(pic)) == 1
    planes = [(y, pic.y, pic.y_stride), (u, pic.u, pic.uv_stride), (v, pic.v, pic.uv_stride)]
    if a is not None:
        planes.append((a, pic.a, pic.a_stride))
    for plane, ptr, stride in planes:
        dst = np.ctypeslib.as_array(ptr, shape=(plane.shape[0] * stride,))
        for r in range(plane.shape[0]):
            dst[r * stride: r * stride + plane.shape[1]] = plane[r]
    return _encode(pic, q)
