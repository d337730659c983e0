#!/usr/bin/env python3
"""Generates tests/golden/webp_yuv420_libwebp.npz from the SYSTEM libwebp (libwebp.so.7) through ctypes.

The reference reaches libwebp through webp 0.3.0 / libwebp-sys 0.9.6 (vendored C, Cargo.lock:4019,2156):
Encoder::from_image -> WebPPictureImportRGBA (use_argb = 1) -> WebPEncode -> WebPPictureARGBToYUVA.
This script drives the same two C entry points of the distribution's libwebp build and stores the
Y/U/V planes it produces for a few small seeded RGBA pictures, so the CPU oracle's restatement of
that colour front end is pinned against real libwebp output (the libwebp version differs from the
vendored one; the RGB->YUV420 code in picture_csp_enc.c / dsp/yuv.h is the same in both).
"""
import ctypes as C
import ctypes.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402


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


def libwebp_yuv420(lib, abi, rgba):
    h, w, _ = rgba.shape
    pic = WebPPicture()
    assert lib.WebPPictureInitInternal(C.byref(pic), abi) == 1
    pic.use_argb, pic.width, pic.height = 1, w, h
    buf = np.ascontiguousarray(rgba)
    assert lib.WebPPictureImportRGBA(C.byref(pic), buf.ctypes.data_as(C.POINTER(C.c_uint8)), w * 4) == 1
    assert lib.WebPPictureARGBToYUVA(C.byref(pic), 0) == 1  # WEBP_YUV420
    cw, ch = (w + 1) // 2, (h + 1) // 2
    y = np.array([[pic.y[r * pic.y_stride + c] for c in range(w)] for r in range(h)], np.uint8)
    u = np.array([[pic.u[r * pic.uv_stride + c] for c in range(cw)] for r in range(ch)], np.uint8)
    v = np.array([[pic.v[r * pic.uv_stride + c] for c in range(cw)] for r in range(ch)], np.uint8)
    a = None
    if pic.a:  # WEBP_YUV420A: some pixel is not opaque
        a = np.array([[pic.a[r * pic.a_stride + c] for c in range(w)] for r in range(h)], np.uint8)
    lib.WebPPictureFree(C.byref(pic))
    return y, u, v, a


def main():
    path = ctypes.util.find_library("webp") or "libwebp.so.7"
    lib = C.CDLL(path)
    lib.WebPGetEncoderVersion.restype = C.c_int
    ver = lib.WebPGetEncoderVersion()
    abi = None
    for cand in (0x020f, 0x020e, 0x0210, 0x0209):  # WEBP_ENCODER_ABI_VERSION of 1.1-1.4 / 1.0 / 1.5 / 0.5
        pic = WebPPicture()
        if lib.WebPPictureInitInternal(C.byref(pic), cand) == 1:
            abi = cand
            break
    assert abi is not None, "no matching encoder ABI"
    out = {"libwebp_version": np.array([ver >> 16, (ver >> 8) & 255, ver & 255])}
    cases = {"u_16x12": synth.uniform(12, 16, 3, index=900), "u_17x13": synth.uniform(13, 17, 3, index=901),
             "p_40x30": synth.photo(30, 40, 3, index=902), "u_1x1": synth.uniform(1, 1, 3, index=903),
             "u_2x5": synth.uniform(5, 2, 3, index=904), "edges": synth.edges(8, 8, 3)["checker"]}
    for name, rgb in cases.items():
        rgba = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
        y, u, v, a = libwebp_yuv420(lib, abi, rgba)
        assert a is None
        out[name + "_rgba"], out[name + "_y"], out[name + "_u"], out[name + "_v"] = rgba, y, u, v
    # translucent pictures: libwebp weights the chroma of partly transparent 2x2 blocks by alpha and keeps an alpha plane
    rng = np.random.default_rng(0xA1FA)
    for name, (h, w) in {"a_16x12": (12, 16), "a_17x13": (13, 17), "a_1x1": (1, 1), "a_2x5": (5, 2), "a_40x30": (30, 40)}.items():
        rgb = synth.uniform(h, w, 3, index=950 + h)
        alpha = rng.integers(0, 256, (h, w, 1), dtype=np.uint8)
        alpha[rng.random(alpha.shape) < 0.3] = 255
        alpha[rng.random(alpha.shape) < 0.2] = 0
        rgba = np.concatenate([rgb, alpha], axis=2)
        y, u, v, a = libwebp_yuv420(lib, abi, rgba)
        assert a is not None
        out[name + "_rgba"], out[name + "_y"], out[name + "_u"], out[name + "_v"], out[name + "_a"] = rgba, y, u, v, a
    dst = os.path.join(ROOT, "tests", "golden", "webp_yuv420_libwebp.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, "libwebp", ".".join(str(int(x)) for x in out["libwebp_version"]), "abi", hex(abi))


if __name__ == "__main__":
    main()
