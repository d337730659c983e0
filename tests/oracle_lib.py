"""ctypes binding of the CPU oracle (oracle/libfanlin_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libfanlin_oracle.so")

ARITH_REF, ARITH_FMA = 0, 1
FILTER_LANCZOS3, FILTER_GAUSSIAN, FILTER_NEAREST, FILTER_TRIANGLE = 0, 1, 2, 3


class fo_image(C.Structure):
    _fields_ = [("w", C.c_uint32), ("h", C.c_uint32), ("c", C.c_uint32), ("px", C.POINTER(C.c_uint8))]


class fo_params(C.Structure):
    _fields_ = [("has_dims", C.c_int), ("w", C.c_uint32), ("h", C.c_uint32), ("fill", C.c_uint8 * 3),
                ("crop", C.c_int), ("blur_sigma", C.c_float), ("grayscale", C.c_int), ("inverse", C.c_int), ("orientation", C.c_int), ("filter", C.c_int)]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.fo_build_weights.restype = C.c_long
        lib.fo_free.argtypes = [C.c_void_p]

    @staticmethod
    def _img(a):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        im = fo_image(a.shape[1], a.shape[0], a.shape[2], a.ctypes.data_as(C.POINTER(C.c_uint8)))
        return im, a

    def _take(self, im):
        n = im.w * im.h * im.c
        out = np.ctypeslib.as_array(im.px, shape=(n,)).copy().reshape(im.h, im.w, im.c) if n else np.zeros((im.h, im.w, im.c), np.uint8)
        self.lib.fo_free(im.px)
        return out

    def resize_dimensions(self, w, h, nw, nh, fill):
        ow, oh = C.c_uint32(), C.c_uint32()
        self.lib.fo_resize_dimensions(w, h, nw, nh, int(fill), C.byref(ow), C.byref(oh))
        return ow.value, oh.value

    def build_weights(self, in_size, out_size, filt=FILTER_LANCZOS3, sigma=0.0):
        left = (C.c_uint32 * out_size)()
        count = (C.c_uint32 * out_size)()
        off = (C.c_uint32 * (out_size + 1))()
        ratio = max(in_size / out_size, 1.0)
        support = 3.0 if filt == FILTER_LANCZOS3 else 2.0 * sigma
        cap = int(out_size * (2 * support * ratio + 4)) + 16
        w = (C.c_float * cap)()
        n = self.lib.fo_build_weights(in_size, out_size, filt, C.c_float(sigma), left, count, off, w, C.c_size_t(cap))
        assert n >= 0
        return (np.array(left, dtype=np.uint32), np.array(count, dtype=np.uint32), np.array(off, dtype=np.uint32),
                np.array(w[:n], dtype=np.float32))

    def process_pixels(self, image, w=None, h=None, fill=(32, 32, 32), crop=False, blur_sigma=0.0, grayscale=False,
                       inverse=False, orientation=0, filter=FILTER_LANCZOS3, arith=ARITH_REF):
        im, keep = self._img(image)
        p = fo_params()
        p.has_dims = int(w is not None and h is not None)
        p.w, p.h = (w or 0), (h or 0)
        p.fill[0], p.fill[1], p.fill[2] = fill
        p.crop, p.blur_sigma, p.grayscale, p.inverse = int(crop), blur_sigma, int(grayscale), int(inverse)
        p.orientation = orientation
        p.filter = filter
        out = fo_image()
        rc = self.lib.fo_process_pixels(C.byref(im), C.byref(p), arith, C.byref(out))
        assert rc == 0
        return self._take(out)

    def resize_exact(self, image, nw, nh, arith=ARITH_REF, filt=FILTER_LANCZOS3):
        im, keep = self._img(image)
        out = fo_image()
        assert self.lib.fo_resize_exact(C.byref(im), nw, nh, filt, arith, C.byref(out)) == 0
        return self._take(out)

    def blur(self, image, sigma, arith=ARITH_REF):
        im, keep = self._img(image)
        out = fo_image()
        assert self.lib.fo_blur(C.byref(im), C.c_float(sigma), arith, C.byref(out)) == 0
        return self._take(out)

    # ---- JPEG encoder back half (fanlin_oracle_jpeg.c) ----
    def jpeg_qtables(self, quality):
        a, b = (C.c_uint8 * 64)(), (C.c_uint8 * 64)()
        self.lib.fo_jpeg_qtables(int(quality), a, b)
        return np.array(a, np.uint8), np.array(b, np.uint8)

    def jpeg_fdct(self, samples):
        s = np.ascontiguousarray(samples, dtype=np.uint8).reshape(64)
        out = np.zeros(64, np.int32)
        self.lib.fo_jpeg_fdct(s.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out.reshape(8, 8)

    def jpeg_coefficients(self, image, quality):
        im, keep = self._img(image)
        bx, by = (im.w + 7) // 8, (im.h + 7) // 8
        out = np.zeros((by * bx * 3, 64), np.int16)
        assert self.lib.fo_jpeg_coefficients(C.byref(im), int(quality), out.ctypes.data_as(C.c_void_p)) == 0
        return out

    def jpeg_header(self, w, h, quality):
        buf = np.zeros(1024, np.uint8)
        self.lib.fo_jpeg_header.restype = C.c_size_t
        n = self.lib.fo_jpeg_header(C.c_uint32(w), C.c_uint32(h), int(quality), buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
        return buf[:n].tobytes()

    def jpeg_encode(self, image, quality):
        im, keep = self._img(image)
        cap = 1024 + im.w * im.h * 8 + 4096
        buf = np.zeros(cap, np.uint8)
        self.lib.fo_jpeg_encode.restype = C.c_size_t
        n = self.lib.fo_jpeg_encode(C.byref(im), int(quality), buf.ctypes.data_as(C.c_void_p), C.c_size_t(cap))
        assert 0 < n <= cap
        return buf[:n].tobytes()

    def jpeg_info(self, data: bytes):
        """(rc, w, h, components, exif orientation); rc 0 ok, -1 malformed, -2 not covered"""
        w, h, c, o = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.lib.fo_jpeg_info.argtypes = [C.c_char_p, C.c_size_t] + [C.POINTER(C.c_uint32)] * 4
        rc = self.lib.fo_jpeg_info(data, len(data), C.byref(w), C.byref(h), C.byref(c), C.byref(o))
        return rc, w.value, h.value, c.value, o.value

    def jpeg_decode(self, data: bytes):
        """Baseline JPEG -> Luma8 / Rgb8 pixels with zune-jpeg's IDCT / upsampling / colour arithmetic (restated)."""
        rc, w, h, c, _ = self.jpeg_info(data)
        assert rc == 0, rc
        out = np.zeros((h, w, c), np.uint8)
        self.lib.fo_jpeg_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        rc = self.lib.fo_jpeg_decode(data, len(data), out.ctypes.data_as(C.c_void_p))
        assert rc == 0, rc
        return out

    def jpeg_adobe_transform(self, data: bytes) -> int:
        self.lib.fo_jpeg_adobe_transform.argtypes = [C.c_char_p, C.c_size_t]
        return int(self.lib.fo_jpeg_adobe_transform(data, len(data)))

    def jpeg_file_coefficients(self, data: bytes):
        """Quantised coefficients of every block of a JPEG file, [block][64] zig-zag (entropy decoding only)."""
        nb = C.c_uint32()
        self.lib.fo_jpeg_coefficients_of.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
        assert self.lib.fo_jpeg_coefficients_of(data, len(data), None, 0, C.byref(nb)) == 0
        out = np.zeros((nb.value, 64), np.int16)
        assert self.lib.fo_jpeg_coefficients_of(data, len(data), out.ctypes.data_as(C.c_void_p), nb.value, C.byref(nb)) == 0
        return out

    def cmyk_to_rgb(self, cmyk, clut):
        a = np.ascontiguousarray(cmyk, dtype=np.uint8)
        t = np.ascontiguousarray(clut, dtype=np.uint16)
        grid = t.shape[0]
        assert t.shape == (grid,) * 4 + (3,) and a.shape[-1] == 4
        out = np.empty(a.shape[:-1] + (3,), np.uint8)
        self.lib.fo_cmyk_to_rgb(a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size // 4), t.ctypes.data_as(C.c_void_p), C.c_uint32(grid),
                                out.ctypes.data_as(C.c_void_p))
        return out

    def apply_orientation(self, image, exif):
        src, keep = self._img(image)
        dst = fo_image()
        assert self.lib.fo_apply_orientation(C.byref(src), int(exif), C.byref(dst)) == 0
        return self._take(dst)

    def grayscale(self, image):
        im, keep = self._img(image)
        out = fo_image()
        assert self.lib.fo_grayscale(C.byref(im), C.byref(out)) == 0
        return self._take(out)

    def invert(self, image):
        a = np.ascontiguousarray(image, dtype=np.uint8).copy()
        if a.ndim == 2:
            a = a[:, :, None]
        im = fo_image(a.shape[1], a.shape[0], a.shape[2], a.ctypes.data_as(C.POINTER(C.c_uint8)))
        self.lib.fo_invert(C.byref(im))
        return a

    def letterbox(self, image, w, h, fill):
        im, keep = self._img(image)
        out = fo_image()
        f = (C.c_uint8 * 3)(*fill)
        assert self.lib.fo_letterbox(C.byref(im), w, h, f, C.byref(out)) == 0
        return self._take(out)

    def jpeg_ycbcr444(self, image):
        im, keep = self._img(image)
        pw, ph = (im.w + 7) & ~7, (im.h + 7) & ~7
        buf = np.empty(3 * pw * ph, np.uint8)
        ow, oh = C.c_uint32(), C.c_uint32()
        assert self.lib.fo_jpeg_ycbcr444(C.byref(im), buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(ow), C.byref(oh)) == 0
        n = pw * ph
        return buf[:n].reshape(ph, pw), buf[n:2 * n].reshape(ph, pw), buf[2 * n:].reshape(ph, pw)

    def webp_yuv420(self, image):
        im, keep = self._img(image)
        assert im.c == 4
        cw, ch = (im.w + 1) // 2, (im.h + 1) // 2
        buf = np.zeros(im.w * im.h * 2 + 2 * cw * ch, np.uint8)
        has_alpha = self.lib.fo_webp_yuv420(C.byref(im), buf.ctypes.data_as(C.POINTER(C.c_uint8)))
        assert has_alpha >= 0
        ny, nc = im.w * im.h, cw * ch
        return (buf[:ny].reshape(im.h, im.w), buf[ny:ny + nc].reshape(ch, cw), buf[ny + nc:ny + 2 * nc].reshape(ch, cw), bool(has_alpha))

    def ycck_to_cmyk(self, raw):
        a = np.ascontiguousarray(raw, dtype=np.uint8).copy()
        self.lib.fo_ycck_to_cmyk(a.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(a.size // 4))
        return a


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ORACLE_DIR, "fanlin_oracle.c")):
        build()
    return Oracle(C.CDLL(LIB))
