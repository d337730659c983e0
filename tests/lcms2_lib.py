"""ctypes access to the system's Little CMS 2 (liblcms2.so.2) -- the very library the reference's `lcms2` crate
binds (Cargo.lock lcms2 6.1.0 / lcms2-sys).  Test infrastructure: it is the REFERENCE for the CMYK path, used to
pin oracle/ and the HIP kernel, never by the product's pixel path."""
import ctypes as C
import ctypes.util

import numpy as np

TYPE_CMYK_8, TYPE_RGB_8, TYPE_CMYK_16, TYPE_RGB_16 = 0x60021, 0x40019, 0x60022, 0x4001A
INTENT_PERCEPTUAL = 0
FLAGS_NOCACHE, FLAGS_NOOPTIMIZE = 0x0040, 0x0100
GRID = 17

_lib = None


def load():
    global _lib
    if _lib is None:
        for name in ("liblcms2.so.2", ctypes.util.find_library("lcms2")):
            if not name:
                continue
            try:
                _lib = C.CDLL(name)
                break
            except OSError:
                continue
        if _lib is None:
            return None
        _lib.cmsOpenProfileFromMem.restype = C.c_void_p
        _lib.cmsOpenProfileFromMem.argtypes = [C.c_char_p, C.c_uint32]
        _lib.cmsCreate_sRGBProfile.restype = C.c_void_p
        _lib.cmsCreateTransform.restype = C.c_void_p
        _lib.cmsCreateTransform.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib.cmsDoTransform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        _lib.cmsDeleteTransform.argtypes = [C.c_void_p]
        _lib.cmsCloseProfile.argtypes = [C.c_void_p]
        _lib.cmsGetEncodedCMMversion.restype = C.c_int
    return _lib


def version():
    lib = load()
    return lib.cmsGetEncodedCMMversion() if lib else 0


class Cmyk2Rgb:
    """CMYK2RGB::with_icc_profile + convert (reference src/handler.rs:469-493), same formats, intent and flags."""

    def __init__(self, icc: bytes):
        lib = load()
        assert lib is not None, "liblcms2 not available"
        self.lib = lib
        self.src = lib.cmsOpenProfileFromMem(icc, len(icc))
        if not self.src:
            raise ValueError("cmsOpenProfileFromMem failed")
        self.dst = lib.cmsCreate_sRGBProfile()
        self.t8 = lib.cmsCreateTransform(self.src, TYPE_CMYK_8, self.dst, TYPE_RGB_8, INTENT_PERCEPTUAL, FLAGS_NOCACHE)
        if not self.t8:
            raise ValueError("cmsCreateTransform failed (not a CMYK profile?)")

    def convert(self, cmyk):
        a = np.ascontiguousarray(cmyk, dtype=np.uint8)
        out = np.empty(a.shape[:-1] + (3,), np.uint8)
        self.lib.cmsDoTransform(self.t8, a.ctypes.data, out.ctypes.data, a.size // 4)
        return out

    def device_link_nodes(self):
        """The 17^4 x 3 u16 table cmsopt.c samples for this transform: the un-optimised 16-bit transform at the nodes."""
        t16 = self.lib.cmsCreateTransform(self.src, TYPE_CMYK_16, self.dst, TYPE_RGB_16, INTENT_PERCEPTUAL, FLAGS_NOCACHE | FLAGS_NOOPTIMIZE)
        assert t16
        q = np.array([int(np.floor(i * 65535.0 / (GRID - 1) + 0.5)) for i in range(GRID)], np.uint16)
        grid = np.ascontiguousarray(np.stack(np.meshgrid(q, q, q, q, indexing="ij"), -1).reshape(-1, 4))
        out = np.zeros((grid.shape[0], 3), np.uint16)
        self.lib.cmsDoTransform(t16, grid.ctypes.data, out.ctypes.data, grid.shape[0])
        self.lib.cmsDeleteTransform(t16)
        # cmsopt.c FixWhiteMisalignment: the no-ink node is forced to exact white unless it is wildly off
        for k in range(3):
            d = 0xFFFF - int(out[0, k])
            if d > 0xF000:
                break
            if d != 0:
                out[0] = 0xFFFF
                break
        return out.reshape(GRID, GRID, GRID, GRID, 3)
