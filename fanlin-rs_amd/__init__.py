"""fanlin-rs_amd -- host-side mirror of the fanlin-rs image hot path over the MI355X C ABI.

The directory name follows the reference (``fanlin-rs``) and is not a valid Python
identifier; load it with :func:`importlib` under the name ``fanlin_rs_amd`` (see
``tests/conftest.py`` / ``__graft_entry__.py``).

Everything here is plumbing over ``libfanlin_gpu.so`` (``include/fanlin_gpu.h``):

* :class:`Query`  mirrors ``query::Query`` (reference ``src/query.rs:3-94``),
* :class:`Format` mirrors ``content::Format`` (``src/content.rs:12-48``),
* :class:`State`  mirrors the pixel part of ``handler::State::process_image``
  (``src/handler.rs:185-309``): same parameter meaning, same order of operations,
  errors surface as :class:`FanlinError` exactly where the reference returns ``Err``.

There is no CPU fallback: if the shared library is missing or no HIP device is
present, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLGPU_LIB") or os.path.join(_HERE, "libfanlin_gpu.so")

FE_NONE, FE_JFIF444, FE_WEBP420, FE_JPEG = 0, 1, 2, 3
ACCEPT_WEBP, ACCEPT_AVIF = 1, 2
OUT_KEEP, OUT_WEBP, OUT_AVIF = 0, 1, 2
IN_OTHER, IN_JPEG, IN_PNG, IN_WEBP, IN_GIF_FRAME = 0, 1, 2, 3, 4
RESULT_AS_IS, RESULT_JPEG_STREAM, RESULT_WEBP_PLANES, RESULT_PIXELS = 0, 1, 2, 3
MIME = {IN_JPEG: "image/jpeg", IN_PNG: "image/png", IN_WEBP: "image/webp", IN_GIF_FRAME: "image/gif"}
IMG_FRONTEND_PLANES, IMG_HAS_ALPHA, IMG_ENCODED, IMG_PINNED, IMG_JPEG_SOURCE = 1, 2, 4, 8, 16
BATCH_SAME_PARAMS = 1
(OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_OOM, ERR_DEVICE, ERR_PARSE, ERR_BUFFER_TOO_SMALL,
 ERR_SHUTDOWN) = range(9)


class FanlinError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"flgpu status {status}: {message}")
        self.status = status


class flgpu_image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("capacity", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32),
                ("channels", C.c_uint32), ("flags", C.c_uint32), ("bytes", C.c_uint64)]


class flgpu_query(C.Structure):
    _fields_ = [(n, C.c_uint8) for n in ("has_w", "has_h", "has_rgb", "has_quality", "has_crop", "has_blur",
                                          "has_grayscale", "has_inverse", "has_avif", "has_webp",
                                          "quality", "crop", "blur", "grayscale", "inverse", "avif", "webp", "reserved")] + \
               [("w", C.c_uint32), ("h", C.c_uint32), ("rgb", C.c_char * 112)]


class flgpu_params(C.Structure):
    _fields_ = [("has_dims", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32),
                ("fill_r", C.c_uint8), ("fill_g", C.c_uint8), ("fill_b", C.c_uint8), ("crop", C.c_uint8),
                ("blur_sigma", C.c_float),
                ("grayscale", C.c_uint8), ("inverse", C.c_uint8), ("quality", C.c_uint8), ("front_end", C.c_uint8),
                ("orientation", C.c_uint8), ("filter", C.c_uint8), ("reserved", C.c_uint8 * 2)]


class flgpu_plan(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("src_w", "src_h", "mid_c", "resampled", "resized_w", "resized_h", "crop_x", "crop_y",
                                           "letterboxed", "place_x", "place_y", "out_w", "out_h", "out_c",
                                           "plane_w", "plane_h", "chroma_w", "chroma_h")] + \
               [("pixel_bytes", C.c_uint64), ("out_bytes", C.c_uint64), ("max_out_bytes", C.c_uint64)]


MAX_DEVICES = 8


class flgpu_config(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_batch", C.c_uint32), ("flush_timeout_us", C.c_uint32),
                ("profile", C.c_uint32), ("queue_lanes", C.c_uint32), ("n_devices", C.c_uint32), ("use_embedded_profile", C.c_uint32),
                ("decode_threads", C.c_uint32),
                ("devices", C.c_int32 * MAX_DEVICES)]


class flgpu_stats(C.Structure):
    _fields_ = [("images", C.c_uint64), ("batches", C.c_uint64), ("queue_flushes", C.c_uint64),
                ("tables_built", C.c_uint64), ("resample_launches", C.c_uint64), ("resample_ms", C.c_double),
                ("resample_src_bytes", C.c_uint64), ("resample_dst_bytes", C.c_uint64),
                ("generic_launches", C.c_uint64), ("blur_launches", C.c_uint64), ("blur_ms", C.c_double),
                ("frontend_launches", C.c_uint64), ("frontend_ms", C.c_double),
                ("cmyk_pixels", C.c_uint64), ("cmyk_tables_baked", C.c_uint64),
                ("jpeg_sources", C.c_uint64), ("jpeg_file_bytes", C.c_uint64), ("jpeg_upload_bytes", C.c_uint64),
                ("mfma_launches", C.c_uint64), ("jpeg_device_huffman", C.c_uint64), ("jpeg_device_huffman_retries", C.c_uint64),
                ("wtile_launches", C.c_uint64)]


class flgpu_jpeg_info(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("width", "height", "components", "channels", "progressive", "restart_interval",
                                           "h_max", "v_max", "exif_orientation", "supported", "adobe_transform", "has_icc_profile")]


# every symbol include/fanlin_gpu.h declares
EXPORTED_SYMBOLS = (
    "flgpu_query_parse", "flgpu_query_dimensions", "flgpu_query_fill_color", "flgpu_query_quality",
    "flgpu_query_cropping", "flgpu_query_blur", "flgpu_query_grayscale", "flgpu_query_inverse",
    "flgpu_query_use_avif", "flgpu_query_use_webp", "flgpu_query_as_is", "flgpu_query_unsupported_scale_size",
    "flgpu_params_from_query", "flgpu_plan_output", "flgpu_process_image", "flgpu_process_image_plan", "flgpu_create", "flgpu_destroy", "flgpu_transform",
    "flgpu_transform_batch", "flgpu_transform_batch_device", "flgpu_batch_results", "flgpu_plan_shards", "flgpu_devices",
    "flgpu_cmyk_distribution", "flgpu_rccl_selftest", "flgpu_jpeg_info_of", "flgpu_decode_jpeg", "flgpu_process_jpeg", "flgpu_process_jpeg_plan", "flgpu_host_alloc", "flgpu_host_free", "flgpu_ycck_to_cmyk",
    "flgpu_set_cmyk_profile", "flgpu_cmyk_bake_available", "flgpu_set_cmyk_clut", "flgpu_get_cmyk_clut",
    "flgpu_cmyk_to_rgb", "flgpu_cmyk_to_rgb_device", "flgpu_export_tables", "flgpu_copy_tables",
    "flgpu_import_tables", "flgpu_get_stats",
    "flgpu_reset_stats", "flgpu_debug_set", "flgpu_debug_get", "flgpu_strerror", "flgpu_last_error", "flgpu_abi_version", "flgpu_build_info",
    "flgpu_debug_axis_table", "flgpu_debug_stream_schedulable", "flgpu_debug_jpeg_blob", "flgpu_debug_mfma_plan", "flgpu_debug_mfma_plan_arith", "flgpu_debug_assign_items",
)

_lib = None


def load_library() -> C.CDLL:
    """Loads libfanlin_gpu.so (built by ``__graft_entry__.build()``); raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(there is no CPU fallback)")
    # PyTorch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so).  Two HIP runtimes in one
    # process cannot both own the device, so when torch is installed it must be loaded first: the
    # library's libamdhip64.so.7 dependency then binds to torch's copy and both share one runtime.
    # Without torch (e.g. under the Rust host) the system ROCm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.flgpu_query_parse.argtypes = [C.c_char_p, C.POINTER(flgpu_query)]
    lib.flgpu_query_parse.restype = C.c_int
    lib.flgpu_query_dimensions.argtypes = [C.POINTER(flgpu_query), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.flgpu_query_fill_color.argtypes = [C.POINTER(flgpu_query)] + [C.POINTER(C.c_uint8)] * 3
    lib.flgpu_query_fill_color.restype = None
    lib.flgpu_query_quality.argtypes = [C.POINTER(flgpu_query)]
    lib.flgpu_query_quality.restype = C.c_uint8
    lib.flgpu_query_blur.argtypes = [C.POINTER(flgpu_query)]
    lib.flgpu_query_blur.restype = C.c_float
    for name in ("cropping", "grayscale", "inverse", "use_avif", "use_webp", "as_is", "unsupported_scale_size"):
        f = getattr(lib, "flgpu_query_" + name)
        f.argtypes = [C.POINTER(flgpu_query)]
        f.restype = C.c_int
    lib.flgpu_params_from_query.argtypes = [C.POINTER(flgpu_query), C.c_uint32, C.c_int, C.POINTER(flgpu_params), C.POINTER(C.c_int)]
    lib.flgpu_process_image.argtypes = [C.c_void_p, C.POINTER(flgpu_image), C.c_uint8, C.c_char_p, C.c_uint32, C.c_int, C.POINTER(flgpu_image),
                                        C.POINTER(flgpu_plan), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.flgpu_process_image_plan.argtypes = [C.POINTER(flgpu_image), C.c_uint8, C.c_char_p, C.c_uint32, C.c_int, C.POINTER(flgpu_plan), C.POINTER(C.c_int)]
    lib.flgpu_plan_output.argtypes = [C.POINTER(flgpu_params), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(flgpu_plan)]
    lib.flgpu_create.argtypes = [C.POINTER(flgpu_config), C.POINTER(C.c_int)]
    lib.flgpu_create.restype = C.c_void_p
    lib.flgpu_destroy.argtypes = [C.c_void_p]
    lib.flgpu_destroy.restype = None
    lib.flgpu_transform.argtypes = [C.c_void_p, C.POINTER(flgpu_image), C.POINTER(flgpu_params), C.POINTER(flgpu_image)]
    lib.flgpu_transform_batch.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(flgpu_image), C.POINTER(flgpu_params), C.POINTER(flgpu_image)]
    lib.flgpu_transform_batch_device.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(flgpu_image), C.POINTER(flgpu_params),
                                                 C.POINTER(flgpu_image), C.c_void_p, C.c_uint32]
    lib.flgpu_host_alloc.argtypes = [C.c_void_p, C.c_uint64]
    lib.flgpu_host_alloc.restype = C.c_void_p
    lib.flgpu_host_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.flgpu_host_free.restype = None
    lib.flgpu_batch_results.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(flgpu_image)]
    lib.flgpu_plan_shards.argtypes = [C.c_uint32, C.c_size_t, C.POINTER(flgpu_image), C.POINTER(flgpu_params), C.c_uint32,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    lib.flgpu_jpeg_info_of.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(flgpu_jpeg_info)]
    lib.flgpu_decode_jpeg.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(flgpu_image)]
    lib.flgpu_process_jpeg.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint32, C.POINTER(flgpu_image), C.POINTER(flgpu_plan),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.flgpu_process_jpeg_plan.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint32, C.POINTER(flgpu_plan), C.POINTER(C.c_int)]
    lib.flgpu_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_uint32]
    lib.flgpu_devices.restype = C.c_uint32
    lib.flgpu_cmyk_distribution.argtypes = [C.c_void_p]
    lib.flgpu_rccl_selftest.argtypes = [C.c_int, C.POINTER(C.c_uint32)]
    lib.flgpu_rccl_selftest.restype = C.c_int
    lib.flgpu_ycck_to_cmyk.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.flgpu_set_cmyk_profile.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    lib.flgpu_set_cmyk_clut.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.flgpu_get_cmyk_clut.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32)]
    lib.flgpu_cmyk_to_rgb.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32]
    lib.flgpu_cmyk_to_rgb_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    lib.flgpu_export_tables.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.flgpu_copy_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.flgpu_import_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    lib.flgpu_get_stats.argtypes = [C.c_void_p, C.POINTER(flgpu_stats)]
    lib.flgpu_reset_stats.argtypes = [C.c_void_p]
    if hasattr(lib, "flgpu_debug_set"):   # (libraries of rounds 1-4, loaded through FLGPU_LIB by the A/B tools, have no such entry point)
        lib.flgpu_debug_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.flgpu_debug_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
    lib.flgpu_strerror.argtypes = [C.c_int]
    lib.flgpu_strerror.restype = C.c_char_p
    lib.flgpu_last_error.argtypes = [C.c_void_p]
    lib.flgpu_last_error.restype = C.c_char_p
    lib.flgpu_abi_version.restype = C.c_uint32
    lib.flgpu_build_info.restype = C.c_char_p
    lib.flgpu_debug_axis_table.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_uint64)]
    lib.flgpu_debug_stream_schedulable.argtypes = [C.c_uint32] * 4 + [C.POINTER(C.c_uint32)]
    lib.flgpu_debug_mfma_plan.argtypes = [C.c_uint32] * 9 + [C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    lib.flgpu_debug_mfma_plan_arith.argtypes = [C.c_uint32] * 10 + [C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    _lib = lib
    return lib


def build_info() -> str:
    """flgpu_build_info(): which sources (hash), compiler and flags the loaded library was built from."""
    return load_library().flgpu_build_info().decode()


def source_hash() -> str:
    """The hash csrc/Makefile computes over the library's sources as they are on disk now (same file order)."""
    import hashlib
    import re
    csrc = os.path.join(_HERE, "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = re.search(r"^SRCS\s*=\s*(.*)$", mk, re.M).group(1).split()
    hdrs = re.search(r"^HDRS\s*=\s*(.*)$", mk, re.M).group(1).split()
    h = hashlib.sha256()
    for f in srcs + hdrs + ["Makefile"]:
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def _check(status: int, ctx: Optional[int] = None) -> None:
    if status == 0:
        return
    lib = load_library()
    msg = lib.flgpu_strerror(status).decode()
    if ctx:
        detail = lib.flgpu_last_error(ctx).decode()
        if detail:
            msg += f" ({detail})"
    raise FanlinError(status, msg)


class Format:
    """content::Format (reference src/content.rs:12-48): bit 0 webp, bit 1 avif."""

    def __init__(self, flags: int = 0):
        self.flags = flags

    def accept_webp(self) -> None:
        self.flags |= ACCEPT_WEBP

    def webp_accepted(self) -> bool:
        return (self.flags & ACCEPT_WEBP) == ACCEPT_WEBP

    def accept_avif(self) -> None:
        self.flags |= ACCEPT_AVIF

    def avif_accepted(self) -> bool:
        return (self.flags & ACCEPT_AVIF) == ACCEPT_AVIF

    @classmethod
    def from_accept_header(cls, value: str) -> "Format":
        """extract_accepted_image_formats (reference src/main.rs:255-274): exact MIME items, comma separated."""
        f = cls()
        for item in value.split(","):
            if item == "image/webp":
                f.accept_webp()
            elif item == "image/avif":
                f.accept_avif()
        return f


class Query:
    """query::Query (reference src/query.rs:3-94), parsed with axum's Query extractor semantics."""

    def __init__(self, raw: flgpu_query):
        self._q = raw

    @classmethod
    def parse(cls, query_string: str) -> "Query":
        q = flgpu_query()
        _check(load_library().flgpu_query_parse(query_string.encode(), C.byref(q)))
        return cls(q)

    def fields(self) -> dict:
        """Option fields as a dict (None = absent), for comparing with the reference's `want` structs."""
        q = self._q
        return {
            "w": q.w if q.has_w else None, "h": q.h if q.has_h else None,
            "rgb": q.rgb.decode() if q.has_rgb else None,
            "quality": q.quality if q.has_quality else None, "crop": bool(q.crop) if q.has_crop else None,
            "blur": q.blur if q.has_blur else None, "grayscale": bool(q.grayscale) if q.has_grayscale else None,
            "inverse": bool(q.inverse) if q.has_inverse else None, "avif": bool(q.avif) if q.has_avif else None,
            "webp": bool(q.webp) if q.has_webp else None,
        }

    def dimensions(self) -> Optional[Tuple[int, int]]:
        w, h = C.c_uint32(), C.c_uint32()
        return (w.value, h.value) if load_library().flgpu_query_dimensions(C.byref(self._q), C.byref(w), C.byref(h)) else None

    def fill_color(self) -> Tuple[int, int, int]:
        r, g, b = C.c_uint8(), C.c_uint8(), C.c_uint8()
        load_library().flgpu_query_fill_color(C.byref(self._q), C.byref(r), C.byref(g), C.byref(b))
        return (r.value, g.value, b.value)

    def quality(self) -> int:
        return int(load_library().flgpu_query_quality(C.byref(self._q)))

    def cropping(self) -> bool:
        return bool(load_library().flgpu_query_cropping(C.byref(self._q)))

    def blur(self) -> float:
        return float(load_library().flgpu_query_blur(C.byref(self._q)))

    def grayscale(self) -> bool:
        return bool(load_library().flgpu_query_grayscale(C.byref(self._q)))

    def inverse(self) -> bool:
        return bool(load_library().flgpu_query_inverse(C.byref(self._q)))

    def use_avif(self) -> bool:
        return bool(load_library().flgpu_query_use_avif(C.byref(self._q)))

    def use_webp(self) -> bool:
        return bool(load_library().flgpu_query_use_webp(C.byref(self._q)))

    def as_is(self) -> bool:
        return bool(load_library().flgpu_query_as_is(C.byref(self._q)))

    def unsupported_scale_size(self) -> bool:
        return bool(load_library().flgpu_query_unsupported_scale_size(C.byref(self._q)))

    def to_params(self, content: Format = None, input_is_jpeg: bool = False) -> Tuple[flgpu_params, int]:
        p, fmt = flgpu_params(), C.c_int()
        _check(load_library().flgpu_params_from_query(C.byref(self._q), content.flags if content else 0,
                                                      int(input_is_jpeg), C.byref(p), C.byref(fmt)))
        return p, fmt.value


FILTER_LANCZOS3, FILTER_NEAREST = 0, 1
CMYK_GRID, CMYK_INPUT_YCCK = 17, 1


def make_params(w: Optional[int] = None, h: Optional[int] = None, fill=(32, 32, 32), crop=False, blur_sigma=0.0,
                grayscale=False, inverse=False, quality=75, front_end=FE_NONE, orientation=1, filter=FILTER_LANCZOS3) -> flgpu_params:
    p = flgpu_params()
    p.has_dims = 1 if (w is not None and h is not None) else 0
    p.w, p.h = (w or 0), (h or 0)
    p.fill_r, p.fill_g, p.fill_b = fill
    p.crop = int(crop)
    p.blur_sigma = blur_sigma
    p.grayscale, p.inverse = int(grayscale), int(inverse)
    p.quality, p.front_end = quality, front_end
    p.orientation = orientation
    p.filter = filter
    return p


def plan_output(params: flgpu_params, sw: int, sh: int, sc: int) -> flgpu_plan:
    plan = flgpu_plan()
    _check(load_library().flgpu_plan_output(C.byref(params), sw, sh, sc, C.byref(plan)))
    return plan


def rccl_selftest(device: int = 0) -> dict:
    """flgpu_rccl_selftest: the RCCL path of the CMYK table distribution, exercised with one rank on one device."""
    info = (C.c_uint32 * 4)()
    status = load_library().flgpu_rccl_selftest(device, info)
    return {"status": status, "rccl_version": info[0], "bytes_intact": info[1], "tail_untouched": bool(info[2]), "communicator_destroyed": bool(info[3])}


def jpeg_info(data: bytes) -> dict:
    """flgpu_jpeg_info_of: header fields of a JPEG file (pure host function); raises FanlinError(ERR_PARSE) if it is none."""
    info = flgpu_jpeg_info()
    _check(load_library().flgpu_jpeg_info_of(data, len(data), C.byref(info)))
    return {n: getattr(info, n) for n, _ in flgpu_jpeg_info._fields_}


def debug_jpeg_blob(data: bytes):
    """Host half of the JPEG decode front end: returns (header dict, quantised coefficients [block][64] zig-zag, raw blob)."""
    lib = load_library()
    lib.flgpu_debug_jpeg_blob.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    used = C.c_uint64()
    _check(lib.flgpu_debug_jpeg_blob(data, len(data), None, 0, C.byref(used)))
    blob = np.zeros(used.value, np.uint8)
    _check(lib.flgpu_debug_jpeg_blob(data, len(data), blob.ctypes.data, blob.size, C.byref(used)))
    blob = blob[: used.value]
    u32 = blob[: (blob.size // 4) * 4].view(np.uint32)
    hdr = dict(zip(("magic", "width", "height", "nc", "hmax", "vmax", "is_rgb", "nblocks", "blocks_off", "coef_off", "plane_bytes", "total_bytes"), u32[:12].tolist()))
    words = blob[hdr["blocks_off"]: hdr["blocks_off"] + 4 * hdr["nblocks"]].view(np.uint32)
    data = blob[hdr["coef_off"]:]
    out = np.zeros((hdr["nblocks"], 64), np.int16)
    HEAD = 4  # kJpegWideHead
    for b, w in enumerate(words.tolist()):
        cnt, wide, first = ((w >> 1) & 63) + 1, w & 1, (w >> 7) * 2
        head = cnt if wide else min(cnt, HEAD)
        out[b, :head] = data[first: first + 2 * head].view(np.int16)
        if cnt > head:
            out[b, head:cnt] = data[first + 2 * head: first + 2 * head + cnt - head].view(np.int8)
    return hdr, out, blob


def plan_shards(n_shards: int, shapes: Sequence[Tuple[int, int, int]], params) -> Tuple[np.ndarray, np.ndarray]:
    """flgpu_plan_shards: (shard_of[n], shard_bytes[n_shards]) for images of shapes[i] = (height, width, channels);
    ``params`` is one flgpu_params (shared) or a sequence of n.  Pure host function."""
    n = len(shapes)
    srcs = (flgpu_image * max(n, 1))(*[flgpu_image(None, shapes[i][0] * shapes[i][1] * shapes[i][2], shapes[i][1], shapes[i][0],
                                                   shapes[i][2], 0) for i in range(n)])
    if isinstance(params, flgpu_params):
        ps, flags = (flgpu_params * 1)(params), BATCH_SAME_PARAMS
    else:
        ps, flags = (flgpu_params * max(n, 1))(*params), 0
    shard_of = (C.c_uint32 * max(n, 1))()
    shard_bytes = (C.c_uint64 * n_shards)()
    _check(load_library().flgpu_plan_shards(n_shards, n, srcs, ps, flags, shard_of, shard_bytes))
    return np.array(shard_of[:n], dtype=np.uint32), np.array(shard_bytes[:], dtype=np.uint64)


def debug_axis_table(in_size: int, out_size: int, gaussian: bool = False, sigma: float = 0.0):
    """(left, count, weights) exactly as the runtime uploads them for one axis."""
    lib = load_library()
    left = (C.c_uint32 * out_size)()
    count = (C.c_uint32 * out_size)()
    ratio = max(in_size / out_size, 1.0)
    cap = int(out_size * (2 * (2.0 * sigma if gaussian else 3.0) * ratio + 4)) + 16
    w = (C.c_float * cap)()
    total = C.c_uint64()
    _check(lib.flgpu_debug_axis_table(in_size, out_size, int(gaussian), sigma, left, count, w, cap, C.byref(total)))
    return (np.array(left, dtype=np.uint32), np.array(count, dtype=np.uint32), np.array(w[: total.value], dtype=np.float32))


def debug_mfma_plan(sw: int, sh: int, channels: int, rw: int, rh: int, crop=None, packed: bool = False) -> Optional[dict]:
    """Builds and self-checks the matrix-pipe kernel's tables on the host (csrc/fl_query.cpp flgpu_debug_mfma_plan_arith); None
    if the geometry does not fit the kernel.  crop = (cx, cy, cw, ch) in resized coordinates, default the whole picture.
    packed: the tables of rounds 2-3's arithmetic (FLGPU_MFMA_ARITH=packed) instead of the full-width one."""
    lib = load_library()
    cx, cy, cw, ch = crop if crop else (0, 0, rw, rh)
    info = (C.c_uint32 * 8)()
    err = (C.c_double * 2)()
    if not lib.flgpu_debug_mfma_plan_arith(sw, sh, channels, rw, rh, cx, cy, cw, ch, 0 if packed else 1, info, err):
        return None
    keys = ("tiles", "k_blocks", "strips", "max_operands", "hs", "tail", "bad_horizontal", "bad_vertical")
    d = dict(zip(keys, (int(x) for x in info)))
    d["vertical_weight_error"], d["horizontal_weight_error"] = float(err[0]), float(err[1])
    return d


def debug_assign_items(pictures: int, strips: int, tiles: int = 11, workgroups: int = 256):
    """(job, strip, tile0, tile1, lists): the items of a uniform persistent matrix-pipe launch in launch order and, per workgroup,
    (first item, count) -- csrc/fl_batch.cpp assign_items."""
    lib = load_library()
    cap = pictures * strips * tiles
    arrs = [(C.c_uint32 * cap)() for _ in range(4)]
    lists, n = (C.c_uint32 * (2 * workgroups))(), C.c_uint32()
    lib.flgpu_debug_assign_items.argtypes = [C.c_uint32] * 5 + [C.POINTER(C.c_uint32)] * 6
    _check(lib.flgpu_debug_assign_items(pictures, strips, tiles, workgroups, cap, *arrs, lists, C.byref(n)))
    k = int(n.value)
    g = min(workgroups, pictures * strips)
    return tuple(np.array(a[:k], np.int64) for a in arrs) + (np.array(lists[:2 * g], np.int64).reshape(g, 2),)


def debug_stream_schedulable(in_size: int, out_size: int, y0: int = 0, y1: Optional[int] = None) -> Tuple[bool, int]:
    peak = C.c_uint32()
    ok = load_library().flgpu_debug_stream_schedulable(in_size, out_size, y0, out_size if y1 is None else y1, C.byref(peak))
    return bool(ok), peak.value


@dataclass
class Planes:
    """Encoder front-end output: luma plane and two chroma planes (u8)."""
    y: np.ndarray
    u: np.ndarray
    v: np.ndarray
    has_alpha: bool = False
    a: Optional[np.ndarray] = None   # WebP: the alpha plane (all 255 unless has_alpha)


def _as_image_array(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3 or not (1 <= a.shape[2] <= 4):
        raise ValueError("image must be HxW or HxWxC with C in 1..4, dtype uint8")
    return a


def _split_output(buf: np.ndarray, plan: flgpu_plan, front_end: int, flags: int, nbytes: int = 0):
    if front_end == FE_NONE:
        return buf[: plan.pixel_bytes].reshape(plan.out_h, plan.out_w, plan.out_c)
    if front_end == FE_JPEG:
        return buf[:nbytes].tobytes()
    ny = plan.plane_w * plan.plane_h
    nc = plan.chroma_w * plan.chroma_h
    return Planes(y=buf[:ny].reshape(plan.plane_h, plan.plane_w),
                  u=buf[ny:ny + nc].reshape(plan.chroma_h, plan.chroma_w),
                  v=buf[ny + nc:ny + 2 * nc].reshape(plan.chroma_h, plan.chroma_w),
                  has_alpha=bool(flags & IMG_HAS_ALPHA),
                  a=buf[ny + 2 * nc:2 * ny + 2 * nc].reshape(plan.plane_h, plan.plane_w) if front_end == FE_WEBP420 else None)


class State:
    """Device context + the pixel part of handler::State::process_image.

    ``process_pixels`` is the single-request entry (goes through the request-batching
    queue, like one tokio worker calling ``process_image``); ``process_batch`` runs
    many host images in one set of launches; ``process_batch_device`` takes device
    pointers (PyTorch tensors) for HBM-resident batches.
    """

    def __init__(self, device: int = -1, max_batch: int = 0, flush_timeout_us: int = 0, profile: bool = False,
                 queue_lanes: int = 0, devices: Optional[Sequence[int]] = None, use_embedded_profile: bool = False,
                 decode_threads: int = 0):
        """``devices``: two or more HIP ordinals make ONE context that shards every batch across those GPUs (an ordinal
        may repeat: two shards on one GPU); None / one entry = a single-device context."""
        lib = load_library()
        cfg = flgpu_config()
        cfg.device, cfg.max_batch, cfg.flush_timeout_us, cfg.profile = device, max_batch, flush_timeout_us, int(profile)
        cfg.queue_lanes = queue_lanes
        cfg.use_embedded_profile = int(use_embedded_profile)
        cfg.decode_threads = decode_threads  # callers that may run the host half of the JPEG decoder at once (0 = the CPUs the process may use)
        if devices:
            if len(devices) > MAX_DEVICES:
                raise ValueError("at most 8 devices")
            cfg.n_devices = len(devices)
            for k, d in enumerate(devices):
                cfg.devices[k] = d
        st = C.c_int()
        self._ctx = lib.flgpu_create(C.byref(cfg), C.byref(st))
        if not self._ctx:
            _check(st.value or 3)
        self._lib = lib

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            self._lib.flgpu_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- host memory ---------------------------------------------------------
    def process_pixels(self, image: np.ndarray, params: flgpu_params, capacity: int = 0):
        img = _as_image_array(image)
        plan = plan_output(params, img.shape[1], img.shape[0], img.shape[2])
        # capacity < 0: exactly -capacity bytes (tests of the too-small path); otherwise the format's worst case, so that the
        # call fails as rarely as JpegEncoder::encode_image into a Vec does (never)
        out = np.empty(-capacity if capacity < 0 else max(int(plan.max_out_bytes), capacity, 1), dtype=np.uint8)
        src = flgpu_image(img.ctypes.data, img.nbytes, img.shape[1], img.shape[0], img.shape[2], 0)
        dst = flgpu_image(out.ctypes.data, out.nbytes, 0, 0, 0, 0)
        _check(self._lib.flgpu_transform(self._ctx, C.byref(src), C.byref(params), C.byref(dst)), self._ctx)
        return _split_output(out, plan, params.front_end, dst.flags, dst.bytes)

    def process_image(self, decoded: np.ndarray, query_string: str, content: "Format" = None, input_format: int = IN_JPEG,
                      orientation: int = 1):
        """State::process_image after the decoder (reference src/handler.rs:198-308): returns (mime, kind, payload) where
        payload is None (AS_IS), the JPEG body (bytes), WebP planes (Planes) or the pixels for a host encoder (ndarray)."""
        img = _as_image_array(decoded)
        src = flgpu_image(img.ctypes.data, img.nbytes, img.shape[1], img.shape[0], img.shape[2], 0)
        plan, kind, fmt = flgpu_plan(), C.c_int(), C.c_int()
        flags = content.flags if content else 0
        qs = query_string.encode()
        _check(self._lib.flgpu_process_image_plan(C.byref(src), orientation, qs, flags, input_format, C.byref(plan), C.byref(kind)))
        if kind.value == RESULT_AS_IS:
            return MIME.get(input_format, "application/octet-stream"), RESULT_AS_IS, None
        out = np.empty(max(int(plan.max_out_bytes), 1), dtype=np.uint8)
        dst = flgpu_image(out.ctypes.data, out.nbytes, 0, 0, 0, 0)
        _check(self._lib.flgpu_process_image(self._ctx, C.byref(src), orientation, qs, flags, input_format, C.byref(dst), C.byref(plan),
                                             C.byref(kind), C.byref(fmt)), self._ctx)
        mime = "image/webp" if fmt.value == OUT_WEBP else "image/avif" if fmt.value == OUT_AVIF else MIME.get(input_format, "application/octet-stream")
        fe = {RESULT_JPEG_STREAM: FE_JPEG, RESULT_WEBP_PLANES: FE_WEBP420, RESULT_PIXELS: FE_NONE}[kind.value]
        return mime, kind.value, _split_output(out, plan, fe, dst.flags, dst.bytes)

    # -- JPEG sources: the decode front end (handler.rs:205-220) --------------------------------------------------
    def decode_jpeg(self, data: bytes) -> np.ndarray:
        """DynamicImage::from_decoder(JpegDecoder::new(..)) on the device: (h, w, 1 or 3) uint8."""
        info = jpeg_info(data)
        out = np.empty((info["height"], info["width"], max(info["channels"], 1)), np.uint8)
        dst = flgpu_image(out.ctypes.data, out.nbytes, 0, 0, 0, 0)
        _check(self._lib.flgpu_decode_jpeg(self._ctx, data, len(data), C.byref(dst)), self._ctx)
        return out

    def process_jpeg_pixels(self, data: bytes, params: flgpu_params):
        """process_pixels with a JPEG FILE as the source (FLGPU_IMG_JPEG_SOURCE): decode + pipeline in one device pass."""
        info = jpeg_info(data)
        plan = plan_output(params, info["width"], info["height"], max(info["channels"], 1))
        out = np.empty(max(int(plan.max_out_bytes), 1), dtype=np.uint8)
        buf = C.create_string_buffer(data, len(data))
        src = flgpu_image(C.cast(buf, C.c_void_p).value, len(data), info["width"], info["height"], max(info["channels"], 1), IMG_JPEG_SOURCE)
        dst = flgpu_image(out.ctypes.data, out.nbytes, 0, 0, 0, 0)
        _check(self._lib.flgpu_transform(self._ctx, C.byref(src), C.byref(params), C.byref(dst)), self._ctx)
        return _split_output(out, plan, params.front_end, dst.flags, dst.bytes)

    def process_jpeg(self, data: bytes, query_string: str, content: "Format" = None):
        """State::process_image for a JPEG input from the file bytes on: (mime, kind, payload) as process_image."""
        plan, kind, fmt = flgpu_plan(), C.c_int(), C.c_int()
        flags = content.flags if content else 0
        qs = query_string.encode()
        _check(self._lib.flgpu_process_jpeg_plan(data, len(data), qs, flags, C.byref(plan), C.byref(kind)))
        if kind.value == RESULT_AS_IS:
            return "image/jpeg", RESULT_AS_IS, None
        out = np.empty(max(int(plan.max_out_bytes), 1), dtype=np.uint8)
        dst = flgpu_image(out.ctypes.data, out.nbytes, 0, 0, 0, 0)
        _check(self._lib.flgpu_process_jpeg(self._ctx, data, len(data), qs, flags, C.byref(dst), C.byref(plan), C.byref(kind), C.byref(fmt)), self._ctx)
        mime = "image/webp" if fmt.value == OUT_WEBP else "image/avif" if fmt.value == OUT_AVIF else "image/jpeg"
        fe = {RESULT_JPEG_STREAM: FE_JPEG, RESULT_WEBP_PLANES: FE_WEBP420, RESULT_PIXELS: FE_NONE}[kind.value]
        return mime, kind.value, _split_output(out, plan, fe, dst.flags, dst.bytes)

    def process_batch(self, images: Sequence, params: Sequence[flgpu_params]) -> List:
        """images[i]: an ndarray of pixels, or `bytes` holding a JPEG file (decoded by the library)."""
        n = len(images)
        infos = [jpeg_info(a) if isinstance(a, (bytes, bytearray)) else None for a in images]
        keep = [C.create_string_buffer(bytes(a), len(a)) if infos[i] else None for i, a in enumerate(images)]
        imgs = [None if infos[i] else _as_image_array(a) for i, a in enumerate(images)]
        shapes = [(infos[i]["height"], infos[i]["width"], max(infos[i]["channels"], 1)) if infos[i] else imgs[i].shape for i in range(n)]
        plans = [plan_output(params[i], shapes[i][1], shapes[i][0], shapes[i][2]) for i in range(n)]
        outs = [np.empty(max(int(pl.max_out_bytes), 1), dtype=np.uint8) for pl in plans]
        srcs = (flgpu_image * n)(*[flgpu_image(C.cast(keep[i], C.c_void_p).value, len(images[i]), shapes[i][1], shapes[i][0], shapes[i][2], IMG_JPEG_SOURCE)
                                   if infos[i] else flgpu_image(imgs[i].ctypes.data, imgs[i].nbytes, shapes[i][1], shapes[i][0], shapes[i][2], 0) for i in range(n)])
        dsts = (flgpu_image * n)(*[flgpu_image(o.ctypes.data, o.nbytes, 0, 0, 0, 0) for o in outs])
        ps = (flgpu_params * n)(*params)
        _check(self._lib.flgpu_transform_batch(self._ctx, n, srcs, ps, dsts), self._ctx)
        return [_split_output(outs[i], plans[i], params[i].front_end, dsts[i].flags, dsts[i].bytes) for i in range(n)]

    # -- device memory -------------------------------------------------------
    def process_batch_device(self, src_ptrs: Sequence[int], shapes: Sequence[Tuple[int, int, int]],
                             params, dst_ptrs: Sequence[int], dst_caps: Sequence[int], stream: int = 0) -> None:
        """Enqueues n HBM-resident images. shapes[i] = (height, width, channels). ``params`` is one
        flgpu_params (shared) or a sequence of n.  Asynchronous on ``stream`` (a hipStream_t value)."""
        n = len(src_ptrs)
        srcs = (flgpu_image * n)(*[flgpu_image(src_ptrs[i], shapes[i][0] * shapes[i][1] * shapes[i][2], shapes[i][1],
                                               shapes[i][0], shapes[i][2], 0) for i in range(n)])
        dsts = (flgpu_image * n)(*[flgpu_image(dst_ptrs[i], dst_caps[i], 0, 0, 0, 0) for i in range(n)])
        if isinstance(params, flgpu_params):
            ps, flags = (flgpu_params * 1)(params), BATCH_SAME_PARAMS
        else:
            ps, flags = (flgpu_params * n)(*params), 0
        self._keep = (srcs, dsts, ps)
        _check(self._lib.flgpu_transform_batch_device(self._ctx, n, srcs, ps, dsts, C.c_void_p(stream), flags), self._ctx)

    def batch_results(self) -> List[Tuple[int, int]]:
        """(flags, bytes) of every image of the last ``process_batch_device`` call; waits for it."""
        srcs, dsts, ps = self._keep
        _check(self._lib.flgpu_batch_results(self._ctx, len(dsts), dsts), self._ctx)
        return [(d.flags, d.bytes) for d in dsts]

    def prepared_batch(self, src_ptrs, shapes, params, dst_ptrs, dst_caps):
        """Pre-marshals a device batch so that the timed loop only pays for the C call."""
        n = len(src_ptrs)
        srcs = (flgpu_image * n)(*[flgpu_image(src_ptrs[i], shapes[i][0] * shapes[i][1] * shapes[i][2], shapes[i][1],
                                               shapes[i][0], shapes[i][2], 0) for i in range(n)])
        dsts = (flgpu_image * n)(*[flgpu_image(dst_ptrs[i], dst_caps[i], 0, 0, 0, 0) for i in range(n)])
        if isinstance(params, flgpu_params):
            ps, flags = (flgpu_params * 1)(params), BATCH_SAME_PARAMS
        else:
            ps, flags = (flgpu_params * n)(*params), 0
        ctx, lib = self._ctx, self._lib

        def run(stream: int = 0):
            _check(lib.flgpu_transform_batch_device(ctx, n, srcs, ps, dsts, C.c_void_p(stream), flags), ctx)
        run._keep = (srcs, dsts, ps)
        self._keep = run._keep   # (so that batch_results() can collect the last run of a prepared batch too)
        return run

    def ycck_to_cmyk(self, raw: np.ndarray) -> np.ndarray:
        """handler.rs:423-438 on an (..., 4) uint8 array of (Y, Cb, Cr, K) pixels; returns a converted copy."""
        a = np.ascontiguousarray(raw, dtype=np.uint8).copy()
        if a.size % 4:
            raise ValueError("expected 4 bytes per pixel")
        _check(self._lib.flgpu_ycck_to_cmyk(self._ctx, a.ctypes.data, a.size // 4), self._ctx)
        return a

    # ---- CMYK / YCCK JPEG sources (handler.rs:398-493) ----
    def set_cmyk_profile(self, icc: bytes) -> None:
        """create_cmyk_to_rgb_converter (main.rs:74-76): bakes the profile's CMYK_8 -> sRGB device-link table."""
        _check(self._lib.flgpu_set_cmyk_profile(self._ctx, icc, len(icc)), self._ctx)

    def set_cmyk_clut(self, rgb_nodes: np.ndarray) -> None:
        a = np.ascontiguousarray(rgb_nodes, dtype=np.uint16)
        if a.size != CMYK_GRID ** 4 * 3:
            raise ValueError("expected a 17^4 x 3 table")
        _check(self._lib.flgpu_set_cmyk_clut(self._ctx, CMYK_GRID, a.ctypes.data), self._ctx)

    def get_cmyk_clut(self) -> np.ndarray:
        a = np.zeros((CMYK_GRID,) * 4 + (3,), np.uint16)
        grid = C.c_uint32()
        _check(self._lib.flgpu_get_cmyk_clut(self._ctx, a.ctypes.data, a.size, C.byref(grid)), self._ctx)
        return a

    def cmyk_to_rgb(self, cmyk: np.ndarray, embedded_icc: Optional[bytes] = None, ycck: bool = False) -> np.ndarray:
        """CMYK2RGB::convert (handler.rs:490-492) on an (..., 4) uint8 array; returns (..., 3)."""
        a = np.ascontiguousarray(cmyk, dtype=np.uint8)
        if a.ndim < 1 or a.shape[-1] != 4:
            raise ValueError("expected 4 bytes per pixel")
        out = np.empty(a.shape[:-1] + (3,), np.uint8)
        n = a.size // 4
        _check(self._lib.flgpu_cmyk_to_rgb(self._ctx, a.ctypes.data, n, out.ctypes.data, embedded_icc,
                                           len(embedded_icc) if embedded_icc else 0, CMYK_INPUT_YCCK if ycck else 0), self._ctx)
        return out

    def cmyk_to_rgb_device(self, d_cmyk: int, d_rgb: int, n_pixels: int, ycck: bool = False, stream: int = 0) -> None:
        _check(self._lib.flgpu_cmyk_to_rgb_device(self._ctx, C.c_void_p(d_cmyk), C.c_void_p(d_rgb), n_pixels,
                                                  CMYK_INPUT_YCCK if ycck else 0, C.c_void_p(stream)), self._ctx)

    def cmyk_distribution(self) -> int:
        """How the configured CMYK table reached the devices of a multi-device context: 2 = RCCL broadcast, 1 = copies, 0 = n/a."""
        return int(self._lib.flgpu_cmyk_distribution(self._ctx))

    def devices(self) -> List[int]:
        buf = (C.c_int32 * MAX_DEVICES)()
        n = self._lib.flgpu_devices(self._ctx, buf, MAX_DEVICES)
        return [int(buf[k]) for k in range(n)]

    def export_tables(self) -> Tuple[int, int]:
        ptr, nbytes = C.c_void_p(), C.c_uint64()
        _check(self._lib.flgpu_export_tables(self._ctx, C.byref(ptr), C.byref(nbytes)), self._ctx)
        return int(ptr.value or 0), int(nbytes.value)

    def copy_tables(self, dst_ptr: int, capacity: int) -> int:
        """Device-to-device copy of the table blob into a caller-owned buffer; returns its size."""
        nbytes = C.c_uint64()
        _check(self._lib.flgpu_copy_tables(self._ctx, C.c_void_p(dst_ptr), capacity, C.byref(nbytes)), self._ctx)
        return int(nbytes.value)

    def import_tables(self, src_ptr: int, nbytes: int) -> None:
        _check(self._lib.flgpu_import_tables(self._ctx, C.c_void_p(src_ptr), nbytes), self._ctx)

    def stats(self) -> dict:
        s = flgpu_stats()
        _check(self._lib.flgpu_get_stats(self._ctx, C.byref(s)), self._ctx)
        return {name: getattr(s, name) for name, _ in flgpu_stats._fields_}

    def reset_stats(self) -> None:
        _check(self._lib.flgpu_reset_stats(self._ctx), self._ctx)

    # -- test / experiment switches (flgpu_debug_set: the library reads the environment only in flgpu_create) ---------------
    def debug_set(self, key: str, value: int = 1) -> None:
        """``key``: "no_mfma", "no_wtile", "mfma_arith" (1 = packed), "force_bands", "host_huffman", ... (include/fanlin_gpu.h);
        "reset" restores every default."""
        _check(self._lib.flgpu_debug_set(self._ctx, key.encode(), int(value)), self._ctx)

    def debug_get(self, key: str) -> int:
        v = C.c_int64()
        _check(self._lib.flgpu_debug_get(self._ctx, key.encode(), C.byref(v)), self._ctx)
        return int(v.value)

    def switches(self, **kw):
        """``with state.switches(no_mfma=1, force_bands=3): ...`` -- sets the switches, restores their old values on exit."""
        state = self

        class _Scope:
            def __enter__(self_inner):
                self_inner.old = {k: state.debug_get(k) for k in kw}
                for k, v in kw.items():
                    state.debug_set(k, v)
                return state

            def __exit__(self_inner, *exc):
                for k, v in self_inner.old.items():
                    state.debug_set(k, v)
                return False

        return _Scope()
