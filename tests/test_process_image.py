"""flgpu_process_image = State::process_image after the decoder, in one call.  The table below replays the image rows
of the reference's own test_generic_handler (src/main.rs:346-408: same URLs' queries, same Accept header of
image/webp + image/avif, same expected status / Content-Type); 512x512 lenna.* are stood in for by a synthetic picture
of that size since only status and MIME are asserted there."""
import io

import numpy as np
import pytest

import synth

# (input format, query, want status, want Content-Type, want result kind)
CASES = [
    ("jpeg", "", 200, "image/jpeg", "AS_IS"),                                   # /foo/lenna.jpg
    ("jpeg", "w=300&h=200", 200, "image/jpeg", "JPEG_STREAM"),                  # /foo/lenna.jpg?w=300&h=200
    ("jpeg", "w=300&h=200&avif=true", 200, "image/avif", "PIXELS"),
    ("jpeg", "w=300&h=200&webp=true", 200, "image/webp", "WEBP_PLANES"),
    ("jpeg", "w=9999&h=9999", 400, "text/plain; charset=utf-8", None),          # size gate, src/main.rs:134-138
    ("png", "", 200, "image/png", "AS_IS"),
    ("png", "w=300&h=200&avif=true", 200, "image/avif", "PIXELS"),
    ("gif", "", 200, "image/gif", "AS_IS"),
    ("gif", "w=300&h=200&webp=true", 200, "image/gif", "PIXELS"),               # process_gif ignores the negotiation
]


def _fmt(fl, name):
    return {"jpeg": fl.IN_JPEG, "png": fl.IN_PNG, "gif": fl.IN_GIF_FRAME, "webp": fl.IN_WEBP}[name]


def _accept(fl):
    return fl.Format.from_accept_header("image/webp,image/avif")


@pytest.mark.parametrize("src,query,status,mime,kind", CASES)
def test_request_table_host_side(fl, src, query, status, mime, kind):
    import ctypes as C
    lib = fl.load_library()
    img = fl.flgpu_image(None, 512 * 512 * 3, 512, 512, 4 if src == "gif" else 3, 0)
    plan, k = fl.flgpu_plan(), C.c_int(-1)
    rc = lib.flgpu_process_image_plan(C.byref(img), 1, query.encode(), _accept(fl).flags, _fmt(fl, src), C.byref(plan), C.byref(k))
    if status == 400:
        assert rc == fl.ERR_PARSE
        return
    assert rc == fl.OK
    assert k.value == getattr(fl, "RESULT_" + kind)
    if kind == "AS_IS":
        assert plan.out_bytes == 0
    elif kind == "JPEG_STREAM":
        assert (plan.out_w, plan.out_h, plan.out_c) == (300, 200, 4) and plan.out_bytes >= 623 + 3 * 304 * 200
    elif kind == "WEBP_PLANES":
        assert plan.out_bytes == 2 * 300 * 200 + 2 * 150 * 100
    else:
        assert plan.out_bytes == 300 * 200 * 4


def test_webp_source_stays_webp(fl):
    import ctypes as C
    lib = fl.load_library()
    img = fl.flgpu_image(None, 64 * 64 * 4, 64, 64, 4, 0)
    plan, k = fl.flgpu_plan(), C.c_int(-1)
    assert lib.flgpu_process_image_plan(C.byref(img), 1, b"w=32&h=32", 0, fl.IN_WEBP, C.byref(plan), C.byref(k)) == fl.OK
    assert k.value == fl.RESULT_WEBP_PLANES                       # handler.rs:286-297 with the default quality 75
    assert lib.flgpu_process_image_plan(C.byref(img), 1, b"w=32&h=32&quality=100", 0, fl.IN_WEBP, C.byref(plan), C.byref(k)) == fl.OK
    assert k.value == fl.RESULT_PIXELS                            # q == 100: lossless encoder takes the pixels (handler.rs:289-292)
    assert lib.flgpu_process_image_plan(C.byref(img), 1, b"w=oops", 0, fl.IN_WEBP, C.byref(plan), C.byref(k)) == fl.ERR_PARSE


@pytest.mark.gpu
@pytest.mark.parametrize("src,query,status,mime,kind", CASES)
def test_request_table_on_device(fl, gpu_state, oracle, src, query, status, mime, kind):
    PIL = pytest.importorskip("PIL.Image")
    import oracle_lib
    img = synth.photo(512, 512, 4 if src == "gif" else 3, index=5)
    if status == 400:
        with pytest.raises(fl.FanlinError) as e:
            gpu_state.process_image(img, query, _accept(fl), _fmt(fl, src))
        assert e.value.status == fl.ERR_PARSE
        return
    got_mime, got_kind, payload = gpu_state.process_image(img, query, _accept(fl), _fmt(fl, src))
    assert got_mime == mime and got_kind == getattr(fl, "RESULT_" + kind)
    if kind == "AS_IS":
        assert payload is None
        return
    nearest = src == "gif"
    import parity
    if nearest:
        want_px = oracle.process_pixels(img, 300, 200, arith=oracle_lib.ARITH_FMA, filter=oracle_lib.FILTER_NEAREST)
    else:
        want_px = parity.expected_pixels(fl, gpu_state, oracle, img, w=300, h=200)
    if kind == "JPEG_STREAM":
        assert payload == oracle.jpeg_encode(want_px, 75)
        assert PIL.open(io.BytesIO(payload)).size == (300, 200)
    elif kind == "WEBP_PLANES":
        y, u, v, _ = oracle.webp_yuv420(want_px)
        assert np.array_equal(payload.y, y) and np.array_equal(payload.u, u) and np.array_equal(payload.v, v)
    else:
        assert np.array_equal(payload, want_px)
