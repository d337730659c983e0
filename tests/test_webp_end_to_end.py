"""The lossy-WebP arm of process_image (reference src/handler.rs:286-305) end to end: colour front end on the
GPU, prediction + entropy coding in libwebp on the host -- and the resulting FILE compared with what the reference's
call sequence (`webp::Encoder::from_image(&img).encode(q)` = WebPPictureImportRGBA + WebPEncode) produces from the
same pixels.  Equal planes give byte-identical files, so this pins the front end against real libwebp at full
picture sizes (the committed golden vectors cover only small pictures)."""
import numpy as np
import pytest

import synth
import webp_lib

pytestmark = pytest.mark.skipif(webp_lib.load() is None, reason="libwebp not installed")


def opaque(img):
    return np.concatenate([img, np.full(img.shape[:2] + (1,), 255, np.uint8)], axis=2)


@pytest.mark.parametrize("shape,q", [((200, 300), 75), ((169, 300), 40), ((61, 83), 90), ((33, 17), 10), ((1, 1), 75)])
def test_oracle_planes_give_libwebps_own_file(oracle, shape, q):
    rgba = opaque(synth.photo(shape[0], shape[1], 3, index=q))
    y, u, v, has_alpha = oracle.webp_yuv420(rgba)
    assert not has_alpha
    assert webp_lib.encode_planes(y, u, v, q) == webp_lib.encode_rgba(rgba, q)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,q", [((200, 300), 75), ((61, 83), 90), ((33, 17), 10)])
def test_gpu_planes_give_libwebps_own_file(fl, gpu_state, shape, q):
    rgba = opaque(synth.photo(shape[0], shape[1], 3, index=q))
    pl = gpu_state.process_pixels(rgba, fl.make_params(quality=q, front_end=fl.FE_WEBP420))
    assert not pl.has_alpha
    assert webp_lib.encode_planes(pl.y, pl.u, pl.v, q) == webp_lib.encode_rgba(rgba, q)


@pytest.mark.gpu
def test_gpu_full_request_webp(fl, gpu_state):
    # "w=300&h=200&webp=true&quality=85" on a 1080p source with Accept: image/webp (BASELINE config 4's request)
    q = fl.Query.parse("w=300&h=200&webp=true&quality=85")
    params, out_format = q.to_params(fl.Format.from_accept_header("image/avif,image/webp"), input_is_jpeg=True)
    assert out_format == fl.OUT_WEBP and params.front_end == fl.FE_WEBP420
    img = synth.photo(1080, 1920, 3)
    pl = gpu_state.process_pixels(img, params)
    pixels = gpu_state.process_pixels(img, fl.make_params(300, 200))          # what the reference hands to the webp crate
    assert pixels.shape == (200, 300, 4)
    assert webp_lib.encode_planes(pl.y, pl.u, pl.v, 85) == webp_lib.encode_rgba(pixels, 85)
