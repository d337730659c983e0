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


def translucent(shape, seed):
    rng = np.random.default_rng(seed)
    rgb = synth.photo(shape[0], shape[1], 3, index=seed)
    alpha = rng.integers(0, 256, (shape[0], shape[1], 1), dtype=np.uint8)
    alpha[rng.random(alpha.shape) < 0.4] = 255
    alpha[rng.random(alpha.shape) < 0.2] = 0
    return np.concatenate([rgb, alpha], axis=2)


@pytest.mark.parametrize("shape,q", [((200, 300), 75), ((61, 83), 40), ((33, 17), 90)])
def test_translucent_pictures_oracle(oracle, shape, q):
    # PNG-with-alpha -> WebP: libwebp weights the chroma of partly transparent blocks by alpha and codes an alpha plane
    rgba = translucent(shape, q)
    y, u, v, has_alpha = oracle.webp_yuv420(rgba)
    assert has_alpha
    assert webp_lib.encode_planes(y, u, v, q, a=rgba[:, :, 3]) == webp_lib.encode_rgba(rgba, q)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,q", [((200, 300), 75), ((61, 83), 40), ((33, 17), 90), ((1, 1), 50)])
def test_translucent_pictures_gpu(fl, gpu_state, oracle, shape, q):
    rgba = translucent(shape, q)
    pl = gpu_state.process_pixels(rgba, fl.make_params(quality=q, front_end=fl.FE_WEBP420))
    assert pl.has_alpha and np.array_equal(pl.a, rgba[:, :, 3])
    y, u, v, _ = oracle.webp_yuv420(rgba)
    assert np.array_equal(pl.y, y) and np.array_equal(pl.u, u) and np.array_equal(pl.v, v)
    assert webp_lib.encode_planes(pl.y, pl.u, pl.v, q, a=pl.a) == webp_lib.encode_rgba(rgba, q)
    # LumaA sources go the same way (DynamicImage::into_rgba8 first, handler.rs:287)
    la = np.ascontiguousarray(rgba[:, :, [0, 3]])
    pl2 = gpu_state.process_pixels(la, fl.make_params(quality=q, front_end=fl.FE_WEBP420))
    as_rgba = np.concatenate([la[:, :, :1]] * 3 + [la[:, :, 1:]], axis=2)
    y2, u2, v2, _ = oracle.webp_yuv420(as_rgba)
    assert pl2.has_alpha and np.array_equal(pl2.y, y2) and np.array_equal(pl2.u, u2) and np.array_equal(pl2.v, v2)


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
