"""GPU parity tests proper: every call goes through the C ABI (libfanlin_gpu.so) and is compared
with the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): bit-exact for grayscale / inverse / crop / fill; within
+-1 LSB per channel for resample and blur against the reference arithmetic (oracle
ARITH_REF = separate multiply and add, as rustc emits).  Additionally the kernels are
required to be BIT-EXACT against the oracle's ARITH_FMA mode (same taps, same order, one
fused multiply-add per tap), which pins every index, weight and rounding decision.
"""
import numpy as np
import pytest

import oracle_lib
import synth

pytestmark = pytest.mark.gpu

from parity import TOL_LSB, check_resample, expected_pixels, maxdiff  # the bars themselves: tests/parity.py


def both_bars(fl, st, oracle, got, img, **kw):
    """A result that came out of a batch, the queue or a band split: equal to the same request sent alone, which in turn
    has cleared the bars of the kernel that served it (parity.expected_pixels)."""
    want = expected_pixels(fl, st, oracle, img, **kw)
    assert got.shape == want.shape and np.array_equal(got, want), f"differs from the same request sent alone: maxdiff {maxdiff(got, want)} {kw}"


# ---------------------------------------------------------------- pointwise (bit-exact) --

@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_grayscale_bit_exact(fl, gpu_state, oracle, c):
    img = synth.uniform(37, 53, c, index=c)
    got = gpu_state.process_pixels(img, fl.make_params(grayscale=True))
    assert np.array_equal(got, oracle.grayscale(img))


@pytest.mark.parametrize("c", [1, 2, 3, 4])
@pytest.mark.parametrize("kw", [dict(grayscale=True), dict(inverse=True), dict(w=53, h=61), dict(w=61, h=37, fill=(7, 130, 250)),
                                dict(w=59, h=47, inverse=True), dict(w=53, h=37)])
def test_placement_four_pixels_per_thread_equals_the_pixel_wise_kernel(fl, gpu_state, oracle, c, kw, monkeypatch):
    """Requests without a resampling pass (pre-op only, letterbox only: 53 x 37 already fits the target) are placed by a kernel that
    handles four destination pixels per thread wherever they lie inside the picture's window, pixel by pixel at its edges: widths and
    offsets that are not multiples of four, every channel count, against the one-pixel-per-thread kernel (FLGPU_NO_PLACE4=1)."""
    img = synth.uniform(37, 53, c, index=40 + c)
    got = gpu_state.process_pixels(img, fl.make_params(**kw))
    gpu_state.debug_set("no_place4", 1)
    want = gpu_state.process_pixels(img, fl.make_params(**kw))
    assert got.shape == want.shape and np.array_equal(got, want)
    if kw == dict(grayscale=True):
        assert np.array_equal(got, oracle.grayscale(img))
    if kw == dict(inverse=True):
        assert np.array_equal(got, oracle.invert(img))


def test_grayscale_known_answers(fl, gpu_state):
    # (255,255,255)->255, (255,0,0)->54, (0,255,0)->182, (0,0,255)->18, (1,1,1)->1
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 1, 1]]], np.uint8)
    got = gpu_state.process_pixels(px, fl.make_params(grayscale=True))
    assert got[0, :, 0].tolist() == [255, 54, 182, 18, 1]


@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_inverse_bit_exact(fl, gpu_state, oracle, c):
    img = synth.uniform(41, 29, c, index=10 + c)
    got = gpu_state.process_pixels(img, fl.make_params(inverse=True))
    assert np.array_equal(got, oracle.invert(img))


def test_grayscale_wins_over_inverse(fl, gpu_state, oracle):
    img = synth.uniform(16, 16, 3)
    got = gpu_state.process_pixels(img, fl.make_params(grayscale=True, inverse=True))
    assert np.array_equal(got, oracle.grayscale(img))


@pytest.mark.parametrize("exif", range(1, 9))
@pytest.mark.parametrize("c", [1, 2, 3, 4])
def test_exif_orientation_bit_exact(fl, gpu_state, oracle, exif, c):
    # handler.rs:221-223 img.apply_orientation(): a pure permutation, odd sizes so that every edge is hit
    img = synth.uniform(37, 301, c, index=exif * 4 + c)
    got = gpu_state.process_pixels(img, fl.make_params(orientation=exif))
    assert np.array_equal(got, oracle.apply_orientation(img, exif))


@pytest.mark.parametrize("exif", [3, 6, 7])
def test_exif_orientation_then_pipeline(fl, gpu_state, oracle, exif):
    # portrait-stored 1080p photo: orientation first, then the config-1 resize + letterbox
    img = synth.photo(1920, 1080, 3, index=exif)
    check_resample(fl, gpu_state, oracle, img, w=300, h=200, orientation=exif)
    small = synth.photo(333, 250, 4, index=exif)
    check_resample(fl, gpu_state, oracle, small, w=120, h=90, crop=True, grayscale=True, blur_sigma=10.0, orientation=exif)


@pytest.mark.parametrize("shape,req,kw", [
    ((120, 160), (300, 200), {}),                              # small GIF scaled up, letterboxed
    ((480, 640), (300, 200), {}),                              # scaled down
    ((480, 640), (300, 200), dict(crop=True)),
    ((333, 517), (120, 90), dict(grayscale=True)),             # LumaA frames over the fill colour
    ((97, 61), (64, 64), dict(inverse=True, crop=True)),
    ((50, 50), (20, 1000), dict(fill=(255, 0, 7))),
])
def test_gif_frame_nearest_bit_exact(fl, gpu_state, oracle, shape, req, kw):
    # process_gif (handler.rs:327-353): frames are Rgba8 (alpha from the palette's transparent index),
    # FilterType::Nearest; one tap of weight 1.0, so REF and FMA arithmetic coincide and the bar is bit-exact
    img = synth.uniform(shape[0], shape[1], 4, index=shape[0])
    img[::3, ::5, 3] = 0            # transparent pixels keep the fill colour under overlay
    img[1::3, 1::5, 3] = 255
    got = gpu_state.process_pixels(img, fl.make_params(req[0], req[1], filter=fl.FILTER_NEAREST, **kw))
    want = oracle.process_pixels(img, req[0], req[1], filter=oracle_lib.FILTER_NEAREST, **kw)
    assert np.array_equal(got, want)
    assert np.array_equal(want, oracle.process_pixels(img, req[0], req[1], filter=oracle_lib.FILTER_NEAREST, arith=oracle_lib.ARITH_FMA, **kw))


def test_identity_copy(fl, gpu_state):
    img = synth.uniform(33, 17, 3)
    assert np.array_equal(gpu_state.process_pixels(img, fl.make_params()), img)
    # same-size request: no resample, no letterbox (handler.rs:231)
    assert np.array_equal(gpu_state.process_pixels(img, fl.make_params(17, 33)), img)


def test_letterbox_only_is_exact(fl, gpu_state, oracle):
    # 300x169 source, request 300x200: resize_dimensions gives 300x169 again -> copy + overlay
    img = synth.uniform(169, 300, 3)
    got = gpu_state.process_pixels(img, fl.make_params(300, 200, fill=(1, 2, 3)))
    want = oracle.process_pixels(img, 300, 200, fill=(1, 2, 3))
    assert np.array_equal(got, want)
    assert got[0, 0].tolist() == [1, 2, 3, 255] and np.array_equal(got[15:184, :, :3], img)


# ------------------------------------------------------------------------ resample --

@pytest.mark.parametrize("dist", ["uniform", "photo"])
def test_config1_1080p_to_300x200(fl, gpu_state, oracle, dist):
    img = getattr(synth, dist)(1080, 1920, 3, index=3)
    got = check_resample(fl, gpu_state, oracle, img, w=300, h=200)
    assert got.shape == (200, 300, 4)
    assert (got[:15] == np.array([32, 32, 32, 255], np.uint8)).all() and (got[184:] == np.array([32, 32, 32, 255], np.uint8)).all()


def test_config1_crop(fl, gpu_state, oracle):
    img = synth.uniform(1080, 1920, 3, index=4)
    got = check_resample(fl, gpu_state, oracle, img, w=300, h=200, crop=True)
    assert got.shape == (200, 300, 3)


def test_config0_512_square(fl, gpu_state, oracle):
    img = synth.photo(512, 512, 3, index=5)
    got = check_resample(fl, gpu_state, oracle, img, w=300, h=200)
    assert got.shape == (200, 300, 4) and (got[:, :50, :3] == 32).all() and (got[:, 250:, :3] == 32).all()


@pytest.mark.parametrize("pre", ["grayscale", "inverse"])
@pytest.mark.parametrize("c", [3, 4])
def test_preop_then_resample(fl, gpu_state, oracle, pre, c):
    img = synth.uniform(360, 640, c, index=6 + c)
    check_resample(fl, gpu_state, oracle, img, w=300, h=200, **{pre: True})


@pytest.mark.parametrize("shape,req", [
    ((2160, 3840, 3), (300, 200)),      # 4K
    ((120, 160, 3), (300, 200)),        # up-scale -> generic kernels
    ((511, 513, 3), (300, 200)),        # odd pitch -> generic kernels
    ((400, 600, 4), (200, 100)),        # RGBA source with random alpha: blended letterbox
    ((300, 400, 1), (120, 120)),        # Luma8
    ((300, 400, 2), (120, 120)),        # LumaA8
    ((1080, 1920, 3), (2000, 1000)),    # largest request the handler admits
    ((64, 64, 3), (20, 20)),            # smallest request
    ((1000, 30, 3), (300, 200)),        # extreme aspect ratio
])
def test_resample_shapes(fl, gpu_state, oracle, shape, req):
    img = synth.uniform(*shape, index=sum(shape))
    check_resample(fl, gpu_state, oracle, img, w=req[0], h=req[1], fill=(10, 200, 30))


@pytest.mark.parametrize("crop", [False, True])
def test_fill_colour_and_crop_geometry(fl, gpu_state, oracle, crop):
    img = synth.photo(700, 500, 3, index=12)
    check_resample(fl, gpu_state, oracle, img, w=333, h=222, crop=crop, fill=(255, 0, 128))


@pytest.mark.parametrize("w", [1919, 1366, 1001, 513])
def test_unaligned_row_pitch_uses_the_fused_kernel(fl, gpu_state, oracle, w, monkeypatch):
    # Rgb8 rows whose byte pitch is not a multiple of 4 (3 * w): funnel-shift variant of the streaming kernel
    gpu_state.debug_set("no_wtile", 1)   # (w = 513 is ratio 2.7: the window-tile kernel's by default since round 4, any pitch -- tests/test_wtile.py)
    img = synth.uniform(540, w, 3, index=w)
    before = gpu_state.stats()
    check_resample(fl, gpu_state, oracle, img, w=300, h=200)
    after = gpu_state.stats()
    assert after["resample_launches"] == before["resample_launches"] + 1 and after["generic_launches"] == before["generic_launches"]
    # an unaligned base pointer as well (a view one byte into a buffer)
    buf = np.zeros(img.size + 1, np.uint8)
    view = buf[1:].reshape(img.shape)
    view[...] = img
    got = gpu_state.process_batch([view], [fl.make_params(300, 200)])[0]
    both_bars(fl, gpu_state, oracle, got, img, w=300, h=200)


@pytest.mark.parametrize("c,kw", [(1, {}), (1, dict(inverse=True)), (2, {}), (2, dict(inverse=True)), (1, dict(grayscale=True))])
def test_luma_sources_use_the_fused_kernel(fl, gpu_state, oracle, c, kw):
    img = synth.uniform(720, 1280, c, index=90 + c)
    before = gpu_state.stats()
    check_resample(fl, gpu_state, oracle, img, w=300, h=200, **kw)
    after = gpu_state.stats()
    # one fused launch -- or two: where the matrix-pipe kernel serves the request, check_resample repeats it on the streaming kernel
    assert after["resample_launches"] - before["resample_launches"] in (1, 2) and after["generic_launches"] == before["generic_launches"]


def test_constant_image_stays_constant(fl, gpu_state):
    for v in (0, 1, 127, 255):
        img = np.full((1080, 1920, 3), v, np.uint8)
        got = gpu_state.process_pixels(img, fl.make_params(300, 200, crop=True))
        assert (got == v).all()


def test_edge_distributions(fl, gpu_state, oracle, monkeypatch):
    import parity
    for name, img in synth.edges(360, 640, 3).items():
        # (ratio 2.13: the window-tile matrix-pipe kernel since round 4.  The checkerboard's outputs all sit on the 127.5 rounding
        # boundary: 1 LSB holds, a RATE of off-by-one bytes means nothing there -- tests/parity.py RATE_BAR)
        monkeypatch.setattr(parity, "RATE_BAR", name != "checker")
        check_resample(fl, gpu_state, oracle, img, w=300, h=200)


# ----------------------------------------------------------------------------- blur --

@pytest.mark.parametrize("sigma", [10.0, 20.0])
def test_blur_only(fl, gpu_state, oracle, sigma, monkeypatch):
    gpu_state.debug_set("no_wtile", 1)   # the f32 vector blur kernel and its bit-exact bar (the matrix-pipe blur: tests/test_wtile.py)
    img = synth.uniform(200, 300, 4, index=20)
    got = gpu_state.process_pixels(img, fl.make_params(blur_sigma=sigma))
    assert np.array_equal(got, oracle.blur(img, sigma, arith=oracle_lib.ARITH_FMA))
    assert maxdiff(got, oracle.blur(img, sigma, arith=oracle_lib.ARITH_REF)) <= TOL_LSB


def test_config2_gray_resize_blur(fl, gpu_state, oracle):
    img = synth.uniform(1080, 1920, 3, index=21)
    got = check_resample(fl, gpu_state, oracle, img, w=300, h=200, grayscale=True, blur_sigma=10.0)
    assert got.shape == (200, 300, 4) and (got[..., 3] == 255).all()


@pytest.mark.parametrize("kw", [
    dict(w=300, h=200, blur_sigma=10.0),                                      # colour, letterboxed: 3 of 4 channels filtered
    dict(w=300, h=200, blur_sigma=20.0, grayscale=True, fill=(9, 9, 9)),       # grey on grey fill: 1 channel filtered
    dict(w=300, h=200, blur_sigma=12.0, grayscale=True, fill=(200, 10, 10)),   # grey picture, coloured fill: 3 channels
    dict(w=300, h=200, blur_sigma=10.0, crop=True),                            # Rgb8 output, no letterbox
    dict(w=640, h=360, blur_sigma=15.0),                                       # same size: blur of the source itself
])
def test_blur_channel_shortcuts_are_exact(fl, gpu_state, oracle, kw):
    img = synth.photo(360, 640, 3, index=22)
    check_resample(fl, gpu_state, oracle, img, **kw)


def test_blur_wide_image_tiles(fl, gpu_state, oracle, monkeypatch):
    gpu_state.debug_set("no_wtile", 1)   # (as above)
    img = synth.uniform(90, 700, 3, index=23)       # several column tiles, halo across tile borders
    got = gpu_state.process_pixels(img, fl.make_params(blur_sigma=20.0))
    assert np.array_equal(got, oracle.blur(img, 20.0, arith=oracle_lib.ARITH_FMA))
    tiny = synth.uniform(5, 7, 4, index=24)         # window larger than the image on both axes
    got = gpu_state.process_pixels(tiny, fl.make_params(blur_sigma=10.0))
    assert np.array_equal(got, oracle.blur(tiny, 10.0, arith=oracle_lib.ARITH_FMA))


# ---------------------------------------------------------------- encoder front ends --

def test_jfif444_front_end(fl, gpu_state, oracle):
    img = synth.uniform(360, 640, 3, index=30)
    pix = expected_pixels(fl, gpu_state, oracle, img, w=300, h=200)
    planes = gpu_state.process_pixels(img, fl.make_params(300, 200, front_end=fl.FE_JFIF444))
    y, cb, cr = oracle.jpeg_ycbcr444(pix)
    assert planes.y.shape == (200, 304)
    assert np.array_equal(planes.y, y) and np.array_equal(planes.u, cb) and np.array_equal(planes.v, cr)


def test_jfif444_gray_stays_three_component(fl, gpu_state, oracle):
    img = synth.uniform(97, 131, 3, index=31)
    planes = gpu_state.process_pixels(img, fl.make_params(grayscale=True, front_end=fl.FE_JFIF444))
    y, cb, cr = oracle.jpeg_ycbcr444(oracle.grayscale(img))
    assert np.array_equal(planes.y, y) and np.array_equal(planes.u, cb) and np.array_equal(planes.v, cr)


@pytest.mark.parametrize("shape", [(200, 300), (201, 301), (1, 1), (2, 5)])
def test_webp420_front_end(fl, gpu_state, oracle, shape):
    img = synth.uniform(shape[0], shape[1], 3, index=32)
    rgba = np.concatenate([img, np.full(shape + (1,), 255, np.uint8)], axis=2)
    planes = gpu_state.process_pixels(img, fl.make_params(front_end=fl.FE_WEBP420))
    y, u, v, has_alpha = oracle.webp_yuv420(rgba)
    assert not has_alpha
    assert np.array_equal(planes.y, y) and np.array_equal(planes.u, u) and np.array_equal(planes.v, v)


def test_ycck_loop_bit_exact(fl, gpu_state, oracle):
    # every (Y, Cb, Cr) combination on a coarse grid plus random pixels; K inverted
    g = np.arange(0, 256, 5, dtype=np.uint8)
    y, cb, cr = np.meshgrid(g, g, g, indexing="ij")
    grid = np.stack([y.ravel(), cb.ravel(), cr.ravel(), (y.ravel() * 7 + 3).astype(np.uint8)], axis=1)
    rnd = synth.uniform(1, 100003, 4, index=99)[0]
    raw = np.concatenate([grid, rnd], axis=0)
    assert np.array_equal(gpu_state.ycck_to_cmyk(raw), oracle.ycck_to_cmyk(raw).reshape(raw.shape))


def test_many_geometries_overflow_the_table_arena(fl, oracle, monkeypatch):
    # a deliberately tiny arena: planning 40 different geometries must reset and rebuild the table cache, not fail
    monkeypatch.setenv("FLGPU_ARENA_WORDS", "262144")  # 1 MiB
    with fl.State(device=0) as st:
        for i in range(40):
            h, w = 200 + 13 * i, 320 + 17 * i
            img = synth.uniform(h, w, 3, index=300 + i)
            got = st.process_pixels(img, fl.make_params(120 + i, 90, blur_sigma=10.0 if i % 7 == 0 else 0.0))
            both_bars(fl, st, oracle, got, img, w=120 + i, h=90, blur_sigma=10.0 if i % 7 == 0 else 0.0)
        assert st.stats()["tables_built"] > 80


# ----------------------------------------------------------------- batch + queue paths --

def test_batch_mixed_requests(fl, gpu_state, oracle):
    imgs = [synth.uniform(360, 640, 3, index=40), synth.photo(2160, 3840, 3, index=41), synth.uniform(120, 160, 3, index=42),
            synth.uniform(360, 640, 4, index=43), synth.uniform(200, 300, 3, index=44)]
    reqs = [dict(w=300, h=200), dict(w=300, h=200, crop=True), dict(w=300, h=200, fill=(9, 8, 7)),
            dict(w=100, h=100, inverse=True), dict(grayscale=True)]
    outs = gpu_state.process_batch(imgs, [fl.make_params(**r) for r in reqs])
    for img, r, got in zip(imgs, reqs, outs):
        okw = dict(w=r.get("w"), h=r.get("h"), fill=r.get("fill", (32, 32, 32)), crop=r.get("crop", False),
                   grayscale=r.get("grayscale", False), inverse=r.get("inverse", False))
        both_bars(fl, gpu_state, oracle, got, img, **okw)


def test_request_queue_concurrent_callers(fl, gpu_state, oracle):
    import threading
    imgs = [synth.uniform(360, 640, 3, index=50 + i) for i in range(24)]
    got = [None] * len(imgs)
    gate = threading.Barrier(len(imgs))   # (all callers enter together: started one after the other, a fast device serves each alone now and then)

    def call(i):
        gate.wait()
        got[i] = gpu_state.process_pixels(imgs[i], fl.make_params(300, 200))

    before = gpu_state.stats()
    ts = [threading.Thread(target=call, args=(i,)) for i in range(len(imgs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    after = gpu_state.stats()
    for g, im in zip(got, imgs):
        both_bars(fl, gpu_state, oracle, g, im, w=300, h=200)
    # 24 concurrent requests must have shared launches
    assert after["queue_flushes"] - before["queue_flushes"] < len(imgs)


def test_device_resident_batch_properties(fl, gpu_state, oracle):
    """BASELINE configs 1/3 shape (many 1080p images, HBM-resident, one call): size-independent checks --
    duplicates of an image anywhere in the batch give identical bytes, constant images stay constant, and a
    sample of the batch matches the oracle."""
    import torch
    n, h, w, c = 96, 1080, 1920, 3
    base = [torch.from_numpy(synth.uniform(h, w, c, index=70 + i)) for i in range(4)]
    src = torch.empty((n, h, w, c), dtype=torch.uint8, device="cuda")
    for i in range(n):
        src[i] = base[i % 4].cuda() if i % 8 != 7 else torch.full((h, w, c), (i * 37) % 256, dtype=torch.uint8, device="cuda")
    params = fl.make_params(300, 200, fill=(5, 6, 7))
    plan = fl.plan_output(params, w, h, c)
    stride = (int(plan.out_bytes) + 255) // 256 * 256
    dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
    gpu_state.process_batch_device([src.data_ptr() + i * h * w * c for i in range(n)], [(h, w, c)] * n, params,
                                   [dst.data_ptr() + i * stride for i in range(n)], [stride] * n,
                                   stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = dst[:, :plan.out_bytes].cpu().numpy().reshape(n, 200, 300, 4)
    for i in range(n):
        if i % 8 == 7:
            assert (out[i, 15:184, :, :3] == (i * 37) % 256).all() and (out[i, :15, :, :3] == np.array([5, 6, 7])).all()
        else:
            assert np.array_equal(out[i], out[i % 4]), i
    for i in range(4):
        both_bars(fl, gpu_state, oracle, out[i], base[i].numpy(), w=300, h=200, fill=(5, 6, 7))


def _full_size_batch(fl, gpu_state, params, n=1024):
    """The exact batch bench.py times: n x 1920x1080 RGB8, device resident, one call.  A few distinct pictures are
    repeated through the batch (6.4 GB of distinct noise would only test the random generator) plus constants."""
    import torch
    h, w, c = 1080, 1920, 3
    base = [synth.uniform(h, w, c, index=170 + i) for i in range(6)]
    src = torch.empty((n, h, w, c), dtype=torch.uint8, device="cuda")
    dev = [torch.from_numpy(b).cuda() for b in base]
    for i in range(n):
        if i % 16 == 15:
            src[i].fill_((i * 29) % 256)
        else:
            src[i] = dev[(i * 5) % 6]
    plan = fl.plan_output(params, w, h, c)
    stride = (int(plan.max_out_bytes) + 255) // 256 * 256
    dst = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
    before = gpu_state.stats()
    gpu_state.process_batch_device([src.data_ptr() + i * h * w * c for i in range(n)], [(h, w, c)] * n, params,
                                   [dst.data_ptr() + i * stride for i in range(n)], [stride] * n,
                                   stream=torch.cuda.current_stream().cuda_stream)
    res = gpu_state.batch_results()
    torch.cuda.synchronize()
    after = gpu_state.stats()
    assert after["images"] - before["images"] == n
    return base, dst, res, plan


def test_config1_at_bench_size(fl, gpu_state, oracle):
    """BASELINE config 1 at the size bench.py runs it: 1024 x 1080p -> w=300&h=200 + letterbox + JPEG encode."""
    n = 1024
    params = fl.make_params(300, 200, quality=75, front_end=fl.FE_JPEG)
    base, dst, res, plan = _full_size_batch(fl, gpu_state, params, n)
    out = dst.cpu().numpy()
    first = {}
    for i in range(n):
        flags, nb = res[i]
        assert flags & fl.IMG_ENCODED and nb > 600
        if i % 16 == 15:
            continue
        k = (i * 5) % 6
        if k in first:
            j = first[k]
            assert nb == res[j][1] and np.array_equal(out[i, :nb], out[j, :nb]), i     # duplicates anywhere in the batch: identical streams
        else:
            first[k] = i
    for k, i in first.items():                                                           # every distinct picture against the oracle chain
        px = expected_pixels(fl, gpu_state, oracle, base[k], w=300, h=200)     # (checked against the reference arithmetic in there)
        assert out[i, :res[i][1]].tobytes() == oracle.jpeg_encode(px, 75)
    i = 15                                                                                # a constant picture: flat 300x169 on the fill colour
    v = (i * 29) % 256
    flat = np.full((200, 300, 4), 255, np.uint8)
    flat[:, :, :3] = 32
    flat[15:184, :, :3] = v
    assert out[i, :res[i][1]].tobytes() == oracle.jpeg_encode(flat, 75)


def test_config2_at_bench_size(fl, gpu_state, oracle):
    """BASELINE config 2 at full size: the same 1024-image batch with grayscale=true & blur=10."""
    n = 1024
    params = fl.make_params(300, 200, grayscale=True, blur_sigma=10.0)
    base, dst, res, plan = _full_size_batch(fl, gpu_state, params, n)
    out = dst[:, :plan.out_bytes].cpu().numpy().reshape(n, 200, 300, 4)
    first = {}
    for i in range(n):
        if i % 16 == 15:
            continue
        k = (i * 5) % 6
        if k in first:
            assert np.array_equal(out[i], out[first[k]]), i
        else:
            first[k] = i
    for k, i in first.items():
        both_bars(fl, gpu_state, oracle, out[i], base[k], w=300, h=200, grayscale=True, blur_sigma=10.0)
    assert (out[..., 3] == 255).all()


def test_mixed_size_batch_with_webp_front_end(fl, gpu_state, oracle):
    """BASELINE config 4 shape: 4K / 1080p / thumbnail sources in one batch (1:6:3), webp=true quality=85."""
    shapes = [(2160, 3840)] * 1 + [(1080, 1920)] * 6 + [(120, 160)] * 3
    order = [3, 0, 7, 9, 1, 4, 8, 2, 5, 6]                      # seed-shuffled
    imgs = [synth.photo(*shapes[k], 3, index=80 + k) for k in order]
    q = fl.Query.parse("w=300&h=200&webp=true&quality=85")
    p, fmt = q.to_params(fl.Format(fl.ACCEPT_WEBP), input_is_jpeg=True)
    assert fmt == fl.OUT_WEBP and p.front_end == fl.FE_WEBP420
    outs = gpu_state.process_batch(imgs, [p] * len(imgs))
    for img, planes in zip(imgs, outs):
        pix = expected_pixels(fl, gpu_state, oracle, img, w=300, h=200)
        y, u, v, has_alpha = oracle.webp_yuv420(pix)
        assert not has_alpha and planes.y.shape == (200, 300) and planes.u.shape == (100, 150)
        assert np.array_equal(planes.y, y) and np.array_equal(planes.u, u) and np.array_equal(planes.v, v)


def test_band_split_small_batch_matches(fl, gpu_state, oracle, monkeypatch):
    img = synth.uniform(1080, 1920, 3, index=60)
    for bands in ("1", "3", "7"):
        gpu_state.debug_set("force_bands", int(bands))
        both_bars(fl, gpu_state, oracle, gpu_state.process_pixels(img, fl.make_params(300, 200)), img, w=300, h=200)


def test_generic_and_stream_kernels_agree(fl, gpu_state, monkeypatch):
    img = synth.uniform(1080, 1920, 3, index=61)
    m = gpu_state.process_pixels(img, fl.make_params(300, 200))           # the matrix-pipe kernel
    gpu_state.debug_set("no_mfma", 1)
    a = gpu_state.process_pixels(img, fl.make_params(300, 200))           # the streaming kernel
    gpu_state.debug_set("force_generic", 1)
    b = gpu_state.process_pixels(img, fl.make_params(300, 200))           # the two-pass generic kernels
    assert np.array_equal(a, b) and maxdiff(m, a) <= TOL_LSB and not np.array_equal(m, a)


def test_errors_are_reported_not_swallowed(fl, gpu_state):
    img = synth.uniform(8, 8, 3)
    with pytest.raises(fl.FanlinError):
        gpu_state.process_pixels(img, fl.make_params(0, 10))
    with pytest.raises(ValueError):
        gpu_state.process_pixels(np.zeros((4, 4, 5), np.uint8), fl.make_params())


def test_native_kernels_were_used(gpu_state):
    s = gpu_state.stats()
    assert s["resample_launches"] > 0 and s["generic_launches"] > 0 and s["blur_launches"] > 0 and s["frontend_launches"] > 0


# ---------------------------------------------------------------- seeded sweep over request space --

def _sweep_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        sh, sw = int(rng.integers(20, 640)), int(rng.integers(20, 900))
        c = int(rng.choice([1, 2, 3, 3, 3, 4]))
        w, h = int(rng.integers(20, 500)), int(rng.integers(20, 400))
        kw = dict(w=w, h=h, crop=bool(rng.integers(0, 2)), grayscale=bool(rng.integers(0, 4) == 0), inverse=bool(rng.integers(0, 4) == 0),
                  fill=tuple(int(x) for x in rng.integers(0, 256, 3)), orientation=int(rng.choice([1, 1, 1, 3, 6, 8])))
        if rng.integers(0, 5) == 0:
            kw["blur_sigma"] = float(rng.choice([10.0, 14.0, 20.0]))
        cases.append((i, (sh, sw, c), kw))
    return cases


@pytest.mark.parametrize("idx,shape,kw", _sweep_cases(48, 0xFA17), ids=lambda v: str(v) if isinstance(v, int) else None)
def test_seeded_request_sweep(fl, gpu_state, oracle, idx, shape, kw):
    # whatever the request, the device path must equal the fused-arithmetic oracle bit for bit and stay within
    # 1 LSB of the reference arithmetic: exercises up- and down-scales, strips, bands, unaligned rows, all layouts
    img = synth.photo(*shape, index=3000 + idx) if idx % 2 else synth.uniform(*shape, index=3000 + idx)
    check_resample(fl, gpu_state, oracle, img, **kw)


def test_seeded_sweep_as_one_mixed_batch(fl, gpu_state, oracle):
    cases = _sweep_cases(24, 0xBA7C)
    imgs = [synth.uniform(*shape, index=4000 + i) for i, shape, _ in cases]
    ps = [fl.make_params(**kw) for _, _, kw in cases]
    outs = gpu_state.process_batch(imgs, ps)
    for (i, shape, kw), img, got in zip(cases, imgs, outs):
        both_bars(fl, gpu_state, oracle, got, img, **kw)


@pytest.mark.parametrize("shape,kw", [
    ((1080, 1920, 3), dict(w=1000, h=562)),                      # ratio 1.92: too mild for the fused kernels' row schedules
    ((1080, 1920, 3), dict(w=2000, h=1000)),                     # the largest target the size gate allows: a slight up-scale
    ((120, 160, 3), dict(w=300, h=200)),                         # BASELINE config 4's thumbnails, up-scaled and letterboxed
    ((540, 961, 3), dict(w=300, h=200)),                         # 2883-byte rows
    ((333, 250, 4), dict(w=120, h=90, crop=True, grayscale=True)),
    ((200, 320, 1), dict(w=640, h=400)),
    ((97, 61, 2), dict(w=64, h=64, inverse=True, crop=True)),
    ((700, 999, 3), dict(w=999, h=700)),                         # identity size: no resampling at all
])
def test_two_pass_resample_through_an_lds_tile_equals_the_one_through_hbm(fl, gpu_state, oracle, monkeypatch, shape, kw):
    """Round 3: requests neither fused kernel takes run their two passes through an LDS tile (resample_tile_kernel) instead of an
    f32 intermediate in HBM.  Same arithmetic, same order: bit-identical to the HBM form (FLGPU_NO_TILE=1) and to the oracle's
    fused-order mode, within 1 LSB of the reference arithmetic (parity.check_pixels holds both bars)."""
    import parity
    gpu_state.debug_set("no_wtile", 1)   # (round 4: down-scales among these go to the window-tile matrix-pipe kernel by default, tests/test_wtile.py)
    img = synth.uniform(*shape, index=shape[0] + shape[1])
    got, used = parity.device_pixels(fl, gpu_state, img, **kw)
    assert not used
    parity.check_pixels(oracle, got, img, False, **parity.oracle_kwargs(kw))
    gpu_state.debug_set("no_tile", 1)
    other, _ = parity.device_pixels(fl, gpu_state, img, **kw)
    assert np.array_equal(got, other)


def test_more_callers_than_the_regular_lanes_hold(fl, gpu_state):
    """Default queue: 4 lanes x 16 requests, plus overflow lanes that take the full batches waiting behind them (csrc/fl_queue.cpp).  112 callers at
    once: every result equals the one the same request gives alone, whichever lane served it."""
    import threading
    imgs = [synth.uniform(270, 480, 3, index=200 + i % 7) for i in range(112)]
    params = [fl.make_params(120 + 8 * (i % 3), 90) for i in range(len(imgs))]
    alone = {}
    for i in range(7 * 3):
        alone[(i % 7, i % 3)] = gpu_state.process_pixels(imgs[i % 7], fl.make_params(120 + 8 * (i % 3), 90))
    got = [None] * len(imgs)
    gate = threading.Barrier(len(imgs))

    def call(i):
        gate.wait()
        for _ in range(3): got[i] = gpu_state.process_pixels(imgs[i], params[i])

    before = gpu_state.stats()
    ts = [threading.Thread(target=call, args=(i,)) for i in range(len(imgs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    after = gpu_state.stats()
    for i, g in enumerate(got):
        assert np.array_equal(np.asarray(g), np.asarray(alone[(i % 7, i % 3)])), i
    assert after["queue_flushes"] - before["queue_flushes"] < 3 * len(imgs) // 4
