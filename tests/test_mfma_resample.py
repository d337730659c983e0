"""The matrix-pipe resample kernel (csrc/fl_mfma.hip; reference: image 0.25.6 imageops/sample.rs vertical_sample +
horizontal_sample behind src/handler.rs:229-247).  Its bars are in tests/parity.py: every byte within 1 LSB of the oracle's
reference arithmetic, a bounded rate of off-by-one bytes, and bit-for-bit agreement with itself however a request reaches
the device.  The geometries below aim at the kernel's seams: 16-row tile and 32-row K-block boundaries, the 2048-byte
strips and their 16-byte alignment, partial last tiles, band splits, crops that start inside a tile, both destinations."""
import os

import numpy as np
import pytest

import parity
import synth

pytestmark = pytest.mark.gpu

GEOMETRIES = [
    # source h, w -> request w, h, crop
    (1080, 1920, 300, 200, False),   # BASELINE config 1: letterboxed Rgba8, 3 strips, 11 tiles (the last one 9 rows)
    (1080, 1920, 300, 200, True),    # resize_to_fill 356x200, centre crop: columns start inside a strip
    (1080, 1920, 300, 169, False),   # no letterbox: Rgb8 destination, byte stores
    (2160, 3840, 300, 200, False),   # 4K: ratio 12.8, 68 K-blocks
    (2160, 3840, 640, 360, False),   # ratio 6: five strips
    (720, 1280, 160, 90, False),     # ratio 8
    (1080, 1920, 480, 270, False),   # ratio 4: tiles finish every second K-block
    (1088, 1936, 333, 222, True),    # odd targets, windows of irregular length
    (600, 800, 100, 100, False),     # letterboxed on the sides
    (1080, 1920, 317, 200, False),   # 317 x 178: the short last tile ends in the same K-block as the tile before it (one extra pass)
    (1080, 1920, 320, 240, False),   # horizontal operands that do not repeat: read from the L2, not from LDS
    (1080, 1920, 150, 100, False),   # ratio 12.8 on 1080p: four strips
    (1080, 1920, 256, 144, False),   # three unequal strips: 88 + 80 + 88 columns
    (1080, 1920, 640, 360, False),   # ratio 3: the wide layout (214 pixels per strip, 93 KB of LDS output tiles, operands from the L2)
    (1080, 1920, 512, 200, True),    # wide layout with a crop and a letterbox
]

# geometries round 2's planner kept on the streaming kernel ("would not pay": four strips where 2.8 would do and horizontal
# operands read from the L2).  With the round-3 tile stage the matrix-pipe kernel is faster on all of them, the rule is gone, and
# they must clear the matrix-pipe kernel's bars like any other geometry
FORMERLY_NOT_WORTH_IT = [(333, 1024, 90, 30), (4000, 6000, 300, 200), (1080, 1920, 256, 144), (1080, 1920, 512, 288), (1080, 1920, 640, 360)]


@pytest.mark.parametrize("h,w,ow,oh,crop", GEOMETRIES)
def test_geometry_against_the_oracle_and_the_streaming_kernel(fl, gpu_state, oracle, h, w, ow, oh, crop):
    img = synth.uniform(h, w, 3, index=h + ow)
    before = gpu_state.stats()["mfma_launches"]
    got = parity.check_resample(fl, gpu_state, oracle, img, w=ow, h=oh, crop=crop)   # both kernels, each against its own bar
    assert gpu_state.stats()["mfma_launches"] > before, "this geometry is meant to reach the matrix-pipe kernel"
    again, used = parity.device_pixels(fl, gpu_state, img, w=ow, h=oh, crop=crop)
    assert used and np.array_equal(got, again)                                       # same request, same bytes


@pytest.mark.parametrize("h,w,ow,oh", FORMERLY_NOT_WORTH_IT)
def test_geometries_that_round_2_left_to_the_streaming_kernel(fl, gpu_state, oracle, h, w, ow, oh):
    img = synth.uniform(h, w, 3, index=h + ow)
    got = parity.check_resample(fl, gpu_state, oracle, img, w=ow, h=oh)              # whichever kernel serves it: that kernel's bars, and the other one's on the same request
    again, _ = parity.device_pixels(fl, gpu_state, img, w=ow, h=oh)
    assert np.array_equal(got, again)


@pytest.mark.parametrize("c,h,w,ow,oh,crop", [
    (4, 1080, 1920, 300, 200, False),   # Rgba8 with random alpha: letterboxed -> every pixel is blended onto the fill colour
    (4, 1080, 1920, 300, 169, False),   # Rgba8 -> Rgba8, no letterbox
    (4, 1080, 1920, 300, 200, True),
    (4, 1080, 1920, 600, 338, False),   # Rgba8 in the wide layout
    (1, 1080, 1920, 300, 200, False),   # Luma8 (grey JPEG sources)
    (1, 2160, 3840, 640, 360, False),
    (2, 1080, 1920, 300, 200, False),   # LumaA8
    (2, 1200, 1600, 250, 188, False),
])
def test_other_channel_counts(fl, gpu_state, oracle, c, h, w, ow, oh, crop):
    img = synth.uniform(h, w, c, index=c * 100 + ow)
    got = parity.check_resample(fl, gpu_state, oracle, img, w=ow, h=oh, crop=crop)
    again, used = parity.device_pixels(fl, gpu_state, img, w=ow, h=oh, crop=crop)
    assert used and np.array_equal(got, again)
    if c == 4:                                                                        # opaque Rgba8 must equal the Rgb8 result
        opaque = img.copy()
        opaque[..., 3] = 255
        a, _ = parity.device_pixels(fl, gpu_state, opaque, w=ow, h=oh, crop=crop)
        b, _ = parity.device_pixels(fl, gpu_state, np.ascontiguousarray(opaque[..., :3]), w=ow, h=oh, crop=crop)
        assert np.array_equal(a[..., :3], b[..., :3])


@pytest.mark.parametrize("h,w,c,rw,rh", [(1080, 1920, 3, 300, 169), (1080, 1920, 3, 352, 198), (1080, 1920, 3, 320, 180), (1080, 1920, 4, 300, 169), (2160, 3840, 1, 640, 360)])
@pytest.mark.parametrize("arith", ["full", "packed"])
def test_device_matches_the_fixed_point_model(fl, gpu_state, h, w, c, rw, rh, arith, monkeypatch):
    """Byte for byte against tests/parity.py mfma_model, in both arithmetics of the kernel: device and model may differ (by 1)
    only where the matrix unit's f32 rounding tips the rounding of an intermediate value and that tips a final rounding --
    packed arithmetic (1/64 steps): measured 1-2 bytes in 10,000 (profiles/r02_mfma_model_rate.txt); full width (2^-14 steps,
    sums handed over in 2^-20 steps): a few bytes per million."""
    if arith == "packed":
        gpu_state.debug_set("mfma_arith", 1)
    img = synth.uniform(h, w, c, index=7 * c + rw)
    got, used = parity.device_pixels(fl, gpu_state, img, w=rw, h=rh)
    assert used and got.shape == (rh, rw, c)
    want = parity.mfma_model(fl, img, rw, rh)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert int(d.max()) <= 1 and float((d > 0).mean()) < (0.0005 if arith == "packed" else 0.00002), (int(d.max()), float((d > 0).mean()))


@pytest.mark.parametrize("h,w,c,ow,oh,crop", [(1080, 1920, 3, 300, 200, False), (1080, 1920, 3, 300, 200, True), (2160, 3840, 3, 640, 360, False),
                                              (1080, 1920, 4, 300, 200, False), (1080, 1920, 1, 300, 200, False), (1080, 1920, 3, 640, 360, False)])
def test_packed_arithmetic_stays_selectable(fl, gpu_state, oracle, monkeypatch, h, w, c, ow, oh, crop):
    """Rounds 2-3's arithmetic (22-bit vertical weights, 1/64-step intermediate, 14..17-bit horizontal weights) is kept behind
    FLGPU_MFMA_ARITH=packed: every byte within 1 LSB of the reference arithmetic, at most 0.6 % of them off by one, and within
    1 LSB of what the full-width arithmetic gives for the same request."""
    img = synth.uniform(h, w, c, index=3 * c + ow)
    full, used = parity.device_pixels(fl, gpu_state, img, w=ow, h=oh, crop=crop)
    assert used
    parity.check_pixels(oracle, full, img, True, **parity.oracle_kwargs(dict(w=ow, h=oh, crop=crop)))
    gpu_state.debug_set("mfma_arith", 1)
    packed, used = parity.device_pixels(fl, gpu_state, img, w=ow, h=oh, crop=crop)
    assert used
    parity.check_pixels(oracle, packed, img, True, **parity.oracle_kwargs(dict(w=ow, h=oh, crop=crop)))
    assert parity.maxdiff(full, packed) <= 1


def test_photo_like_input_and_constant_input(fl, gpu_state, oracle):
    img = synth.photo(1080, 1920, 3, index=5)
    parity.check_resample(fl, gpu_state, oracle, img, w=300, h=200)
    for v in (0, 1, 127, 128, 200, 255):                                             # a flat picture stays exactly flat
        flat = np.full((1080, 1920, 3), v, np.uint8)
        got, used = parity.device_pixels(fl, gpu_state, flat, w=300, h=200, fill=(9, 8, 7))
        assert used and (got[15:184, :, :3] == v).all() and (got[:15, :, :3] == np.array([9, 8, 7])).all() and (got[..., 3] == 255).all()
    rng = np.random.default_rng(3)
    extreme = rng.choice(np.array([0, 255], np.uint8), size=(1080, 1920, 3))          # worst case for overshoot and for the fixed point
    parity.check_resample(fl, gpu_state, oracle, extreme, w=300, h=200)


def test_band_splits_and_batches_give_the_same_bytes(fl, gpu_state, oracle, monkeypatch):
    img = synth.uniform(1080, 1920, 3, index=77)
    alone = parity.expected_pixels(fl, gpu_state, oracle, img, w=300, h=200)
    for bands in ("1", "2", "5", "11", "16"):
        gpu_state.debug_set("force_bands", int(bands))
        got, used = parity.device_pixels(fl, gpu_state, img, w=300, h=200)
        assert used and np.array_equal(got, alone), bands
    gpu_state.debug_set("force_bands", 0)
    others = [synth.uniform(1080, 1920, 3, index=78 + i) for i in range(5)]
    outs = gpu_state.process_batch([others[0], img, others[1], img] + others[2:], [fl.make_params(300, 200)] * 7)
    assert np.array_equal(outs[1], alone) and np.array_equal(outs[3], alone)
    mixed = gpu_state.process_batch([img, synth.uniform(700, 999, 3, index=90), img],
                                    [fl.make_params(300, 200), fl.make_params(300, 200), fl.make_params(300, 200, crop=True)])
    assert np.array_equal(mixed[0], alone)                                           # next to a streaming-kernel job and another geometry


def test_requests_the_kernel_does_not_take(fl, gpu_state, oracle, monkeypatch):
    """Unaligned rows, pre-ops and mild ratios stay with the streaming / generic kernels (the window-tile matrix-pipe kernel, which
    takes the mild ratios among them since round 4, is switched off here: tests/test_wtile.py)."""
    gpu_state.debug_set("no_wtile", 1)
    cases = [(synth.uniform(540, 961, 3, index=1), dict(w=300, h=200)),            # 2883-byte rows
             (synth.uniform(540, 962, 4, index=2), dict(w=150, h=100)),            # Rgba8, 3848-byte rows: not a multiple of 16
             (synth.uniform(540, 1000, 1, index=3), dict(w=150, h=100)),           # Luma8, 1000-byte rows
             (synth.uniform(540, 960, 3, index=4), dict(w=300, h=200, grayscale=True)),
             (synth.uniform(540, 960, 3, index=5), dict(w=300, h=200, inverse=True)),
             (synth.uniform(540, 960, 3, index=6), dict(w=600, h=400)),            # ratio 1.6: more than two tiles alive per K-block
             (synth.uniform(200, 320, 3, index=7), dict(w=640, h=400))]            # up-scale
    for img, kw in cases:
        got, used = parity.device_pixels(fl, gpu_state, img, **kw)
        assert not used, kw
        parity.check_pixels(oracle, got, img, False, **parity.oracle_kwargs(kw))


def test_switch_keeps_the_streaming_kernel(fl, gpu_state, oracle, monkeypatch):
    img = synth.uniform(1080, 1920, 3, index=11)
    gpu_state.debug_set("no_mfma", 1)
    got, used = parity.device_pixels(fl, gpu_state, img, w=300, h=200)
    assert not used
    parity.check_pixels(oracle, got, img, False, w=300, h=200)


def test_expired_wait_is_an_error_not_a_picture(fl, gpu_state, monkeypatch):
    """The kernel's waves hand tiles to one another through LDS counters; every wait on one is bounded, and a wave whose wait
    expires sets the batch's device error word.  FLGPU_MFMA_SPIN_LIMIT=0 makes every such wait expire at once: the request
    must come back as FLGPU_ERR_DEVICE -- through flgpu_transform and through flgpu_transform_batch_device +
    flgpu_batch_results -- never as FLGPU_OK with pixels that were not synchronised (reference: any Err of process_image
    makes the handler serve its fallback, src/main.rs:185-195)."""
    import torch
    img = synth.uniform(1080, 1920, 3, index=123)
    good, used = parity.device_pixels(fl, gpu_state, img, w=300, h=200)
    assert used
    gpu_state.debug_set("mfma_spin_limit", 0)
    with pytest.raises(fl.FanlinError) as e:
        gpu_state.process_pixels(img, fl.make_params(300, 200))
    assert e.value.status == fl.ERR_DEVICE and "wait" in str(e.value)
    src = torch.from_numpy(img).cuda()
    dst = torch.zeros(240000, dtype=torch.uint8, device="cuda")
    gpu_state.process_batch_device([src.data_ptr()], [img.shape], fl.make_params(300, 200), [dst.data_ptr()], [240000])
    with pytest.raises(fl.FanlinError) as e:
        gpu_state.batch_results()
    assert e.value.status == fl.ERR_DEVICE
    gpu_state.debug_set("mfma_spin_limit", 1 << 22)
    again, used = parity.device_pixels(fl, gpu_state, img, w=300, h=200)   # the context is fine afterwards
    assert used and np.array_equal(again, good)
    gpu_state.process_batch_device([src.data_ptr()], [img.shape], fl.make_params(300, 200), [dst.data_ptr()], [240000])
    gpu_state.batch_results()
    assert np.array_equal(dst.cpu().numpy().reshape(200, 300, 4), good)


def test_uniform_batch_on_persistent_workgroups(fl, gpu_state, oracle):
    """Round 5: a launch of many pictures of one geometry runs on persistent workgroups that keep their strip and walk from one
    picture's rows into the next one's (light transitions: no new set-up, the last tile converted on the way, the conversion
    context of the previous picture in LDS), with the pictures left over after the whole rounds cut into row bands.  300 pictures
    of two strips each: every result equals the picture sent alone (which runs as bands on workgroups of its own)."""
    import torch
    n, h, w = 300, 96, 1024                                # 3072-byte rows: two strips; ratio 4
    imgs = [synth.uniform(h, w, 3, index=500 + k) for k in range(n)]
    p = fl.make_params(256, 24)
    src = torch.from_numpy(np.stack(imgs)).cuda()
    plan = fl.plan_output(p, w, h, 3)
    nb = int(plan.out_bytes)
    dst = torch.zeros((n, nb), dtype=torch.uint8, device="cuda")
    before = gpu_state.stats()["mfma_launches"]
    gpu_state.process_batch_device([src.data_ptr() + k * h * w * 3 for k in range(n)], [(h, w, 3)] * n, p, [dst.data_ptr() + k * nb for k in range(n)], [nb] * n)
    gpu_state.batch_results()
    assert gpu_state.stats()["mfma_launches"] == before + 1
    got = dst.cpu().numpy()
    for k in list(range(0, n, 37)) + [n - 2, n - 1]:       # whole-round pictures and banded leftovers
        alone, used = parity.device_pixels(fl, gpu_state, imgs[k], w=256, h=24)
        assert used
        assert np.array_equal(got[k].reshape(alone.shape), alone), k
    parity.check_pixels(oracle, got[5].reshape(alone.shape), imgs[5], True, **parity.oracle_kwargs(dict(w=256, h=24)))
    # the same batch again (the workgroups' LDS state is fresh per launch) and a batch whose size leaves no picture over
    dst.zero_()
    m = 256                                                  # 512 items on 256 workgroups: two whole rounds of 128 pairs
    gpu_state.process_batch_device([src.data_ptr() + k * h * w * 3 for k in range(m)], [(h, w, 3)] * m, p, [dst.data_ptr() + k * nb for k in range(m)], [nb] * m)
    gpu_state.batch_results()
    assert np.array_equal(dst[:m].cpu().numpy(), got[:m])


def test_bytes_do_not_depend_on_the_base_address_of_a_device_source(fl, gpu_state):
    """Which resample kernel serves a request is decided by the request alone: the same picture at device addresses 0, 4, 8
    and 12 bytes past a 16-byte boundary gives identical bytes through flgpu_transform_batch_device (misaligned sources of a
    matrix-pipe geometry are copied to aligned scratch; the two kernels may differ by 1 LSB, an address must not choose)."""
    import torch
    img = synth.uniform(1080, 1920, 3, index=321)
    flat = torch.zeros(img.size + 64, dtype=torch.uint8, device="cuda")
    base = flat.data_ptr()
    assert base % 16 == 0
    outs = []
    for off in (0, 4, 8, 12, 1, 7):
        flat[off:off + img.size] = torch.from_numpy(img.reshape(-1)).cuda()
        dst = torch.zeros(240000, dtype=torch.uint8, device="cuda")
        before = gpu_state.stats()["mfma_launches"]
        gpu_state.process_batch_device([base + off], [img.shape], fl.make_params(300, 200), [dst.data_ptr()], [240000])
        gpu_state.batch_results()
        assert gpu_state.stats()["mfma_launches"] == before + 1, off
        outs.append(dst.cpu().numpy())
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
