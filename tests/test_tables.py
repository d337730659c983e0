"""Host logic of the product (runs without a GPU): the weight tables the runtime uploads are bit-identical
to the oracle's, the streaming kernel's row schedule is complete, and output geometry (flgpu_plan_output)
agrees with what the oracle actually produces."""
import numpy as np
import pytest

import oracle_lib
import synth

PAIRS = [(1080, 169), (1920, 300), (2160, 169), (3840, 300), (512, 200), (120, 200), (160, 267), (1000, 1000 - 1),
         (20, 2000), (2000, 20), (7, 3), (3, 7), (1, 5), (5, 1), (333, 222), (1079, 168)]


@pytest.mark.parametrize("n_in,n_out", PAIRS)
def test_lanczos_tables_bit_identical_to_oracle(fl, oracle, n_in, n_out):
    left, count, w = fl.debug_axis_table(n_in, n_out)
    ol, oc, oo, ow = oracle.build_weights(n_in, n_out)
    assert np.array_equal(left, ol) and np.array_equal(count, oc)
    assert np.array_equal(w.view(np.uint32), ow.view(np.uint32))  # bit for bit


@pytest.mark.parametrize("n,sigma", [(200, 10.0), (300, 20.0), (17, 10.0), (1, 12.0), (1000, 15.0)])
def test_gaussian_tables_bit_identical_to_oracle(fl, oracle, n, sigma):
    left, count, w = fl.debug_axis_table(n, n, gaussian=True, sigma=sigma)
    ol, oc, oo, ow = oracle.build_weights(n, n, oracle_lib.FILTER_GAUSSIAN, sigma)
    assert np.array_equal(left, ol) and np.array_equal(count, oc) and np.array_equal(w.view(np.uint32), ow.view(np.uint32))


def test_stream_schedule(fl):
    ok, peak = fl.debug_stream_schedulable(1080, 169)
    assert ok and peak == 7                      # 1080p -> 169 rows: at most 7 output rows alive per source row
    ok, peak = fl.debug_stream_schedulable(2160, 169)
    assert ok and peak <= 8
    ok, peak = fl.debug_stream_schedulable(1080, 169, 40, 97)   # a row band
    assert ok
    ok, peak = fl.debug_stream_schedulable(120, 200)            # up-scaling keeps > 8 rows alive: generic kernels
    assert not ok and peak > 8
    for n_in, n_out in [(720, 300), (1000, 999), (4000, 100), (513, 64)]:
        ok, peak = fl.debug_stream_schedulable(n_in, n_out)
        assert ok == (peak <= 8), (n_in, n_out, peak)


def test_plan_matches_oracle_geometry(fl, oracle):
    rng = np.random.default_rng(1234)
    for _ in range(60):
        sw, sh, c = int(rng.integers(1, 90)), int(rng.integers(1, 90)), int(rng.integers(1, 5))
        w, h = int(rng.integers(1, 120)), int(rng.integers(1, 120))
        crop, gray = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        img = synth.uniform(sh, sw, c, index=int(rng.integers(0, 1000)))
        out = oracle.process_pixels(img, w, h, crop=crop, grayscale=gray)
        plan = fl.plan_output(fl.make_params(w, h, crop=crop, grayscale=gray), sw, sh, c)
        assert (plan.out_h, plan.out_w, plan.out_c) == out.shape, (sw, sh, c, w, h, crop, gray)
        assert plan.pixel_bytes == out.size


def test_plan_known_geometry(fl):
    p = fl.plan_output(fl.make_params(300, 200), 1920, 1080, 3)
    assert (p.resized_w, p.resized_h, p.letterboxed, p.place_x, p.place_y, p.out_w, p.out_h, p.out_c) == (300, 169, 1, 0, 15, 300, 200, 4)
    # EXIF 5..8 swap the source size before anything is planned (handler.rs:221-223)
    q = fl.plan_output(fl.make_params(300, 200, orientation=6), 1080, 1920, 3)
    assert (q.src_w, q.src_h, q.resized_w, q.resized_h) == (1920, 1080, 300, 169)
    q = fl.plan_output(fl.make_params(300, 200, orientation=3), 1080, 1920, 3)
    assert (q.src_w, q.src_h, q.resized_w, q.resized_h) == (1080, 1920, 113, 200)
    p = fl.plan_output(fl.make_params(300, 200, crop=True), 1920, 1080, 3)
    assert (p.resized_w, p.resized_h, p.crop_x, p.crop_y, p.letterboxed, p.out_w, p.out_h, p.out_c) == (356, 200, 28, 0, 0, 300, 200, 3)
    p = fl.plan_output(fl.make_params(300, 200), 512, 512, 3)
    assert (p.resized_w, p.resized_h, p.place_x, p.place_y) == (200, 200, 50, 0)
    p = fl.plan_output(fl.make_params(300, 200), 160, 120, 3)
    assert (p.resized_w, p.resized_h, p.place_x, p.place_y) == (267, 200, 16, 0)
    p = fl.plan_output(fl.make_params(300, 200, grayscale=True), 300, 200, 3)     # same size: no resample, Luma8 out
    assert (p.resampled, p.letterboxed, p.out_c) == (0, 0, 1)
    p = fl.plan_output(fl.make_params(300, 200, front_end=fl.FE_JFIF444), 1920, 1080, 3)
    assert (p.plane_w, p.plane_h, p.out_bytes) == (304, 200, 3 * 304 * 200)
    p = fl.plan_output(fl.make_params(301, 201, front_end=fl.FE_WEBP420), 301, 201, 3)
    assert (p.chroma_w, p.chroma_h, p.out_bytes) == (151, 101, 2 * 301 * 201 + 2 * 151 * 101)   # Y | U | V | A
    with pytest.raises(fl.FanlinError):
        fl.plan_output(fl.make_params(0, 5), 10, 10, 3)
    with pytest.raises(fl.FanlinError):
        fl.plan_output(fl.make_params(), 10, 10, 5)


def test_degenerate_cover_size_is_refused_not_guessed(fl):
    # resize_to_fill of a 1-pixel-wide source to 658x240 first builds a 658 x 1,074,514 covering image in the reference
    # (gigabytes of Rgba32F): the device path answers FLGPU_ERR_UNSUPPORTED and the handler's error arm takes over
    with pytest.raises(fl.FanlinError) as e:
        fl.plan_output(fl.make_params(658, 240, crop=True), 1, 1633, 4)
    assert e.value.status == fl.ERR_UNSUPPORTED
    assert fl.plan_output(fl.make_params(658, 240), 1, 1633, 4).out_w == 658        # without crop: 1 x 240 letterboxed, fine
