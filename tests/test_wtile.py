"""The window-tile matrix-pipe kernel (csrc/fl_wtile.h): mild down-scales and Gaussian blurs at the full-width arithmetic.
Reference: image 0.25.6 imageops/sample.rs vertical_sample + horizontal_sample behind resize_exact and blur
(/root/reference/src/handler.rs:229-255).

Without a device: the kernel's own tables, run operand for operand on the host (oracle/wtile_model.cpp), against the oracle --
that pins the table builder (windows, f16 terms, byte digits, strips) and the arithmetic's error budget.  With a device: the
kernel's bytes against the oracle's bars for the matrix-pipe kernels (tests/parity.py) AND against that model."""
import os

import numpy as np
import pytest

import parity
import synth
import wtile_model

MODEL_CASES = [  # (source shape, resize target (w, h) or None, blur sigma)
    ((270, 480, 3), (250, 141), 0.0),     # ratio 1.92, three channels: two K-steps per tile on both axes
    ((120, 160, 3), (300, 200), 0.0),     # up-scale: weights above 1/2, horizontal scale 2^22
    ((100, 150, 1), (180, 120), 0.0),
    ((90, 120, 4), (100, 75), 0.0),
    ((333, 517, 2), (400, 258), 0.0),     # odd row pitch, two channels
    ((540, 960, 3), (500, 281), 0.0),     # several strips
    ((120, 160, 3), None, 10.0),          # blur: windows cut at every border of a small picture
    ((100, 140, 4), None, 3.0),
    ((64, 100, 3), None, 0.6),            # three taps: the centre weight is above 1/2
    ((200, 260, 3), None, 20.0),          # the largest sigma the request model allows
]


@pytest.mark.parametrize("shape,target,sigma", MODEL_CASES)
def test_host_model_of_the_tables_is_within_one_lsb_of_the_reference_arithmetic(fl, oracle, shape, target, sigma):
    img = synth.uniform(*shape, index=sum(shape))
    if target:
        r = wtile_model.run(img, target[0], target[1])
        want = oracle.resize_exact(img, target[0], target[1])
    else:
        r = wtile_model.run(img, blur_sigma=sigma)
        want = oracle.blur(img, sigma)
    assert r is not None, "the geometry is meant to fit the kernel"
    got, info = r
    want = np.asarray(want).reshape(got.shape)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert int(d.max()) <= parity.TOL_LSB
    assert float((d > 0).mean()) <= parity.MFMA_OFF_BY_ONE_FULL, f"{1e6 * float((d > 0).mean()):.0f} ppm"
    assert info["lds_bytes"] <= 150 * 1024 and info["nslot"] * info["nkmax"] == 6


def test_plan_choices(fl):
    _, down = wtile_model.run(None, 1000, 562, shape=(1080, 1920, 3))
    assert down["hs"] == 23 and down["strips"] >= 2 and down["nkmax"] in (2, 3)      # Lanczos3 at ratio 1.92: the centre weight is just above 1/2
    _, up = wtile_model.run(None, 300, 200, shape=(120, 160, 3))
    assert up["hs"] in (22, 23) and up["nkmax"] == 1                                  # an up-scale's centre tap is near 1
    _, blur = wtile_model.run(None, blur_sigma=20.0, shape=(1000, 2000, 3))
    assert blur["hs"] == 24 and blur["nslot"] == 1 and blur["nkmax"] == 6             # interior column tiles share ONE operand block
    assert blur["table_words"] < 400_000                                              # ... which is why the tables stay small
    assert wtile_model.run(None, 100, 100, shape=(1080, 1920, 3)) is None        # ratio 10.8: more rows per tile than the ring holds


DEVICE_CASES = [
    ((540, 960, 3), dict(w=600, h=400)),                  # ratio 1.6, letterboxed
    ((1080, 1920, 3), dict(w=1000, h=562)),
    ((301, 403, 3), dict(w=250, h=190, crop=True)),       # odd pitch, crop
    ((300, 400, 1), dict(w=300, h=225)),
    ((300, 400, 4), dict(w=320, h=240)),                  # translucent source onto the fill
    ((300, 400, 2), dict(w=250, h=190)),
    ((200, 300, 3), dict(blur_sigma=10.0)),
    ((250, 330, 4), dict(blur_sigma=3.0)),
    ((256, 384, 3), dict(blur_sigma=20.0)),
    ((1080, 1920, 3), dict(w=300, h=200, blur_sigma=8.0)),  # streaming matrix-pipe resample, then the blur on this kernel
    ((300, 400, 3), dict(w=250, h=190, inverse=True)),     # inverse: an XOR on the way into the LDS ring
    ((300, 400, 4), dict(w=240, h=180, inverse=True)),     # ... that leaves alpha alone
    ((300, 400, 2), dict(w=250, h=190, inverse=True, crop=True)),
    ((540, 513, 3), dict(w=300, h=200)),                   # ratio 2.7, 1539-byte rows: the streaming matrix-pipe kernel refuses them, this one is asked next
    ((1080, 1920, 3), dict(w=800, h=450)),                 # ratio 2.4: the upper end of what goes to this kernel before the fused ones
    ((1080, 1922, 3), dict(w=640, h=360)),                 # ratio 3 with rows the streaming matrix-pipe kernel cannot take (5766 bytes): asked again
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,kw", DEVICE_CASES)
def test_device_bytes_hold_the_matrix_pipe_bars_and_the_vector_kernels_their_own(fl, gpu_state, oracle, shape, kw):
    img = synth.uniform(*shape, index=shape[0] + shape[1])
    before = gpu_state.stats()["wtile_launches"]
    got, used = parity.device_pixels(fl, gpu_state, img, **kw)
    assert used and gpu_state.stats()["wtile_launches"] > before, "meant to reach the window-tile kernel"
    parity.check_pixels(oracle, got, img, True, **parity.oracle_kwargs(kw))
    again, _ = parity.device_pixels(fl, gpu_state, img, **kw)
    assert np.array_equal(got, again)
    gpu_state.debug_set("no_wtile", 1)
    try:
        before = gpu_state.stats()["wtile_launches"]
        other, used2 = parity.device_pixels(fl, gpu_state, img, **kw)
        assert gpu_state.stats()["wtile_launches"] == before
    finally:
        gpu_state.debug_set("no_wtile", 0)
    parity.check_pixels(oracle, other, img, used2, **parity.oracle_kwargs(kw))
    assert parity.maxdiff(got, other) <= parity.TOL_LSB


@pytest.mark.gpu
@pytest.mark.parametrize("shape,target,sigma", [((270, 480, 3), (250, 141), 0.0), ((540, 960, 3), (640, 360), 0.0), ((300, 400, 1), (320, 240), 0.0),
                                                ((200, 300, 3), None, 10.0), ((256, 384, 3), None, 20.0), ((250, 330, 4), None, 3.0)])
def test_device_equals_the_host_model_of_its_tables(fl, gpu_state, shape, target, sigma, monkeypatch):
    """Sharper than 1 LSB: every index, weight, digit and rounding rule.  What may differ is the matrix unit's f32 summation order in
    the vertical pass tipping the rounding of an intermediate value (2^-14 steps; the f32 sums of values above 128 have 2^-16 ones)
    AND that tipping a byte: blurs (many small taps) a few bytes in a million, mild down-scales (six taps near 1/2) a few in 100,000."""
    pass  # (every geometry the window-tile planner accepts goes to that kernel since round 4)
    img = synth.uniform(*shape, index=7 + shape[0])
    if target:
        model, _ = wtile_model.run(img, target[0], target[1])
        got, used = parity.device_pixels(fl, gpu_state, img, w=target[0], h=target[1])   # same aspect: nothing but the picture in the frame
        got = got[:, :, :shape[2]]   # Rgba8 frame of a same-aspect request: (r, g, b, 255) / (l, l, l, 255)
    else:
        model, _ = wtile_model.run(img, blur_sigma=sigma)
        got, used = parity.device_pixels(fl, gpu_state, img, blur_sigma=sigma)
    assert used
    assert got.shape == model.shape
    d = got.astype(np.int16) - model.astype(np.int16)
    assert int(np.abs(d).max()) <= 1 and float((d != 0).mean()) <= (1e-4 if target else 2e-5), f"{1e6 * float((d != 0).mean()):.1f} ppm differ from the model"


@pytest.mark.gpu
def test_batched_banded_and_alone_are_the_same_bytes(fl, gpu_state, oracle, monkeypatch):
    """A picture's bytes do not depend on the batch it travels in or on how its rows are cut into bands of M-tiles."""
    imgs = [synth.uniform(540, 960, 3, index=k) for k in range(5)]
    p = fl.make_params(w=600, h=400)
    alone = [gpu_state.process_pixels(im, p) for im in imgs]
    for a, b in zip(alone, gpu_state.process_batch(imgs, [p] * len(imgs))):
        assert np.array_equal(a, np.asarray(b).reshape(a.shape))
    for bands in ("1", "3", "7"):
        gpu_state.debug_set("force_bands", int(bands))
        assert np.array_equal(gpu_state.process_pixels(imgs[0], p), alone[0]), bands
    gpu_state.debug_set("force_bands", 0)
    blur = fl.make_params(blur_sigma=6.0)
    one = gpu_state.process_pixels(imgs[1], blur)
    gpu_state.debug_set("force_bands", 5)
    assert np.array_equal(gpu_state.process_pixels(imgs[1], blur), one)


@pytest.mark.gpu
def test_large_blur_at_the_sweep_size(fl, gpu_state, oracle):
    """2000 x 1000, sigma 20 (the largest target and sigma query.rs:20-21 allows): ten strips, 63 M-tiles, border tiles on all sides."""
    img = synth.uniform(1000, 2000, 3, index=99)
    got, used = parity.device_pixels(fl, gpu_state, img, blur_sigma=20.0)
    assert used
    parity.check_pixels(oracle, got, img, True, **parity.oracle_kwargs(dict(blur_sigma=20.0)))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_geometries_through_the_host_model(fl, oracle, seed):
    """The table builder on geometries nobody picked by hand: sources of 8 .. 200 pixels a side, 1-4 channels, ratios 0.5 .. 3.2 or
    blurs of sigma 0.3 .. 12; where the kernel takes the geometry, the host run of its tables is within 1 LSB of the reference arithmetic."""
    rng = np.random.default_rng(seed)
    taken = 0
    for k in range(25):
        c = int(rng.choice([1, 2, 3, 3, 4]))
        sh, sw = int(rng.integers(8, 200)), int(rng.integers(8, 200))
        img = synth.uniform(sh, sw, c, index=1000 * seed + k)
        if rng.integers(0, 3) == 0:
            sigma = float(rng.choice([0.3, 0.9, 2.5, 6.0, 12.0]))
            r, want = wtile_model.run(img, blur_sigma=sigma), oracle.blur(img, sigma)
            what = (sh, sw, c, "blur", sigma)
        else:
            ratio = float(rng.uniform(0.5, 3.2))
            rw, rh = max(1, int(sw / ratio)), max(1, int(sh / ratio))
            r, want = wtile_model.run(img, rw, rh), oracle.resize_exact(img, rw, rh)
            what = (sh, sw, c, rw, rh)
        if r is None:
            continue
        taken += 1
        got = r[0]
        d = np.abs(got.astype(np.int16) - np.asarray(want).reshape(got.shape).astype(np.int16))
        assert int(d.max()) <= parity.TOL_LSB, what
        assert float((d > 0).mean()) <= 0.002, what   # (small pictures: a rate means little, a systematic error would be percents)
    assert taken >= 20


def test_table_builder_and_host_model_under_address_sanitizer(tmp_path):
    """csrc/fl_mfma_tables.cpp build_wtile_plan and oracle/wtile_model.cpp over 150 random geometries (1 x 1 to 300 x 300 sources,
    1-4 channels, targets up to 400 x 400, blurs of sigma 0.3 .. 20), built with g++ -fsanitize=address,undefined (host-only code,
    no GPU): windows at picture borders, strips and operand blocks must stay inside their buffers."""
    import shutil
    import subprocess
    if shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs g++ and the HIP headers")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "fanlin-rs_amd", "csrc")
    exe = str(tmp_path / "asan_wtile_tables")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__",
                    "-I/opt/rocm/include", "-I" + csrc, os.path.join(root, "tests", "tools", "asan_wtile_tables.cpp"),
                    os.path.join(csrc, "fl_mfma_tables.cpp"), os.path.join(csrc, "fl_tables.cpp"), os.path.join(root, "oracle", "wtile_model.cpp"), "-o", exe],
                   check=True, capture_output=True, text=True)
    r = subprocess.run([exe, "150"], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stderr[-3000:]
    plans, rejected = (int(x) for x in r.stdout.split())
    assert plans > 100 and plans + rejected == 150
