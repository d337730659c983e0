"""The CPU oracle against everything it can be pinned to without running the reference:
integer known answers, geometry known answers, algebraic invariants of the resample, an independent
float64 numpy restatement, real libwebp output (committed fixture), a loose Pillow sanity bound and
its own frozen outputs.  (The reference holds no pixel-level test vectors: PARITY UNPINNED, DESIGN.md.)"""
import os

import numpy as np
import pytest

import np_restatement as npr
import oracle_lib
import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def maxdiff(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    return int(np.abs(a.astype(np.int16) - b.astype(np.int16)).max()) if a.size else 0


# ---- integer / geometry known answers --------------------------------------

def test_grayscale_known_answers(oracle):
    px = np.array([[[255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 1, 1]]], np.uint8)
    assert oracle.grayscale(px)[0, :, 0].tolist() == [255, 54, 182, 18, 1]
    rgba = np.array([[[255, 0, 0, 77]]], np.uint8)
    assert oracle.grayscale(rgba)[0, 0].tolist() == [54, 77]          # Rgba8 -> LumaA8, alpha kept
    luma = synth.uniform(5, 7, 1)
    assert np.array_equal(oracle.grayscale(luma), luma)               # Luma8 unchanged


def test_invert_keeps_alpha(oracle):
    img = synth.uniform(6, 5, 4)
    out = oracle.invert(img)
    assert np.array_equal(out[..., :3], 255 - img[..., :3]) and np.array_equal(out[..., 3], img[..., 3])
    la = synth.uniform(6, 5, 2)
    out = oracle.invert(la)
    assert np.array_equal(out[..., 0], 255 - la[..., 0]) and np.array_equal(out[..., 1], la[..., 1])


# EXIF orientation (handler.rs:206,221-223): image 0.25.6 Orientation::from_exif maps 1..8 onto
# NoTransforms, FlipHorizontal, Rotate180, FlipVertical, Rotate90FlipH, Rotate90, Rotate270FlipH, Rotate270;
# the numpy forms below are the same permutations stated independently (rot90 with k=-1 is clockwise).
NP_ORIENT = {
    1: lambda a: a,
    2: lambda a: a[:, ::-1],
    3: lambda a: a[::-1, ::-1],
    4: lambda a: a[::-1],
    5: lambda a: np.rot90(a, -1)[:, ::-1],
    6: lambda a: np.rot90(a, -1),
    7: lambda a: np.rot90(a, 1)[:, ::-1],
    8: lambda a: np.rot90(a, 1),
}


@pytest.mark.parametrize("exif", range(0, 9))
@pytest.mark.parametrize("c", [1, 3, 4])
def test_apply_orientation_permutations(oracle, exif, c):
    img = synth.uniform(7, 11, c, index=exif)
    want = np.ascontiguousarray(NP_ORIENT.get(exif, NP_ORIENT[1])(img))
    assert np.array_equal(oracle.apply_orientation(img, exif), want)


def test_nearest_filter_picks_source_pixels(oracle):
    # process_gif (handler.rs:338-340): FilterType::Nearest = support 0, one tap of weight 1 at floor((o + 0.5) * in / out)
    img = synth.uniform(40, 64, 4)
    got = oracle.process_pixels(img, 20, 20, crop=True, filter=oracle_lib.FILTER_NEAREST)      # covering size 32x20, crop x = 6
    ys = np.minimum(np.floor((np.arange(20, dtype=np.float32) + np.float32(0.5)) * np.float32(40 / 20)).astype(int), 39)
    xs = np.minimum(np.floor((np.arange(32, dtype=np.float32) + np.float32(0.5)) * np.float32(64 / 32)).astype(int), 63)
    assert np.array_equal(got, img[ys][:, xs][:, 6:26])
    up = oracle.process_pixels(img[:5, :7], 14, 10, crop=True, filter=oracle_lib.FILTER_NEAREST)
    assert np.array_equal(up, np.repeat(np.repeat(img[:5, :7], 2, axis=0), 2, axis=1))


def test_orientation_known_answer(oracle):
    # a camera held in portrait stores EXIF 6: the stored top-left pixel ends up top-right
    img = np.arange(6, dtype=np.uint8).reshape(2, 3, 1)          # rows [0 1 2] / [3 4 5]
    assert oracle.apply_orientation(img, 6)[:, :, 0].tolist() == [[3, 0], [4, 1], [5, 2]]
    assert oracle.apply_orientation(img, 8)[:, :, 0].tolist() == [[2, 5], [1, 4], [0, 3]]
    assert oracle.apply_orientation(img, 5)[:, :, 0].tolist() == [[0, 3], [1, 4], [2, 5]]
    assert oracle.apply_orientation(img, 7)[:, :, 0].tolist() == [[5, 2], [4, 1], [3, 0]]


def test_orientation_runs_before_everything_else(oracle):
    img = synth.photo(60, 90, 3)
    got = oracle.process_pixels(img, 40, 40, orientation=6, grayscale=True)
    want = oracle.process_pixels(np.ascontiguousarray(np.rot90(img, -1)), 40, 40, grayscale=True)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("args,want", [
    ((1920, 1080, 300, 200, False), (300, 169)), ((1920, 1080, 300, 200, True), (356, 200)),
    ((512, 512, 300, 200, False), (200, 200)), ((512, 512, 300, 200, True), (300, 300)),
    ((3840, 2160, 300, 200, False), (300, 169)), ((160, 120, 300, 200, False), (267, 200)),
    ((100, 200, 200, 500, True), (250, 500)), ((200, 100, 500, 200, True), (500, 250)),   # image crate's own unit tests
    ((1, 1000, 300, 200, False), (1, 200)),
])
def test_resize_dimensions(oracle, fl, args, want):
    assert oracle.resize_dimensions(*args) == want
    p = fl.make_params(args[2], args[3], crop=args[4])
    plan = fl.plan_output(p, args[0], args[1], 3)
    assert (plan.resized_w, plan.resized_h) == want


@pytest.mark.parametrize("n_in,n_out,total,lo,hi", [
    (1080, 169, 6588, 23, 40), (1920, 300, 11707, 23, 40), (2160, 169, 13010, 45, 78)])
def test_tap_counts(oracle, n_in, n_out, total, lo, hi):
    left, count, off, w = oracle.build_weights(n_in, n_out)
    assert int(count.sum()) == total and int(count.min()) == lo and int(count.max()) == hi
    for o in range(n_out):
        assert abs(float(w[off[o]:off[o] + count[o]].astype(np.float64).sum()) - 1.0) < 1e-5


def test_blur_window(oracle):
    for sigma, taps in [(10.0, 41), (20.0, 81)]:
        left, count, off, w = oracle.build_weights(200, 200, oracle_lib.FILTER_GAUSSIAN, sigma)
        assert int(count[100]) == taps and int(left[100]) == 100 - (taps - 1) // 2
        assert int(count[0]) == (taps + 1) // 2          # truncated at the border, then renormalised
        assert abs(float(w[off[0]:off[0] + count[0]].sum()) - 1.0) < 1e-5


def test_letterbox_geometry(oracle):
    fill = (7, 8, 9)
    top = synth.uniform(169, 300, 3)
    out = oracle.letterbox(top, 300, 200, fill)
    assert out.shape == (200, 300, 4)
    assert (out[:15] == np.array(fill + (255,), np.uint8)).all() and (out[184:] == np.array(fill + (255,), np.uint8)).all()
    assert np.array_equal(out[15:184, :, :3], top) and (out[15:184, :, 3] == 255).all()
    sq = synth.uniform(200, 200, 1)
    out = oracle.letterbox(sq, 300, 200, fill)
    assert np.array_equal(out[:, 50:250, 0], sq[..., 0]) and np.array_equal(out[:, 50:250, 2], sq[..., 0])
    # translucent source: alpha 0 keeps the fill, alpha 255 replaces, in between blends and truncates
    rgba = np.zeros((1, 3, 4), np.uint8)
    rgba[0, 0] = (200, 100, 50, 0)
    rgba[0, 1] = (200, 100, 50, 255)
    rgba[0, 2] = (200, 100, 50, 128)
    out = oracle.letterbox(rgba, 3, 3, (10, 20, 30))
    assert out[1, 0].tolist() == [10, 20, 30, 255] and out[1, 1].tolist() == [200, 100, 50, 255]
    a = np.float32(128) / np.float32(255)
    exp = [int(np.float32(255) * ((np.float32(c) / np.float32(255)) * a + (np.float32(b) / np.float32(255)) * (np.float32(1) - a)))
           for c, b in zip((200, 100, 50), (10, 20, 30))]
    assert max(abs(int(x) - y) for x, y in zip(out[1, 2, :3], exp)) <= 1
    # alpha_final = bg_a + fg_a - bg_a * fg_a in f32 is not always exactly 1.0; the truncating cast then gives 254
    af = (np.float32(1) + a) - np.float32(1) * a
    assert out[1, 2, 3] == int(np.float32(255) * af)


# ---- invariants of the resample ----------------------------------------------------------------------

@pytest.mark.parametrize("arith", [oracle_lib.ARITH_REF, oracle_lib.ARITH_FMA])
def test_constant_and_symmetry(oracle, arith):
    for v in (0, 1, 128, 255):
        img = np.full((120, 160, 3), v, np.uint8)
        assert (oracle.resize_exact(img, 37, 29, arith) == v).all()
        assert (oracle.blur(img, 10.0, arith) == v).all()
    # Luma input resampled == any channel of the replicated RGB input
    g = synth.uniform(90, 70, 1, index=5)
    rgb = np.repeat(g, 3, axis=2)
    a, b = oracle.resize_exact(g, 20, 25, arith), oracle.resize_exact(rgb, 20, 25, arith)
    assert np.array_equal(a[..., 0], b[..., 1])
    # mirror-symmetric input -> mirror-symmetric output (within 1 LSB: summation order is not symmetric)
    img = synth.uniform(64, 50, 3, index=6)
    sym = np.concatenate([img, img[:, ::-1]], axis=1)
    out = oracle.resize_exact(sym, 30, 20, arith)
    assert maxdiff(out, out[:, ::-1]) <= 1


def test_fma_mode_within_one_lsb_of_reference_arithmetic(oracle):
    for i, (h, w, c, nw, nh) in enumerate([(270, 480, 3, 75, 42), (120, 160, 4, 267, 200), (333, 77, 1, 20, 90)]):
        img = synth.uniform(h, w, c, index=20 + i)
        a = oracle.resize_exact(img, nw, nh, oracle_lib.ARITH_REF)
        b = oracle.resize_exact(img, nw, nh, oracle_lib.ARITH_FMA)
        assert maxdiff(a, b) <= 1
        assert (a != b).mean() < 0.01  # the two roundings disagree only on near-ties
    img = synth.uniform(100, 150, 4, index=30)
    assert maxdiff(oracle.blur(img, 10.0, oracle_lib.ARITH_REF), oracle.blur(img, 10.0, oracle_lib.ARITH_FMA)) <= 1


# ---- independent restatement -------------------------------------------------------------------------

@pytest.mark.parametrize("shape,req", [((270, 480, 3), (75, 42)), ((120, 160, 3), (267, 200)), ((64, 64, 4), (20, 20)),
                                        ((301, 97, 1), (33, 100))])
def test_against_float64_numpy_restatement(oracle, shape, req):
    for dist in ("uniform", "photo"):
        img = getattr(synth, dist)(*shape, index=40)
        want = npr.resize_exact(img, req[0], req[1])
        got = oracle.resize_exact(img, req[0], req[1], oracle_lib.ARITH_REF)
        assert maxdiff(got, want) <= 1
        assert (got != want).mean() < 0.005


def test_blur_against_numpy_restatement(oracle):
    img = synth.photo(90, 120, 3, index=41)
    for sigma in (10.0, 20.0):
        assert maxdiff(oracle.blur(img, sigma, oracle_lib.ARITH_REF), npr.blur(img, sigma)) <= 1


def test_weights_match_float64(oracle):
    for n_in, n_out in [(1080, 169), (1920, 300), (120, 200), (512, 200)]:
        left, count, off, w = oracle.build_weights(n_in, n_out)
        m = npr.axis_matrix(n_in, n_out, npr.lanczos3, 3.0)
        for o in range(0, n_out, 7):
            assert np.allclose(w[off[o]:off[o] + count[o]], m[o, left[o]:left[o] + count[o]], atol=2e-6)
            assert np.count_nonzero(m[o]) <= count[o]


def test_pillow_sanity_bound(oracle):
    # Pillow's LANCZOS is the same kernel with different (fixed-point, 8-bit intermediate) arithmetic:
    # a loose bound that would catch a wrong convention (pixel centre, support scaling, pass order).
    from PIL import Image
    img = synth.photo(360, 640, 3, index=42)
    ours = oracle.resize_exact(img, 160, 90, oracle_lib.ARITH_REF).astype(np.int16)
    pil = np.asarray(Image.fromarray(img).resize((160, 90), Image.LANCZOS)).astype(np.int16)
    d = np.abs(ours - pil)
    assert d.max() <= 3 and d.mean() < 0.6


# ---- encoder front ends ------------------------------------------------------------------------------

def test_jpeg_ycbcr_known_answers(oracle):
    px = np.array([[[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128]]], np.uint8)
    y, cb, cr = oracle.jpeg_ycbcr444(px)
    assert y.shape == (8, 8)
    assert y[0, :6].tolist() == [0, 255, 76, 149, 29, 128]            # truncating casts
    assert cb[0, :6].tolist() == [128, 128, 84, 43, 255, 128] and cr[0, :6].tolist() == [128, 128, 255, 21, 107, 128]
    assert (y[:, 6:] == y[0, 5]).all() and (y[1:, :6] == y[0, :6]).all()  # edge replication into the 8x8 padding


def test_webp_front_end_matches_libwebp_fixture(oracle):
    # fixture produced by tools/gen_webp_golden.py from the system libwebp (WebPPictureImportRGBA + ARGBToYUVA)
    g = np.load(os.path.join(GOLDEN, "webp_yuv420_libwebp.npz"))
    names = sorted(k[:-5] for k in g.files if k.endswith("_rgba"))
    assert len(names) >= 5
    for n in names:
        y, u, v, has_alpha = oracle.webp_yuv420(g[n + "_rgba"])
        assert has_alpha == (n + "_a" in g.files)             # translucent cases carry libwebp's alpha plane
        assert np.array_equal(y, g[n + "_y"]) and np.array_equal(u, g[n + "_u"]) and np.array_equal(v, g[n + "_v"]), n
        if has_alpha:
            assert np.array_equal(g[n + "_a"], g[n + "_rgba"][:, :, 3])
    assert sum(1 for n in names if n + "_a" in g.files) >= 5


def test_ycck_loop_known_answers(oracle):
    # handler.rs:423-438 (in-repo arithmetic): clamp, truncate, K inverted
    raw = np.array([[0, 128, 128, 0], [255, 128, 128, 255], [100, 0, 255, 10], [100, 255, 0, 200]], np.uint8)
    out = oracle.ycck_to_cmyk(raw)
    def f(y, cb, cr, k):
        r = np.float32(y) + np.float32(1.402) * np.float32(cr) - np.float32(179.456)
        g = np.float32(y) - np.float32(0.34414) * np.float32(cb) - np.float32(0.71414) * np.float32(cr) + np.float32(135.45984)
        b = np.float32(y) + np.float32(1.772) * np.float32(cb) - np.float32(226.816)
        return [int(np.clip(v, 0, 255)) for v in (r, g, b)] + [255 - k]
    for i, row in enumerate(raw.tolist()):
        assert out[i].tolist() == f(*row)


# ---- frozen outputs ----------------------------------------------------------------------------------

def test_frozen_oracle_outputs(oracle):
    # tests/tools/gen_oracle_golden.py; freezes the restatement itself (not reference output)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import gen_oracle_golden as gen
    g = np.load(os.path.join(GOLDEN, "oracle_ref.npz"))
    for name, (h, w, c, dist, kw) in gen.CASES.items():
        img = g[name + "__in"]
        assert np.array_equal(oracle.process_pixels(img, arith=oracle_lib.ARITH_REF, **kw), g[name + "__ref"]), name
        assert np.array_equal(oracle.process_pixels(img, arith=oracle_lib.ARITH_FMA, **kw), g[name + "__fma"]), name
