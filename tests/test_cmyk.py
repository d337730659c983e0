"""CMYK / YCCK JPEG sources (reference src/handler.rs:398-493): lcms2 `transform_pixels` CMYK_8 -> RGB_8.

This path is PINNED against the real thing: Little CMS 2 is a C library that exists in this image
(liblcms2.so.2), so the oracle restatement and the HIP kernel are compared with the library's own answers --
on committed golden vectors made from the reference's profiles/default.icc (tools/gen_cmyk_golden.py) and,
live, on a synthetic CMYK profile built by tests/synth_icc.py.  The bar is bit-exact."""
import os

import numpy as np
import pytest

import lcms2_lib
import synth_icc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cmyk_lcms2.npz")
needs_lcms = pytest.mark.skipif(lcms2_lib.load() is None, reason="liblcms2 not installed")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.fixture(scope="module")
def synth_profile():
    return synth_icc.cmyk_profile()


def seeded_pixels(n, seed):
    rng = np.random.default_rng(seed)
    px = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    px[: n // 8] = rng.choice(np.array([0, 1, 15, 16, 17, 128, 239, 240, 254, 255], np.uint8), (n // 8, 4))
    return px


# ------------------------------------------------------------------ CPU: the oracle is pinned --

def test_oracle_matches_lcms2_golden(oracle, golden):
    got = oracle.cmyk_to_rgb(golden["cmyk"], golden["nodes"])
    assert np.array_equal(got, golden["rgb"])


@needs_lcms
def test_oracle_matches_live_lcms2(oracle, synth_profile):
    t = lcms2_lib.Cmyk2Rgb(synth_profile)
    px = seeded_pixels(200_000, 7)
    assert np.array_equal(oracle.cmyk_to_rgb(px, t.device_link_nodes()), t.convert(px))


@needs_lcms
def test_synthetic_profile_is_a_real_colour_transform(synth_profile):
    t = lcms2_lib.Cmyk2Rgb(synth_profile)
    ink = np.array([[0, 0, 0, 0], [255, 0, 0, 0], [0, 255, 0, 0], [0, 0, 255, 0], [0, 0, 0, 255], [255, 255, 255, 255]], np.uint8)
    rgb = t.convert(ink).astype(int)
    assert rgb[0].min() > 200                                  # paper white
    assert rgb[1][0] < rgb[1][2] and rgb[2][1] < rgb[2][0]     # cyan kills red, magenta kills green
    assert rgb[3][2] < rgb[3][0] and rgb[4].max() < 100 and rgb[5].max() < 40


def test_not_a_cmyk_profile_is_rejected():
    if lcms2_lib.load() is None:
        pytest.skip("liblcms2 not installed")
    with pytest.raises(ValueError):
        lcms2_lib.Cmyk2Rgb(b"\0" * 200)


# ------------------------------------------------------------------ GPU: the kernel, through the C ABI --

@pytest.mark.gpu
def test_gpu_matches_lcms2_golden(fl, gpu_state, golden):
    gpu_state.set_cmyk_clut(golden["nodes"])
    assert np.array_equal(gpu_state.get_cmyk_clut(), golden["nodes"])
    assert np.array_equal(gpu_state.cmyk_to_rgb(golden["cmyk"]), golden["rgb"])
    # every pixel count modulo 4 (the kernel works on groups of four pixels)
    for n in (1, 2, 3, 5, 1023):
        assert np.array_equal(gpu_state.cmyk_to_rgb(golden["cmyk"][:n]), golden["rgb"][:n])


@pytest.mark.gpu
def test_gpu_profile_baking_matches_live_lcms2(fl, gpu_state, oracle, synth_profile):
    if lcms2_lib.load() is None:
        pytest.fail("the GPU box image ships liblcms2; it is required for flgpu_set_cmyk_profile")
    t = lcms2_lib.Cmyk2Rgb(synth_profile)
    gpu_state.set_cmyk_profile(synth_profile)
    assert np.array_equal(gpu_state.get_cmyk_clut(), t.device_link_nodes())
    px = seeded_pixels(1 << 20, 11)
    got = gpu_state.cmyk_to_rgb(px)
    assert np.array_equal(got, t.convert(px))                   # the real library, live
    assert np.array_equal(got, oracle.cmyk_to_rgb(px, t.device_link_nodes()))
    assert gpu_state.stats()["cmyk_pixels"] >= 1 << 20


@pytest.mark.gpu
def test_gpu_ycck_input(fl, gpu_state, oracle, golden):
    # handler.rs:423-438 then 477-491 in one kernel
    gpu_state.set_cmyk_clut(golden["nodes"])
    px = seeded_pixels(50_001, 13)
    want = oracle.cmyk_to_rgb(oracle.ycck_to_cmyk(px), golden["nodes"])
    assert np.array_equal(gpu_state.cmyk_to_rgb(px, ycck=True), want)


@pytest.mark.gpu
def test_gpu_embedded_profile_and_fallbacks(fl, golden, synth_profile):
    st = fl.State(device=0)
    try:
        px = seeded_pixels(4096, 17)
        # handler.rs:399-401 / 458: neither an embedded nor a configured profile -> None; here an error, not a guess
        with pytest.raises(fl.FanlinError) as e:
            st.cmyk_to_rgb(px)
        assert e.value.status == fl.ERR_UNSUPPORTED
        with pytest.raises(fl.FanlinError):
            st.set_cmyk_profile(b"\0" * 256)                   # with_icc_profile(..) == None
        t = lcms2_lib.Cmyk2Rgb(synth_profile)
        # embedded profile, no default: baked once, then served from the cache
        a = st.cmyk_to_rgb(px, embedded_icc=synth_profile)
        b = st.cmyk_to_rgb(px, embedded_icc=synth_profile)
        assert np.array_equal(a, t.convert(px)) and np.array_equal(a, b)
        assert st.stats()["cmyk_tables_baked"] == 1
        # unusable embedded profile -> configured profile (handler.rs:449-455)
        st.set_cmyk_clut(golden["nodes"])
        got = st.cmyk_to_rgb(golden["cmyk"][:4096], embedded_icc=b"garbage" * 40)
        assert np.array_equal(got, golden["rgb"][:4096])
        # and the embedded one still wins when it is usable
        assert np.array_equal(st.cmyk_to_rgb(px, embedded_icc=synth_profile), a)
    finally:
        st.close()


@pytest.mark.gpu
def test_gpu_device_resident_conversion(fl, gpu_state, golden):
    import torch
    gpu_state.set_cmyk_clut(golden["nodes"])
    n = 30_001
    src = torch.zeros((n + 3) // 4 * 16, dtype=torch.uint8, device="cuda")
    src[: n * 4] = torch.from_numpy(golden["cmyk"][:n].reshape(-1)).cuda()
    dst = torch.zeros((n + 3) // 4 * 12, dtype=torch.uint8, device="cuda")
    gpu_state.cmyk_to_rgb_device(src.data_ptr(), dst.data_ptr(), n, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(dst[: n * 3].cpu().numpy().reshape(n, 3), golden["rgb"][:n])
