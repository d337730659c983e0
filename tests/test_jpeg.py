"""JPEG encoder back half (reference src/handler.rs:274-278 -> image 0.25.6 codecs/jpeg/encoder.rs + transform.rs).

The oracle (oracle/fanlin_oracle_jpeg.c) restates the encoder; parity with the Rust crate itself is UNPINNED (no
Rust toolchain), so everything that CAN be checked against independent implementations is: the quantisation and
Huffman tables against the DQT / DHT segments libjpeg writes (via Pillow), the integer DCT against a float64
DCT-II, and whole streams by decoding them with libjpeg.  The GPU encoder must then equal the oracle byte for byte."""
import io
import struct

import numpy as np
import pytest

import synth

PIL = pytest.importorskip("PIL.Image")


def segments(data: bytes):
    """[(marker, payload)] up to and including SOS; then ('scan', entropy-coded bytes incl. EOI)."""
    assert data[:2] == b"\xff\xd8"
    out, i = [], 2
    while True:
        assert data[i] == 0xFF
        m = data[i + 1]
        n = struct.unpack(">H", data[i + 2:i + 4])[0]
        out.append((m, data[i + 4:i + 2 + n]))
        i += 2 + n
        if m == 0xDA:
            out.append(("scan", data[i:]))
            return out


def pillow_jpeg(img, q):
    b = io.BytesIO()
    PIL.fromarray(img).save(b, "JPEG", quality=q, subsampling=0, optimize=False)
    return b.getvalue()


def psnr(a, b):
    return 10 * np.log10(255.0 ** 2 / max(((a.astype(float) - b.astype(float)) ** 2).mean(), 1e-9))


# ----------------------------------------------------------------------------- CPU: oracle vs independent facts --

@pytest.mark.parametrize("q", [1, 10, 30, 49, 50, 75, 90, 100])
def test_tables_match_what_libjpeg_writes(oracle, q):
    ref = segments(pillow_jpeg(synth.photo(16, 16, 3), q))
    ours = segments(oracle.jpeg_encode(synth.photo(16, 16, 3), q))
    dqt_ref = b"".join(p for m, p in ref if m == 0xDB)            # libjpeg puts both tables in one segment
    dqt_ours = b"".join(p for m, p in ours if m == 0xDB)
    assert dqt_ref == dqt_ours
    dht_ref = sorted(p for m, p in ref if m == 0xC4)
    dht_ours = sorted(p for m, p in ours if m == 0xC4)
    if len(dht_ref) == 1:                                           # one segment holding all four tables
        assert dht_ref[0] == b"".join(p for m, p in ours if m == 0xC4)
    else:
        assert dht_ref == dht_ours
    luma, chroma = oracle.jpeg_qtables(q)
    assert luma.min() >= 1 and int(luma[0]) == max(1, min(255, (16 * (5000 // q if q < 50 else 200 - 2 * q) + 50) // 100))


def test_fdct_against_float64_dct(oracle):
    rng = np.random.default_rng(3)
    k = np.arange(8)
    basis = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * np.where(k[:, None] == 0, np.sqrt(0.5), 1.0) * 0.5
    worst = 0.0
    for case in range(200):
        s = rng.integers(0, 256, (8, 8)) if case > 3 else np.full((8, 8), (0, 255, 128, 127)[case])
        true = 8.0 * (basis @ (s - 128.0) @ basis.T)                # jfdctint leaves its output scaled by 8
        got = oracle.jpeg_fdct(s.astype(np.uint8))
        worst = max(worst, np.abs(got - true).max())
        assert got[0, 0] == s.sum() - 8192                          # the DC term is exact
    assert worst < 2.0


def test_header_layout(oracle):
    hdr = oracle.jpeg_header(300, 200, 75)
    assert len(hdr) == 623
    seg = segments(hdr + b"\xff\xd9")
    assert [m for m, _ in seg[:-1]] == [0xE0, 0xC0, 0xDB, 0xDB, 0xC4, 0xC4, 0xC4, 0xC4, 0xDA]
    assert seg[0][1] == b"JFIF\x00\x01\x02\x00\x00\x01\x00\x01\x00\x00"
    assert seg[1][1] == bytes([8, 0, 200, 1, 44, 3, 1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1])    # 4:4:4, 3 components
    assert seg[8][1] == bytes([3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0])


@pytest.mark.parametrize("q", [30, 75, 95])
@pytest.mark.parametrize("shape", [(200, 300, 3), (61, 83, 4), (64, 64, 1)])
def test_oracle_streams_decode_with_libjpeg(oracle, shape, q):
    img = synth.photo(*shape)
    data = oracle.jpeg_encode(img, q)
    dec = np.array(PIL.open(io.BytesIO(data)).convert("RGB"))
    rgb = img[:, :, :3] if shape[2] >= 3 else np.repeat(img[:, :, :1], 3, axis=2)
    assert dec.shape == rgb.shape
    ref = np.array(PIL.open(io.BytesIO(pillow_jpeg(rgb, q))).convert("RGB"))
    assert psnr(dec, rgb) > psnr(ref, rgb) - 1.0                   # as good as libjpeg's own encoder at this quality
    assert 0.8 < len(data) / len(pillow_jpeg(rgb, q)) < 1.25


def test_coefficients_roundtrip_through_the_stream(oracle):
    # stuffing and padding: a noisy picture at q=100 produces plenty of 0xFF bytes in the scan
    img = synth.uniform(40, 56, 3)
    data = oracle.jpeg_encode(img, 100)
    scan = segments(data)[-1][1]
    assert scan.endswith(b"\xff\xd9") and b"\xff\x00" in scan
    body = scan[:-2]
    i = 0
    while True:                                                     # every 0xFF inside the scan is followed by 0x00
        i = body.find(b"\xff", i)
        if i < 0:
            break
        assert body[i + 1] == 0
        i += 2
    dec = np.array(PIL.open(io.BytesIO(data)).convert("RGB"))
    assert psnr(dec, img) > 35


# ----------------------------------------------------------------------------- GPU: kernels vs oracle, byte for byte --

def gpu_jpeg(fl, st, img, q, **kw):
    return st.process_pixels(img, fl.make_params(quality=q, front_end=fl.FE_JPEG, **kw), capacity=img.shape[0] * img.shape[1] * 16 + 4096)


@pytest.mark.gpu
@pytest.mark.parametrize("q", [1, 30, 75, 95, 100])
@pytest.mark.parametrize("shape", [(200, 300, 4), (61, 83, 3), (8, 8, 3), (1, 1, 4), (37, 9, 1), (16, 250, 2)])
def test_gpu_stream_equals_oracle(fl, gpu_state, oracle, shape, q):
    img = synth.photo(*shape, index=q)
    got = gpu_jpeg(fl, gpu_state, img, q)
    want = oracle.jpeg_encode(img, q)
    assert got == want, f"{len(got)} vs {len(want)} bytes, first difference at {next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)}"


@pytest.mark.gpu
def test_gpu_noise_and_flat_pictures(fl, gpu_state, oracle):
    noise = synth.uniform(64, 96, 3)
    for q in (50, 100):                                             # long codes, ZRL runs, many stuffed bytes
        assert gpu_jpeg(fl, gpu_state, noise, q) == oracle.jpeg_encode(noise, q)
    big = synth.uniform(200, 300, 3, index=9)                      # streams of several 32 KB windows; at q 100 every block has more code
    for q in (75, 100):                                             # words than a wave has lanes (the per-component coding path)
        assert gpu_jpeg(fl, gpu_state, big, q) == oracle.jpeg_encode(big, q)
    for v in (0, 128, 255):                                         # DC only: every block is DC + EOB
        flat = np.full((40, 40, 3), v, np.uint8)
        assert gpu_jpeg(fl, gpu_state, flat, 75) == oracle.jpeg_encode(flat, 75)
    checker = (np.indices((48, 48)).sum(0) % 2 * 255).astype(np.uint8)[:, :, None].repeat(3, 2)
    assert gpu_jpeg(fl, gpu_state, checker, 90) == oracle.jpeg_encode(checker, 90)


@pytest.mark.gpu
def test_gpu_coefficients_match(fl, gpu_state, oracle):
    # the first kernel on its own terms: decode our stream's DC terms is not needed -- equal streams imply equal
    # coefficients; here the quantiser's rounding is probed with every sample value
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16)[:, :, None].repeat(3, 2)
    for q in (1, 50, 100):
        assert gpu_jpeg(fl, gpu_state, ramp, q) == oracle.jpeg_encode(ramp, q)


@pytest.mark.gpu
def test_gpu_pipeline_then_jpeg_config1(fl, gpu_state, oracle):
    # handler.rs in full for a JPEG source: Lanczos3 resize, letterbox, JPEG q = 75 (Query::quality default)
    img = synth.photo(1080, 1920, 3)
    p = fl.make_params(300, 200, quality=75, front_end=fl.FE_JPEG)
    got = gpu_state.process_pixels(img, p)
    pixels = gpu_state.process_pixels(img, fl.make_params(300, 200))
    assert got == oracle.jpeg_encode(pixels, 75)
    dec = np.array(PIL.open(io.BytesIO(got)).convert("RGB"))
    assert dec.shape == (200, 300, 3) and psnr(dec, pixels[:, :, :3]) > 30
    gray = gpu_state.process_pixels(img, fl.make_params(300, 169, grayscale=True, blur_sigma=10.0, quality=60, front_end=fl.FE_JPEG))
    gray_px = gpu_state.process_pixels(img, fl.make_params(300, 169, grayscale=True, blur_sigma=10.0))
    assert gray_px.shape == (169, 300, 1) and gray == oracle.jpeg_encode(gray_px, 60)


@pytest.mark.gpu
def test_gpu_stream_that_does_not_fit(fl, gpu_state, oracle):
    noise = synth.uniform(64, 64, 3)
    want = oracle.jpeg_encode(noise, 100)
    plan = fl.plan_output(fl.make_params(quality=100, front_end=fl.FE_JPEG), 64, 64, 3)
    assert plan.out_bytes < len(want) <= plan.max_out_bytes         # incompressible input beats the planning bound, never the worst case
    # a caller that offers only the planning bound learns it once the stream's length is known ...
    with pytest.raises(fl.FanlinError) as e:
        gpu_state.process_pixels(noise, fl.make_params(quality=100, front_end=fl.FE_JPEG), capacity=-int(plan.out_bytes))
    assert e.value.status == fl.ERR_BUFFER_TOO_SMALL
    # ... with exactly enough room it fits, and with plan.max_out_bytes (what the bindings allocate) it cannot fail
    exact = gpu_state.process_pixels(noise, fl.make_params(quality=100, front_end=fl.FE_JPEG), capacity=-len(want))
    assert exact == want
    assert gpu_state.process_pixels(noise, fl.make_params(quality=100, front_end=fl.FE_JPEG)) == want
    # the same through the host batch entry point and the one-call entry point (State::process_image never fails here)
    assert gpu_state.process_batch([noise, noise], [fl.make_params(quality=100, front_end=fl.FE_JPEG)] * 2) == [want, want]
    mime, kind, body = gpu_state.process_image(noise, "quality=100&blur=10", None, fl.IN_JPEG)
    assert mime == "image/jpeg" and kind == fl.RESULT_JPEG_STREAM and len(body) > 0


@pytest.mark.gpu
def test_gpu_batches_and_device_results(fl, gpu_state, oracle):
    import torch
    imgs = [synth.photo(90 + 7 * i, 120 + 5 * i, 3, index=i) for i in range(6)]
    ps = [fl.make_params(quality=40 + 10 * i, front_end=fl.FE_JPEG if i % 2 == 0 else fl.FE_NONE) for i in range(6)]
    outs = gpu_state.process_batch(imgs, ps)
    for i in range(6):
        if i % 2 == 0:
            assert outs[i] == oracle.jpeg_encode(imgs[i], 40 + 10 * i)
        else:
            assert np.array_equal(outs[i], imgs[i])
    # device-resident batch: lengths come back through flgpu_batch_results
    n = 5
    src = [torch.from_numpy(synth.photo(200, 300, 4, index=20 + i)).cuda() for i in range(n)]
    p = fl.make_params(quality=75, front_end=fl.FE_JPEG)
    cap = int(fl.plan_output(p, 300, 200, 4).out_bytes)
    dst = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(n)]
    gpu_state.process_batch_device([t.data_ptr() for t in src], [(200, 300, 4)] * n, p, [t.data_ptr() for t in dst], [cap] * n,
                                   stream=torch.cuda.current_stream().cuda_stream)
    res = gpu_state.batch_results()
    for i in range(n):
        flags, nbytes = res[i]
        assert flags & fl.IMG_ENCODED
        assert dst[i][:nbytes].cpu().numpy().tobytes() == oracle.jpeg_encode(src[i].cpu().numpy(), 75)


@pytest.mark.gpu
def test_webp_alpha_flag_reaches_the_caller(fl, gpu_state):
    img = synth.photo(32, 32, 4)
    img[:, :, 3] = 255
    assert not gpu_state.process_pixels(img, fl.make_params(front_end=fl.FE_WEBP420)).has_alpha
    img[5, 7, 3] = 254
    assert gpu_state.process_pixels(img, fl.make_params(front_end=fl.FE_WEBP420)).has_alpha


def test_integer_quantiser_equals_the_f32_formula_exhaustively():
    # jpeg_dct_quant_kernel replaces `((d / 8) as f32 / q as f32).round()` by sign(n) * ((2|n| + q) * ceil(2^32 / 2q) >> 32);
    # the claim in its comment -- identical for every |n| <= 2048 and 1 <= q <= 255 -- checked value by value
    n = np.arange(-2048, 2049, dtype=np.int64)[:, None]
    q = np.arange(1, 256, dtype=np.int64)[None, :]
    f = (n.astype(np.float32) / q.astype(np.float32)).astype(np.float32)
    t = np.trunc(f)
    want = (t + np.sign(f) * (np.abs(f - t) >= np.float32(0.5))).astype(np.int64)   # f32::round: half away from zero (f - t is exact)
    magic = ((1 << 32) + 2 * q - 1) // (2 * q)
    got = np.sign(n) * (((2 * np.abs(n) + q) * magic) >> 32)
    assert np.array_equal(got, want)
    # and the truncating `d / 8` written with shifts
    d = np.arange(-20000, 20001, dtype=np.int64)
    assert np.array_equal((d + ((d >> 63) & 7)) >> 3, np.trunc(d / 8).astype(np.int64))
