"""The bars a resampled result has to clear, shared by the GPU tests.

Two kernels produce resampled pixels:
  * the streaming kernel (f32 FMA chains): required to be BIT-EXACT against the oracle's ARITH_FMA mode (same taps, same
    order, one fused multiply-add per tap), which pins every index, weight and rounding decision, and within 1 LSB of the
    reference arithmetic ARITH_REF (separate multiply and add, as rustc emits);
  * the matrix-pipe kernel (fl_mfma.hip; down-scales with 16-byte-aligned rows): its vertical sums are accumulated by the
    matrix unit in an order no CPU restatement can pin bit for bit, so its bar is the north-star tolerance itself -- EVERY
    byte within 1 LSB of ARITH_REF -- plus a bound on how many bytes may differ at all, plus equality with itself: the same
    request gives the same bytes alone, in a batch, through the queue and on every device shard.  The bound depends on the
    arithmetic the kernel runs (csrc/fl_mfma.h): FULL WIDTH (the default since round 4: no operand narrower than the
    reference's f32) differs from ARITH_REF as rarely as the f32 streaming kernel does (measured 0-30 bytes per million, the
    streaming kernel 0-23: two f32 summation orders); the PACKED arithmetic of rounds 2-3 (FLGPU_MFMA_ARITH=packed: 22-bit
    vertical weights, a 1/64-step intermediate, 14-17-bit horizontal weights) on ~0.1 % of the bytes.
Which kernel ran is read from the context's statistics, never assumed."""
import os

import numpy as np

import oracle_lib

TOL_LSB = 1              # north_star: "+-1 LSB per channel for resample/blur"
MFMA_OFF_BY_ONE = 0.006         # matrix-pipe kernel, packed arithmetic: at most 0.6 % of the bytes may differ from ARITH_REF (measured: 0.03-0.25 %)
MFMA_OFF_BY_ONE_FULL = 0.0002   # full-width arithmetic: at most 200 bytes per million (measured 0-30; the f32 streaming kernel: 0-23)


STATE = None   # the session's context (tests/conftest.py gpu_state)


def packed_arithmetic():
    """True while the session's context runs the packed arithmetic (its `mfma_arith` switch, flgpu_debug_set)."""
    return STATE is not None and STATE.debug_get("mfma_arith") == 1


def mfma_off_by_one_bar():
    return MFMA_OFF_BY_ONE if packed_arithmetic() else MFMA_OFF_BY_ONE_FULL


def maxdiff(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    return int(np.abs(a.astype(np.int16) - b.astype(np.int16)).max()) if a.size else 0


# The off-by-one RATE bar of the matrix-pipe kernels assumes pictures whose exact results do not sit on rounding boundaries.  A 1-pixel
# checkerboard scaled by ~2 is the opposite: every output is 127.5 +- a few 1e-4, so which side of the boundary a byte lands on
# is decided by the last bits of ANY arithmetic, the reference's own f32 included -- tests of such inputs switch the rate bar off
# (monkeypatch.setattr(parity, "RATE_BAR", False)) and keep the 1 LSB bar.
RATE_BAR = True


def oracle_kwargs(kw):
    return dict(w=kw.get("w"), h=kw.get("h"), fill=kw.get("fill", (32, 32, 32)), crop=kw.get("crop", False),
                blur_sigma=kw.get("blur_sigma", 0.0), grayscale=kw.get("grayscale", False), inverse=kw.get("inverse", False),
                orientation=kw.get("orientation", 0))


def check_pixels(oracle, got, img, used_mfma, **okw):
    """`got` = device pixels of the request described by okw (oracle.process_pixels keywords)."""
    want_ref = oracle.process_pixels(img, arith=oracle_lib.ARITH_REF, **okw)
    assert got.shape == want_ref.shape, (got.shape, want_ref.shape)
    d = np.abs(got.astype(np.int16) - want_ref.astype(np.int16))
    assert int(d.max()) <= TOL_LSB, f"> {TOL_LSB} LSB vs reference arithmetic {okw}"
    if used_mfma:
        if got.size >= 20000 and RATE_BAR:   # (a rate means little on a handful of pixels)
            assert float((d > 0).mean()) <= mfma_off_by_one_bar(), f"{1e6 * float((d > 0).mean()):.0f} bytes per million differ from the reference arithmetic {okw}"
    else:
        want_fma = oracle.process_pixels(img, arith=oracle_lib.ARITH_FMA, **okw)
        assert np.array_equal(got, want_fma), f"not bit-exact vs fused oracle: maxdiff {maxdiff(got, want_fma)} {okw}"


def device_pixels(fl, st, img, **kw):
    """Pixels of one request sent alone, and whether the matrix-pipe kernel produced them."""
    before = st.stats()["mfma_launches"]
    got = st.process_pixels(img, fl.make_params(**kw))
    return got, st.stats()["mfma_launches"] > before


def check_resample(fl, st, oracle, img, **kw):
    """One request through the device, against the bars of whichever kernel served it; then the SAME request with the
    matrix-pipe kernel switched off (the context's `no_mfma` switch), so that the streaming kernel keeps its own,
    bit-exact bar on every geometry the tests use."""
    got, used = device_pixels(fl, st, img, **kw)
    check_pixels(oracle, got, img, used, **oracle_kwargs(kw))
    if used:
        with st.switches(no_mfma=1):
            other, used2 = device_pixels(fl, st, img, **kw)
        assert not used2
        check_pixels(oracle, other, img, False, **oracle_kwargs(kw))
        assert maxdiff(got, other) <= TOL_LSB
    return got


def expected_pixels(fl, st, oracle, img, **kw):
    """What a batch / queue / shard / encoder test compares against: the pixels this request gives when sent alone -- after
    they have cleared their own bar against the oracle.  For the streaming kernel these ARE the oracle's ARITH_FMA pixels."""
    got, used = device_pixels(fl, st, img, **kw)
    check_pixels(oracle, got, img, used, **oracle_kwargs(kw))
    return got


def check_pixels_any_kernel(oracle, got, img, **okw):
    """For results whose context is out of reach (another process): bit-exact against ARITH_FMA, or else the matrix-pipe
    kernel's bars."""
    want_fma = oracle.process_pixels(img, arith=oracle_lib.ARITH_FMA, **okw)
    check_pixels(oracle, got, img, not np.array_equal(got, want_fma), **okw)


def mfma_model(fl, img, rw, rh):
    """A numpy restatement of the matrix-pipe kernel's arithmetic for a plain resize_exact of `img` to rw x rh (no crop, no
    letterbox), in the arithmetic the library is running (full width, or packed under the `mfma_arith` switch).
    Full width: vertical weights as three f16 terms of 2^15 w (= the f32 weight), exact vertical sums, the intermediate rounded
    (half to even) to 2^-14 around 128, horizontal weights round(w 2^hs), hs = 24, with the largest tap absorbing the rounding,
    exact integer horizontal sums, round half up, clamp.  Packed: two f16 terms of 256 w, the intermediate in 1/64 steps, hs =
    14..17.  The device differs from the model only where the matrix unit's f32 accumulation tips the rounding of an
    intermediate value AND that tips a final rounding (packed: a few bytes in ten thousand; full width: a few in a million,
    plus the 2^-20 steps in which a wave hands its part of a sum over).  Far sharper than the 1 LSB bar: it pins every index,
    weight and rounding rule of the kernel and its tables."""
    packed = packed_arithmetic()
    sh, sw, c = img.shape
    d = fl.debug_mfma_plan(sw, sh, c, rw, rh, packed=packed)
    assert d is not None
    hs = d["hs"]
    vscale, nterm, xbits = (256.0, 2, 6) if packed else (32768.0, 3, 14)
    vl, vc, vw = fl.debug_axis_table(sh, rh)
    hl, hc, hw = fl.debug_axis_table(sw, rw)
    rows = img.reshape(sh, sw * c).astype(np.float64)
    xq = np.empty((rh, sw * c), np.int64)
    o = 0
    for y in range(rh):
        n, l = int(vc[y]), int(vl[y])
        w = vw[o:o + n].astype(np.float64) * vscale
        o += n
        rest, wsum = w.copy(), np.zeros_like(w)
        for _ in range(nterm):
            t = rest.astype(np.float16).astype(np.float64)    # round to nearest even, subnormals kept: csrc/fl_mfma_tables.cpp f16_bits
            wsum += t
            rest -= t
        x = (wsum[:, None] * rows[l:l + n]).sum(axis=0) / vscale          # exact in float64: <= 35-bit weights x 8-bit pixels x <= 120 taps
        xq[y] = np.rint((x - 128.0) * float(1 << xbits)).astype(np.int64)
    xq = xq.reshape(rh, sw, c)
    out = np.empty((rh, rw, c), np.uint8)
    o = 0
    for x in range(rw):
        n, l = int(hc[x]), int(hl[x])
        w = hw[o:o + n].astype(np.float64)
        o += n
        q = np.rint(np.ldexp(w, hs))
        q = np.where(np.abs(np.ldexp(w, hs) - np.trunc(np.ldexp(w, hs))) == 0.5, np.sign(w) * np.ceil(np.abs(np.ldexp(w, hs))), q).astype(np.int64)  # llround: halves away from zero
        big = int(np.argmax(np.abs(q)))
        q[big] += (1 << hs) - int(q.sum())
        acc = (q[None, :, None] * xq[:, l:l + n, :]).sum(axis=1)           # [rh][c], exact: < 2^23 x 2^24 x 120 taps
        v = ((acc + (1 << (hs + xbits - 1))) >> (hs + xbits)) + 128
        out[:, x, :] = np.clip(v, 0, 255).astype(np.uint8)
    return out
