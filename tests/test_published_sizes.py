"""The only output facts the reference publishes (README.md:111-127): mean response sizes of
`GET /lenna.jpg?w=300&h=200` (JPEG, 16,021 B) and `...&webp=true&quality=20` (lossy WebP, 2,684 B).  They depend on
the whole chain -- decode, Lanczos3 resize to 200x200, letterbox to 300x200, encoder -- so they make a coarse but
independent anchor for the oracle's restatement of that chain.  Runs only where the reference checkout (its
images/lenna.jpg) is present, i.e. in the build container; never on the GPU box."""
import os

import numpy as np
import pytest

LENNA = "/root/reference/images/lenna.jpg"
PIL = pytest.importorskip("PIL.Image")
pytestmark = pytest.mark.skipif(not os.path.exists(LENNA), reason="reference checkout not present")


@pytest.fixture(scope="module")
def thumbnail(oracle):
    img = np.array(PIL.open(LENNA).convert("RGB"))           # libjpeg-turbo here, zune-jpeg there: +-1 LSB at most
    assert img.shape == (512, 512, 3)
    px = oracle.process_pixels(img, 300, 200)               # resize_dimensions -> 200x200, letterbox offset (50, 0)
    assert px.shape == (200, 300, 4) and tuple(px[0, 0]) == (32, 32, 32, 255) and tuple(px[100, 49]) == (32, 32, 32, 255)
    return px


def test_webp_q20_response_size(oracle, thumbnail):
    import webp_lib
    if webp_lib.load() is None:
        pytest.skip("libwebp not installed")
    y, u, v, _ = oracle.webp_yuv420(thumbnail)
    n = len(webp_lib.encode_planes(y, u, v, 20))
    assert abs(n - 2684) / 2684 < 0.02, n                    # 2,692 B with libwebp 1.2.2 (the reference vendors a newer one)


def test_jpeg_response_size_is_no_anchor_but_the_encoder_tracks_libjpeg(oracle, thumbnail):
    """README.md:115 publishes a mean of 16,021 B for `?w=300&h=200` (no quality parameter).  At the documented default
    (src/query.rs:18, quality 75) the restated chain gives ~12.1 KB -- and so does libjpeg at the same tables, sampling
    and Huffman codes (12.3 KB), so no faithful quality-75 encoder produces 16 KB from this picture: the README figure does
    not come from today's default.  It equals what the chain gives at quality 85 (16,011 B with libjpeg-turbo's decode of
    lenna.jpg, 16,026 B with the restated zune-jpeg decode), but that choice is made after the fact, so it ANCHORS NOTHING
    and DESIGN.md says so.  What IS checked here: the restated encoder stays within 2.5 % of libjpeg's size at identical
    quantisation tables, 4:4:4 sampling and the standard Huffman tables, at four qualities -- an independent bound on the
    whole encoder (tables, DCT, quantiser, entropy coder, framing)."""
    import io
    rgb = PIL.fromarray(thumbnail[:, :, :3])
    sizes = {}
    for q in (75, 80, 85, 90):
        ours = len(oracle.jpeg_encode(thumbnail, q))
        b = io.BytesIO()
        rgb.save(b, "JPEG", quality=q, subsampling=0, optimize=False)
        theirs = len(b.getvalue())
        sizes[q] = (ours, theirs)
        assert abs(ours - theirs) / theirs < 0.025, (q, ours, theirs)
    assert 11500 < sizes[75][0] < 12800 and 11500 < sizes[75][1] < 12800      # quality 75: ~12 KB either way, not 16 KB
    assert abs(sizes[85][0] - 16021) < 100                                      # (recorded, not relied upon)
