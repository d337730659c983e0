"""Request model (host only): query::Query parsing + accessors, content::Format, handler.rs format choice.

The table below is the reference's own `test_query` table (src/query.rs:100-405) expressed as data:
query string, whether axum's Query extractor rejects it, the expected Option fields and the accessor
assertions of each case."""
import pytest

N = None


def fields(**kw):
    base = dict(w=N, h=N, rgb=N, quality=N, crop=N, blur=N, grayscale=N, inverse=N, avif=N, webp=N)
    base.update(kw)
    return base


DEFAULT_ACCESSORS = dict(dimensions=None, fill_color=(32, 32, 32), quality=75, cropping=False, blur=0.0,
                         grayscale=False, inverse=False, use_avif=False, use_webp=False, as_is=True,
                         unsupported_scale_size=False)

CASES = [
    ("http://127.0.0.1:3000", False, fields(), DEFAULT_ACCESSORS),
    ("http://127.0.0.1:3000?w=", True, None, {}),
    ("http://127.0.0.1:3000?unknown=1", False, fields(), {}),
    ("http://127.0.0.1:3000?w=2000&h=1000", False, fields(w=2000, h=1000),
     dict(dimensions=(2000, 1000), as_is=False, unsupported_scale_size=False)),
    ("http://127.0.0.1:3000?w=1618", False, fields(w=1618), dict(dimensions=None, as_is=True, unsupported_scale_size=False)),
    ("http://127.0.0.1:3000?w=2001&h=1001", False, fields(w=2001, h=1001),
     dict(dimensions=(2001, 1001), as_is=False, unsupported_scale_size=True)),
    ("http://127.0.0.1:3000?w=foo&h=bar", True, None, {}),
    ("http://127.0.0.1:3000?rgb=255,255,255", False, fields(rgb="255,255,255"), dict(fill_color=(255, 255, 255), as_is=True)),
    ("http://127.0.0.1:3000?rgb=255,255,255,255", False, fields(rgb="255,255,255,255"), dict(fill_color=(255, 255, 255), as_is=True)),
    ("http://127.0.0.1:3000?rgb=255,255", False, fields(rgb="255,255"), dict(fill_color=(32, 32, 32), as_is=True)),
    ("http://127.0.0.1:3000?rgb=foo,bar,baz", False, fields(rgb="foo,bar,baz"), dict(fill_color=(32, 32, 32), as_is=True)),
    ("http://127.0.0.1:3000?quality=50", False, fields(quality=50), dict(quality=50, as_is=True)),
    ("http://127.0.0.1:3000?quality=foo", True, None, {}),
    ("http://127.0.0.1:3000?crop=true", False, fields(crop=True), dict(cropping=True, as_is=True)),
    ("http://127.0.0.1:3000?crop=foo", True, None, {}),
    ("http://127.0.0.1:3000?blur=10", False, fields(blur=10), dict(blur=10.0, as_is=False)),
    ("http://127.0.0.1:3000?blur=foo", True, None, {}),
    ("http://127.0.0.1:3000?grayscale=true", False, fields(grayscale=True), dict(grayscale=True, as_is=False)),
    ("http://127.0.0.1:3000?grayscale=foo", True, None, {}),
    ("http://127.0.0.1:3000?inverse=true", False, fields(inverse=True), dict(inverse=True, as_is=False)),
    ("http://127.0.0.1:3000?inverse=foo", True, None, {}),
    ("http://127.0.0.1:3000?avif=true", False, fields(avif=True), dict(use_avif=True, as_is=False)),
    ("http://127.0.0.1:3000?avif=foo", True, None, {}),
    ("http://127.0.0.1:3000?webp=true", False, fields(webp=True), dict(use_webp=True, as_is=False)),
    ("http://127.0.0.1:3000?webp=foo", True, None, {}),
]


@pytest.mark.parametrize("uri,error,want,asserts", CASES, ids=[c[0].split("3000")[1] or "-" for c in CASES])
def test_query_reference_table(fl, uri, error, want, asserts):
    if error:
        with pytest.raises(fl.FanlinError) as e:
            fl.Query.parse(uri)
        assert e.value.status == 6  # FLGPU_ERR_PARSE: axum answers 400
        return
    q = fl.Query.parse(uri)
    assert q.fields() == want
    for name, value in asserts.items():
        assert getattr(q, name)() == value, name


def test_query_blur_clamp_and_odd_values(fl):
    # query.rs:59-62: any present value is clamped into 10..=20 (even blur=0)
    for raw, sigma in [("0", 10.0), ("1", 10.0), ("15", 15.0), ("20", 20.0), ("255", 20.0)]:
        assert fl.Query.parse(f"blur={raw}").blur() == sigma
    with pytest.raises(fl.FanlinError):
        fl.Query.parse("blur=256")       # u8 overflow
    with pytest.raises(fl.FanlinError):
        fl.Query.parse("w=-1&h=5")       # u32
    with pytest.raises(fl.FanlinError):
        fl.Query.parse("w=1&w=2")        # serde: duplicate field
    assert fl.Query.parse("w=%33%30%30&h=200").dimensions() == (300, 200)  # percent-decoding
    assert fl.Query.parse("rgb=1%2C2%2C3").fill_color() == (1, 2, 3)
    assert fl.Query.parse("rgb=1,,3").fill_color() == (1, 32, 3)
    assert fl.Query.parse("rgb=300,2,3").fill_color() == (32, 2, 3)  # u8 parse failure -> default per field
    # query.rs:35-49 parses the whole string, however long: u8::from_str takes any number of leading zeros, so a value longer
    # than the ABI's fixed field must not be cut off (it is reduced to the colour it means)
    assert fl.Query.parse("rgb=" + "0" * 150 + "7,8," + "0" * 90 + "9").fill_color() == (7, 8, 9)
    assert fl.Query.parse("rgb=" + "0" * 150 + "7,8").fill_color() == (32, 32, 32)      # two fields: the default colour
    assert fl.Query.parse("rgb=1,2,3," + "x" * 200).fill_color() == (1, 2, 3)            # fields past the third are ignored
    assert fl.Query.parse("rgb=1,2," + "3" * 120).fill_color() == (1, 2, 32)             # a field that overflows u8
    assert fl.Query.parse("").as_is()
    assert fl.Query.parse("crop=false").cropping() is False


def test_size_gate(fl):
    # query.rs:20-21,89-93: w in 20..=2000, h in 20..=1000, missing -> 100
    ok = ["w=20&h=20", "w=2000&h=1000", "h=20", "w=20"]
    bad = ["w=19&h=20", "w=20&h=19", "w=2001&h=20", "w=20&h=1001", "w=0", "h=5000"]
    for q in ok:
        assert not fl.Query.parse(q).unsupported_scale_size(), q
    for q in bad:
        assert fl.Query.parse(q).unsupported_scale_size(), q


def test_accept_header(fl):
    # src/main.rs:474-512
    f = fl.Format.from_accept_header("text/html,application/xhtml+xml,application/xml;q=0.9,image/avif,image/webp,"
                                     "image/apng,*/*;q=0.8,application/signed-exchange;v=b3;q=0.7")
    assert f.webp_accepted() and f.avif_accepted()
    f = fl.Format.from_accept_header("")
    assert not f.webp_accepted() and not f.avif_accepted()
    f = fl.Format()
    assert not f.webp_accepted() and not f.avif_accepted()
    f.accept_webp()
    assert f.webp_accepted() and not f.avif_accepted()  # content.rs:54-65
    f.accept_avif()
    assert f.webp_accepted() and f.avif_accepted()


def test_output_format_and_front_end_choice(fl):
    # handler.rs:256-261: webp wins over avif, each only if asked AND accepted; otherwise the input format stays
    both = fl.Format(fl.ACCEPT_WEBP | fl.ACCEPT_AVIF)
    q = fl.Query.parse("w=300&h=200&webp=true&avif=true")
    p, fmt = q.to_params(both, input_is_jpeg=True)
    assert fmt == fl.OUT_WEBP and p.front_end == fl.FE_WEBP420
    p, fmt = q.to_params(fl.Format(fl.ACCEPT_AVIF), input_is_jpeg=True)
    assert fmt == fl.OUT_AVIF and p.front_end == fl.FE_NONE
    p, fmt = q.to_params(fl.Format(), input_is_jpeg=True)
    assert fmt == fl.OUT_KEEP and p.front_end == fl.FE_JFIF444
    p, fmt = q.to_params(fl.Format(), input_is_jpeg=False)
    assert fmt == fl.OUT_KEEP and p.front_end == fl.FE_NONE
    # quality == 100 selects lossless WebP (handler.rs:288-292): pixels, not YUV planes
    p, fmt = fl.Query.parse("webp=true&quality=100").to_params(both)
    assert fmt == fl.OUT_WEBP and p.front_end == fl.FE_NONE
    p, _ = fl.Query.parse("w=10&h=20&rgb=9,8,7&crop=true&blur=12&grayscale=true&inverse=true&quality=5").to_params(both)
    assert (p.has_dims, p.w, p.h, p.fill_r, p.fill_g, p.fill_b, p.crop, p.blur_sigma, p.grayscale, p.inverse, p.quality) == \
           (1, 10, 20, 9, 8, 7, 1, 12.0, 1, 1, 5)
