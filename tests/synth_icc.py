"""A small synthetic CMYK printer profile (ICC v2, one A2B0 lut16 tag), built from scratch so that tests of
the CMYK -> sRGB path need no profile file: liblcms2 and flgpu_set_cmyk_profile are both fed these bytes.
The colour model is deliberately simple (subtractive mix with dot gain, black that is not neutral); what
matters is that it is non-linear in all four inks so that every interpolation branch is exercised."""
import struct

import numpy as np


def _s15f16(v):
    return struct.pack(">i", int(round(v * 65536.0)))


def _lab_from_cmyk(c, m, y, k):
    # inks in 0..1 -> a made-up device model -> linear RGB -> XYZ (D50) -> Lab
    gain = lambda t: t ** 0.85
    c, m, y, k = gain(c), gain(m), gain(y), gain(k)
    r = (1 - 0.95 * c) * (1 - 0.10 * m) * (1 - 0.03 * y) * (1 - 0.92 * k)
    g = (1 - 0.25 * c) * (1 - 0.90 * m) * (1 - 0.08 * y) * (1 - 0.93 * k)
    b = (1 - 0.08 * c) * (1 - 0.35 * m) * (1 - 0.93 * y) * (1 - 0.90 * k)
    M = np.array([[0.4360747, 0.3850649, 0.1430804], [0.2225045, 0.7168786, 0.0606169], [0.0139322, 0.0971045, 0.7141733]])
    xyz = np.tensordot(np.stack([r, g, b], -1), M.T, 1) * 0.92 + 0.004
    wp = np.array([0.9642, 1.0, 0.8249])
    t = xyz / wp
    f = np.where(t > (6 / 29) ** 3, np.cbrt(t), t / (3 * (6 / 29) ** 2) + 4 / 29)
    L = 116 * f[..., 1] - 16
    a = 500 * (f[..., 0] - f[..., 1])
    bb = 200 * (f[..., 1] - f[..., 2])
    return L, a, bb


def cmyk_profile(grid=9, seed=0):
    """Returns the bytes of an ICC profile: class 'prtr', colour space 'CMYK', PCS Lab, A2B0 = lut16."""
    rng = np.random.default_rng(seed)
    ax = np.linspace(0.0, 1.0, grid)
    c, m, y, k = np.meshgrid(ax, ax, ax, ax, indexing="ij")
    L, a, b = _lab_from_cmyk(c, m, y, k)
    L = np.clip(L + rng.normal(0, 0.3, L.shape), 0, 100)
    # ICC v2 16-bit Lab encoding: L 0..100 -> 0..0xFF00, a/b -128..127.996 -> 0..0xFFFF with 0 at 0x8000
    clut = np.stack([np.clip(np.round(L * 652.80), 0, 65535), np.clip(np.round((a + 128.0) * 256.0), 0, 65535),
                     np.clip(np.round((b + 128.0) * 256.0), 0, 65535)], -1).astype(">u2")
    n_in = 256
    t = np.linspace(0.0, 1.0, n_in)
    in_tables = [np.round((t ** g) * 65535.0).astype(">u2") for g in (1.0, 1.1, 0.9, 1.25)]
    out_tables = [np.array([0, 65535], ">u2")] * 3
    lut = b"mft2" + b"\0" * 4 + bytes([4, 3, grid, 0])
    lut += b"".join(_s15f16(v) for v in (1, 0, 0, 0, 1, 0, 0, 0, 1))
    lut += struct.pack(">HH", n_in, 2)
    lut += b"".join(x.tobytes() for x in in_tables) + clut.tobytes() + b"".join(x.tobytes() for x in out_tables)

    def text_desc(s):
        raw = s.encode("ascii") + b"\0"
        return b"desc" + b"\0" * 4 + struct.pack(">I", len(raw)) + raw + b"\0" * (4 + 4 + 2 + 1 + 67)

    tags = [(b"desc", text_desc("fanlin test CMYK")), (b"cprt", b"text" + b"\0" * 4 + b"public domain\0"),
            (b"wtpt", b"XYZ " + b"\0" * 4 + _s15f16(0.9642) + _s15f16(1.0) + _s15f16(0.8249)), (b"A2B0", lut)]
    offset = 128 + 4 + 12 * len(tags)
    directory, body = struct.pack(">I", len(tags)), b""
    for sig, data in tags:
        data += b"\0" * (-len(data) % 4)
        directory += sig + struct.pack(">II", offset + len(body), len(data))
        body += data
    size = offset + len(body)
    header = struct.pack(">I4sI4s4s4s", size, b"\0\0\0\0", 0x02400000, b"prtr", b"CMYK", b"Lab ")
    header += struct.pack(">6H", 2024, 1, 1, 0, 0, 0) + b"acsp" + b"\0" * 4 + struct.pack(">I", 0) + b"\0" * 8 + b"\0" * 8
    header += struct.pack(">I", 0) + _s15f16(0.9642) + _s15f16(1.0) + _s15f16(0.8249) + b"\0" * 4 + b"\0" * 16 + b"\0" * 28
    assert len(header) == 128
    return header + directory + body
