#!/usr/bin/env python3
"""Generates tests/golden/cmyk_lcms2.npz: golden vectors for the CMYK -> sRGB path produced by the REAL
Little CMS 2 (system liblcms2, the C library behind the reference's `lcms2` crate) on the reference's own data
file profiles/default.icc, with the transform of reference src/handler.rs:469-488 (CMYK_8 -> RGB_8, Perceptual,
NO_CACHE).  Stored: the 17^4 x 3 device-link nodes (what Little CMS precomputes for that transform), seeded
CMYK pixels and the library's RGB answers for them.  Run in the build container only (needs /root/reference);
the tests read just the .npz.

    python tools/gen_cmyk_golden.py [/root/reference/profiles/default.icc]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import lcms2_lib  # noqa: E402


def pixels(seed=0xC3B1):
    rng = np.random.default_rng(seed)
    px = rng.integers(0, 256, (32768, 4), dtype=np.uint8)
    px[:4096] = rng.choice(np.array([0, 1, 15, 16, 17, 127, 128, 239, 240, 254, 255], np.uint8), (4096, 4))  # node boundaries
    ramp = np.arange(256, dtype=np.uint8)
    for k in range(4):                                                                  # single-ink ramps
        px[4096 + 256 * k:4096 + 256 * (k + 1)] = 0
        px[4096 + 256 * k:4096 + 256 * (k + 1), k] = ramp
    px[5120:5376] = ramp[:, None]                                                       # all inks together
    return px


def main():
    icc_path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/profiles/default.icc"
    icc = open(icc_path, "rb").read()
    t = lcms2_lib.Cmyk2Rgb(icc)
    px = pixels()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cmyk_lcms2.npz")
    np.savez_compressed(out, nodes=t.device_link_nodes(), cmyk=px, rgb=t.convert(px),
                        lcms_version=np.int32(lcms2_lib.version()), icc_bytes=np.int64(len(icc)))
    print("wrote", out, os.path.getsize(out), "bytes; lcms", lcms2_lib.version())


if __name__ == "__main__":
    main()
