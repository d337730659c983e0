#!/usr/bin/env python3
"""Generates tests/golden/oracle_ref.npz: outputs of the CPU oracle (reference arithmetic) on small seeded
inputs, committed so that any later change of the oracle's arithmetic shows up as a diff.  These are NOT
reference outputs (the reference cannot be run here); they freeze the restatement."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/tools/ -> repository root
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import synth  # noqa: E402

CASES = {
    # name: (h, w, c, dist, request kwargs)
    "rgb_96x128_to_30x20": (96, 128, 3, "photo", dict(w=30, h=20)),
    "rgb_96x128_to_30x20_crop": (96, 128, 3, "uniform", dict(w=30, h=20, crop=True)),
    "rgb_90x60_to_40x40_fill": (90, 60, 3, "uniform", dict(w=40, h=40, fill=(1, 2, 3))),
    "rgba_64x64_to_21x33": (64, 64, 4, "uniform", dict(w=21, h=33)),
    "rgb_up_24x32_to_50x40": (24, 32, 3, "photo", dict(w=50, h=40)),
    "gray_blur": (48, 64, 3, "photo", dict(w=32, h=32, grayscale=True, blur_sigma=10.0)),
    "inverse_only": (16, 16, 4, "uniform", dict(inverse=True)),
    "blur20_only": (40, 40, 1, "uniform", dict(blur_sigma=20.0)),
}


def main():
    o = oracle_lib.load()
    out = {}
    for i, (name, (h, w, c, dist, kw)) in enumerate(CASES.items()):
        img = getattr(synth, dist)(h, w, c, index=700 + i)
        out[name + "__in"] = img
        out[name + "__ref"] = o.process_pixels(img, arith=oracle_lib.ARITH_REF, **kw)
        out[name + "__fma"] = o.process_pixels(img, arith=oracle_lib.ARITH_FMA, **kw)
    dst = os.path.join(ROOT, "tests", "golden", "oracle_ref.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
