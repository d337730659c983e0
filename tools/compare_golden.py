#!/usr/bin/env python3
"""tools/compare_golden.py -- the two ends of the Rust golden dump (tools/golden_dump.rs).

  export DIR    writes every case of tests/golden/oracle_ref.npz (+ the extra crate-only cases below) as raw inputs,
                DIR/cases.txt and the oracle's outputs (pixels, and JPEG streams at quality 75 and 85)
  compare DIR   after `cargo run` in tools/golden_dump: diffs NAME.crate.raw / NAME.crate.jpg against the oracle's
                files; exit code 0 only if every resample / blur case is within 1 LSB, every exact case and every
                JPEG stream byte-identical
  device DIR    (needs a GPU) runs every pixel case through libfanlin_gpu.so -- the kernels that actually ship, the
                matrix-pipe kernel in its default full-width arithmetic -- writes NAME.device.raw and reports the distance to
                the oracle and, where NAME.crate.raw exists, to the crate itself (max |diff| and differing bytes)

`export` needs the oracle library (oracle/libfanlin_oracle.so); `compare` needs only numpy.  The case list is
tests/tools/gen_oracle_golden.py::CASES plus cases for the operations that table does not hold (orientation, Luma /
LumaA sources, letterbox-only, blur-only on colour, invert + resize)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

EXTRA = {
    # name: (h, w, c, dist, request kwargs)
    "luma_80x100_to_25x25": (80, 100, 1, "photo", dict(w=25, h=25)),
    "lumaa_70x50_to_30x30": (70, 50, 2, "uniform", dict(w=30, h=30)),
    "rgba_translucent_letterbox": (40, 30, 4, "uniform", dict(w=64, h=40)),
    "letterbox_only_rgb": (20, 30, 3, "uniform", dict(w=30, h=40, fill=(200, 100, 50))),
    "invert_resize_rgb": (120, 90, 3, "photo", dict(w=45, h=45, inverse=True)),
    "blur10_rgb": (48, 40, 3, "photo", dict(blur_sigma=10.0)),
    "gray_only_rgba": (24, 24, 4, "uniform", dict(grayscale=True)),
    "orient6_resize": (60, 90, 3, "photo", dict(w=30, h=30, orientation=6)),
    "orient3_only": (17, 23, 3, "uniform", dict(orientation=3)),
    "lenna_like_512_to_300x200": (512, 512, 3, "photo", dict(w=300, h=200)),
    # the shapes the matrix-pipe kernel serves (fl_mfma.hip; since round 4 in full-width arithmetic: u8 x the f32 weight as three
    # f16 terms -> f32, a 23-bit intermediate x 24-bit weights -> exact i32): the first environment with cargo measures THAT
    # kernel's distance to the crate too (`device DIR` below), not only the oracle's
    "config1_1080p_uniform_to_300x200": (1080, 1920, 3, "uniform", dict(w=300, h=200)),
    "config1_1080p_photo_to_300x200": (1080, 1920, 3, "photo", dict(w=300, h=200)),
    "config1_1080p_photo_crop": (1080, 1920, 3, "photo", dict(w=300, h=200, crop=True)),
    "rgba_1080p_to_300x169": (1080, 1920, 4, "uniform", dict(w=300, h=169)),
    # round 3's planner: three unequal strips, and the wide LDS layout for ratios below ~4.7
    "mfma_1080p_to_256x144": (1080, 1920, 3, "photo", dict(w=256, h=144)),
    "mfma_1080p_to_640x360": (1080, 1920, 3, "photo", dict(w=640, h=360)),
    # round 4: a 4K source (68 K-blocks, 7 strips), a Luma8 source (dense one-channel form), LumaA8, and BASELINE config 2's request
    "mfma_4k_to_300x200": (2160, 3840, 3, "photo", dict(w=300, h=200)),
    "mfma_luma_1080p_to_300x200": (1080, 1920, 1, "photo", dict(w=300, h=200)),
    "mfma_lumaa_1080p_to_300x169": (1080, 1920, 2, "uniform", dict(w=300, h=169)),
    "config2_1080p_gray_blur10": (1080, 1920, 3, "uniform", dict(w=300, h=200, grayscale=True, blur_sigma=10.0)),
    # round 4: the window-tile matrix-pipe kernel (fl_wtile.hip: ratios 1.25 .. 3.1 and blurs at the full-width arithmetic): a mild
    # down-scale (this case went through the f32 tile kernel before), ratio 3 with odd rows, a colour blur of a large picture, and a
    # blur behind the flagship resample
    "tile_720p_to_800x450": (720, 1280, 3, "photo", dict(w=800, h=450)),
    "wtile_1080p_to_1000x562": (1080, 1920, 3, "photo", dict(w=1000, h=562)),
    "wtile_odd_pitch_ratio3": (540, 961, 3, "uniform", dict(w=320, h=180)),
    "wtile_blur20_800x600": (600, 800, 3, "photo", dict(blur_sigma=20.0)),
    "wtile_1080p_to_300x200_blur8": (1080, 1920, 3, "photo", dict(w=300, h=200, blur_sigma=8.0)),
    # the tiled two-pass kernel (up-scales), whose bytes must be the crate's up to the order of the f32 sums
    "tile_thumbnail_upscale": (120, 160, 3, "photo", dict(w=300, h=200)),
}
EXACT = {"inverse_only", "letterbox_only_rgb", "gray_only_rgba", "orient3_only"}  # no f32 resampling involved
JPEG_QUALITIES = (75, 85)


def all_cases():
    import gen_oracle_golden
    cases = dict(gen_oracle_golden.CASES)
    cases.update(EXTRA)
    return cases


def case_line(name, h, w, c, kw, quality=None):
    parts = [name, str(h), str(w), str(c)]
    for k, v in kw.items():
        if k == "fill":
            parts.append("fill=%d,%d,%d" % tuple(v))
        elif k == "blur_sigma":
            parts.append("blur=%g" % v)
        elif isinstance(v, bool):
            parts.append("%s=%s" % (k, "true" if v else "false"))
        else:
            parts.append("%s=%s" % (k, v))
    if quality:
        parts.append("jpeg=%d" % quality)
    return " ".join(parts)


def export(d):
    import oracle_lib
    import synth
    os.makedirs(d, exist_ok=True)
    o = oracle_lib.load()
    lines = []
    for i, (name, (h, w, c, dist, kw)) in enumerate(all_cases().items()):
        img = getattr(synth, dist)(h, w, c, index=700 + i)
        img.tofile(os.path.join(d, name + ".in.raw"))
        ref = o.process_pixels(img, arith=oracle_lib.ARITH_REF, **kw)
        ref.tofile(os.path.join(d, name + ".oracle.raw"))
        open(os.path.join(d, name + ".oracle.shape"), "w").write("%d %d %d\n" % (ref.shape[0], ref.shape[1], ref.shape[2] if ref.ndim == 3 else 1))
        lines.append(case_line(name, h, w, c, kw))
        for q in JPEG_QUALITIES:
            jn = "%s_q%d" % (name, q)
            img.tofile(os.path.join(d, jn + ".in.raw"))
            open(os.path.join(d, jn + ".oracle.jpg"), "wb").write(o.jpeg_encode(ref, q))
            lines.append(case_line(jn, h, w, c, kw, quality=q))
    open(os.path.join(d, "cases.txt"), "w").write("\n".join(lines) + "\n")
    print("wrote %d cases to %s" % (len(lines), d))


def compare(d):
    bad = 0
    for line in open(os.path.join(d, "cases.txt")):
        f = line.split()
        if not f or f[0].startswith("#"):
            continue
        name = f[0]
        jpeg = any(k.startswith("jpeg=") for k in f[4:])
        if jpeg:
            a = open(os.path.join(d, name + ".oracle.jpg"), "rb").read()
            try:
                b = open(os.path.join(d, name + ".crate.jpg"), "rb").read()
            except OSError:
                print("%-40s MISSING crate output" % name); bad += 1; continue
            same = a == b
            print("%-40s jpeg  oracle %6d B  crate %6d B  %s" % (name, len(a), len(b), "identical" if same else "DIFFERENT"))
            bad += 0 if same else 1
            continue
        try:
            got = np.fromfile(os.path.join(d, name + ".crate.raw"), np.uint8)
        except OSError:
            print("%-40s MISSING crate output" % name); bad += 1; continue
        want = np.fromfile(os.path.join(d, name + ".oracle.raw"), np.uint8)
        gs = open(os.path.join(d, name + ".crate.shape")).read().split()
        ws = open(os.path.join(d, name + ".oracle.shape")).read().split()
        if gs != ws or got.size != want.size:
            print("%-40s SHAPE crate %s oracle %s" % (name, gs, ws)); bad += 1; continue
        diff = np.abs(got.astype(int) - want.astype(int))
        mx, frac = int(diff.max(initial=0)), float((diff != 0).mean()) if diff.size else 0.0
        base = name.rsplit("_q", 1)[0]
        limit = 0 if base in EXACT else 1
        ok = mx <= limit
        print("%-40s pixels  max |diff| %d (limit %d)  differing %.4f%%  %s" % (name, mx, limit, 100 * frac, "ok" if ok else "FAIL"))
        bad += 0 if ok else 1
    print("%d case(s) outside their bar" % bad)
    return 1 if bad else 0


def device(d):
    import importlib
    import synth
    sys.path.insert(0, ROOT)
    fl = importlib.import_module("fanlin-rs_amd")
    bad = 0
    with fl.State() as st:
        for i, (name, (h, w, c, dist, kw)) in enumerate(all_cases().items()):
            img = getattr(synth, dist)(h, w, c, index=700 + i)
            before = st.stats()["mfma_launches"]
            got = st.process_pixels(img, fl.make_params(**kw))
            used = st.stats()["mfma_launches"] > before
            got.tofile(os.path.join(d, name + ".device.raw"))
            line = "%-40s device%s" % (name, " (matrix-pipe kernel)" if used else "")
            for other in ("oracle", "crate"):
                try:
                    want = np.fromfile(os.path.join(d, name + "." + other + ".raw"), np.uint8)
                except OSError:
                    continue
                if want.size != got.size:
                    line += "  vs %s: SHAPE" % other; bad += 1; continue
                diff = np.abs(got.reshape(-1).astype(int) - want.astype(int))
                mx = int(diff.max(initial=0))
                line += "  vs %s: max |diff| %d, differing %.4f%%" % (other, mx, 100 * float((diff != 0).mean()))
                bad += 0 if mx <= (0 if name in EXACT else 1) else 1
            print(line)
    print("%d case(s) outside their bar" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) != 3 or sys.argv[1] not in ("export", "compare", "device"):
        sys.exit(__doc__)
    sys.exit({"export": export, "compare": compare, "device": device}[sys.argv[1]](sys.argv[2]))
