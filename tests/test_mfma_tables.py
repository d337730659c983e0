"""Host logic of the matrix-pipe resample kernel (csrc/fl_mfma_tables.cpp), checked WITHOUT a device: flgpu_debug_mfma_plan
builds the tables for a geometry and decodes them again the way the kernel reads them -- K-block by K-block with the
accumulator sets moving as tiles complete, operand by operand through the tile tables -- and compares the result with the
plain per-axis weight tables (image 0.25.6 imageops/sample.rs index and weight maths, tests/test_tables.py):
  * every vertical tap lands in the right output row exactly once -- full-width arithmetic (the default): as three f16 terms whose
    sum IS the f32 weight (exact for every |w| >= 2^-16, to 2^-39 below); packed arithmetic (rounds 2-3, FLGPU_MFMA_ARITH=packed):
    as two f16 terms that give it back to 2^-24;
  * every horizontal tap lands on the right (output, source byte) pair exactly once, channel structure included, as a
    fixed-point weight (full: three byte digits, 2^-24 steps; packed: two digits, 2^-14..2^-17) within one rounding of the f32
    weight, and each output's weights sum to exactly 1;
  * nothing else is non-zero (rows and columns outside a window, outputs outside a strip, lanes of unused tile slots)."""
import numpy as np
import pytest

GEOMETRIES = [
    # sw, sh, channels, rw, rh, crop (cx, cy, cw, ch) or None
    (1920, 1080, 3, 300, 169, None),                # BASELINE config 1
    (1920, 1080, 3, 356, 200, (28, 0, 300, 200)),   # crop=true: resize_to_fill + centre crop
    (1920, 1080, 3, 317, 178, None),                # short last tile (2 rows): the extra all-zero pass
    (1920, 1080, 3, 320, 180, None),
    (1920, 1080, 3, 400, 225, None),
    (1920, 1080, 3, 480, 270, None),
    (1920, 1080, 3, 150, 84, None),
    (3840, 2160, 3, 300, 169, None),                # 4K: 68 K-blocks, 7 strips
    (3840, 2160, 3, 640, 360, None),
    (1280, 720, 3, 160, 90, None),
    (1936, 1088, 3, 395, 222, (31, 0, 333, 222)),
    (800, 600, 3, 100, 75, None),
    (1920, 1080, 4, 300, 169, None),                # Rgba8
    (1920, 1080, 1, 300, 169, None),                # Luma8
    (1600, 1200, 2, 250, 188, None),                # LumaA8
    (6000, 4000, 3, 300, 200, None),                # ratio 20
    (1024, 333, 3, 90, 29, None),                   # two tiles
    (1920, 1080, 3, 256, 144, None),                # three UNEQUAL strips (the inner one has two halos)
    (1920, 1080, 3, 640, 360, None),                # wide layout: 214 pixels per strip
    (1920, 1080, 3, 512, 288, None),
    (1920, 1080, 4, 600, 338, None),                # wide layout, Rgba8
]


@pytest.mark.parametrize("packed", [False, True], ids=["full", "packed"])
@pytest.mark.parametrize("sw,sh,c,rw,rh,crop", GEOMETRIES)
def test_tables_decode_back_to_the_axis_weights(fl, sw, sh, c, rw, rh, crop, packed):
    d = fl.debug_mfma_plan(sw, sh, c, rw, rh, crop, packed=packed)
    assert d is not None, "this geometry is meant to fit the kernel"
    assert d["bad_vertical"] == 0 and d["bad_horizontal"] == 0, d
    if packed:
        assert d["vertical_weight_error"] < 2.0 ** -24, d       # two f16 terms carry 22 bits of a weight < 1/2
        assert 14 <= d["hs"] <= 17, d
    else:
        assert d["vertical_weight_error"] < 2.0 ** -39, d       # three f16 terms ARE the f32 weight (f16 subnormals end at 2^-24 / 2^15)
        assert d["hs"] == 24, d
    # fixed point with 2^-hs steps; the largest tap of an output absorbs the rounding of the others (so the sum is exactly 1)
    assert d["horizontal_weight_error"] < 64 * 2.0 ** -(d["hs"] + 1), d
    rows = crop[3] if crop else rh
    assert d["tiles"] == (rows + 15) // 16 and d["k_blocks"] == (sh + 31) // 32 and d["strips"] >= 1


def test_geometries_the_kernel_refuses(fl):
    assert fl.debug_mfma_plan(1920, 1080, 3, 1200, 675) is None       # ratio 1.6: more than two tiles alive per K-block
    assert fl.debug_mfma_plan(320, 200, 3, 640, 400) is None          # up-scale
    assert fl.debug_mfma_plan(1920, 1080, 3, 300, 169, (0, 0, 301, 169)) is None   # crop outside the resized picture
    assert fl.debug_mfma_plan(1920, 1080, 5, 300, 169) is None        # not a channel count


@pytest.mark.parametrize("packed", [False, True], ids=["full", "packed"])
def test_strips_are_as_few_as_the_source_bytes_allow(fl, packed):
    """A strip is one workgroup's walk over all rows, whatever its width: equal strips needed a fourth one for 256 columns (the
    inner strips carry two halos, the outer ones one), and 408 outputs per strip cut 480 / 512 / 640 columns into 4 / 4 / 5."""
    for rw, rh, strips in ((300, 169, 3), (256, 144, 3), (200, 113, 3), (480, 270, 3), (512, 288, 3), (640, 360, 3), (150, 84, 4)):
        assert fl.debug_mfma_plan(1920, 1080, 3, rw, rh, packed=packed)["strips"] == strips, (rw, rh)
    assert fl.debug_mfma_plan(1920, 1080, 1, 640, 360, packed=packed)["strips"] == 2     # (1- and 2-channel sources keep the narrow layout)


def test_short_last_tile_is_flagged(fl):
    assert fl.debug_mfma_plan(1920, 1080, 3, 317, 178)["tail"] == 1   # rows 176-177 end with the tile before them
    assert fl.debug_mfma_plan(1920, 1080, 3, 300, 169)["tail"] == 0


@pytest.mark.parametrize("pictures,strips,workgroups", [(1024, 3, 256), (1025, 3, 256), (1100, 3, 256), (100, 3, 256), (1000, 4, 256), (512, 2, 256),
                                                       (7, 3, 256), (300, 3, 304), (90, 2, 64)])
def test_persistent_workgroups_get_every_item_once_and_keep_their_strip(fl, pictures, strips, workgroups):
    """Round 5: the full-width kernel's workgroups are persistent and walk lists of items (csrc/fl_batch.cpp assign_items).  For a uniform
    launch every (picture, strip, tile) is covered exactly once; a workgroup's whole-picture items all have ONE strip (so that it goes
    from picture to picture without a new set-up); the S strips of a picture run in the same position of S workgroups' lists (same
    time), most of them on one XCD; the pictures left over after the whole rounds are cut into row bands, and no workgroup ends up
    with more than one band's worth of work above the lightest."""
    tiles = 11
    job, strip, t0, t1, lists = fl.debug_assign_items(pictures, strips, tiles, workgroups)
    seen = set()
    for j, s, a, b in zip(job, strip, t0, t1):
        for t in range(a, b):
            assert (j, s, t) not in seen
            seen.add((j, s, t))
    assert len(seen) == pictures * strips * tiles
    assert lists[:, 1].sum() == len(job) and (lists[1:, 0] == lists[:-1, 0] + lists[:-1, 1]).all() and lists[0, 0] == 0
    whole = (t1 - t0) == tiles
    where = {}                                            # (picture, strip) of a whole item -> (workgroup, position in its list)
    for b, (first, cnt) in enumerate(lists):
        ks = np.arange(first, first + cnt)
        assert len(set(strip[ks][whole[ks]])) <= 1, "a workgroup changes strips between whole pictures"
        for pos, k in enumerate(ks):
            if whole[k]:
                where[(job[k], strip[k])] = (b, pos)
    if pictures * strips >= 3 * workgroups:              # a launch with whole rounds
        assert whole.sum() >= (pictures * strips) * 0.75
        same_xcd = 0
        pics = {j for j, _ in where}
        for j in pics:
            at = [where[(j, s)] for s in range(strips)]
            assert len({pos for _, pos in at}) == 1, "the strips of a picture run in different rounds"
            same_xcd += len({b % 8 for b, _ in at}) == 1
        assert same_xcd >= 0.9 * len(pics)
    # balance, in K-blocks of the debug plan (a tile ends every third K-block and needs five; an item's set-up counts like two)
    load = np.array([sum(3 * (t1[k] - t0[k]) + 2 + 2 for k in range(f, f + c)) for f, c in lists])
    if len(job) >= workgroups:
        assert load.max() - load.min() <= 3 * tiles + 2 + 12, (load.min(), load.max())
