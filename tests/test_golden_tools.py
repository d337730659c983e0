"""tools/compare_golden.py (the two ends of the Rust golden dump, tools/golden_dump.rs): export + compare work end to end.
The crate itself cannot run here (no Rust toolchain), so the oracle's own files stand in for its output."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "compare_golden.py")


def test_export_then_compare(tmp_path):
    d = str(tmp_path / "cases")
    subprocess.run([sys.executable, TOOL, "export", d], check=True)
    lines = [l.split() for l in open(os.path.join(d, "cases.txt")) if l.strip()]
    assert len(lines) >= 50
    for f in lines:
        name = f[0]
        if any(k.startswith("jpeg=") for k in f[4:]):
            shutil.copy(os.path.join(d, name + ".oracle.jpg"), os.path.join(d, name + ".crate.jpg"))
        else:
            shutil.copy(os.path.join(d, name + ".oracle.raw"), os.path.join(d, name + ".crate.raw"))
            shutil.copy(os.path.join(d, name + ".oracle.shape"), os.path.join(d, name + ".crate.shape"))
    assert subprocess.run([sys.executable, TOOL, "compare", d]).returncode == 0
    # a 2-LSB error in one resampled picture, and one flipped byte in a stream, are both caught
    p = os.path.join(d, "rgb_96x128_to_30x20.crate.raw")
    b = bytearray(open(p, "rb").read())
    b[10] = (b[10] + 2) % 256 if b[10] < 250 else b[10] - 2
    open(p, "wb").write(bytes(b))
    assert subprocess.run([sys.executable, TOOL, "compare", d], capture_output=True).returncode == 1


def test_golden_dump_source_covers_every_operation_of_process_image():
    src = open(os.path.join(ROOT, "tools", "golden_dump.rs")).read()
    for call in ("apply_orientation", ".grayscale()", ".invert()", "resize_to_fill", ".resize(", "from_pixel", "overlay", ".blur(",
                 "JpegEncoder::new_with_quality"):
        assert call in src, call


import pytest


@pytest.mark.gpu
def test_device_mode_measures_the_shipped_kernels(tmp_path):
    """`compare_golden.py device DIR`: the kernels that ship (the matrix-pipe kernel in its default full-width arithmetic) against
    the exported oracle files -- and against the crate's, wherever a cargo environment has put them next to these."""
    d = str(tmp_path / "cases")
    subprocess.run([sys.executable, TOOL, "export", d], check=True)
    r = subprocess.run([sys.executable, TOOL, "device", d], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "config1_1080p_uniform_to_300x200" in r.stdout and "(matrix-pipe kernel)" in r.stdout
    assert "config2_1080p_gray_blur10" in r.stdout and os.path.exists(os.path.join(d, "mfma_4k_to_300x200.device.raw"))
