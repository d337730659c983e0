"""The C-ABI shared library loads on a machine without a GPU, exports every symbol include/fanlin_gpu.h
declares, agrees with the ctypes mirrors on struct layout, and refuses to run without a device (no CPU
fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fanlin_gpu.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(flgpu_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(fl):
    lib = fl.load_library()
    names = declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(fl.EXPORTED_SYMBOLS) == names, "python binding list is out of date with the header"
    assert lib.flgpu_abi_version() == 6   # 3: flgpu_plan::max_out_bytes, flgpu_config::n_devices / devices; 4: flgpu_stats::jpeg_device_huffman[_retries]; 5: flgpu_stats::wtile_launches; 6: flgpu_debug_set / flgpu_debug_get (no getenv after flgpu_create)


def test_struct_layouts_match_the_header(fl):
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "fanlin_gpu.h"
int main(void) {
  printf("flgpu_image %zu %zu %zu\n", sizeof(flgpu_image), offsetof(flgpu_image, width), offsetof(flgpu_image, flags));
  printf("flgpu_query %zu %zu %zu\n", sizeof(flgpu_query), offsetof(flgpu_query, w), offsetof(flgpu_query, rgb));
  printf("flgpu_params %zu %zu %zu\n", sizeof(flgpu_params), offsetof(flgpu_params, blur_sigma), offsetof(flgpu_params, front_end));
  printf("flgpu_plan %zu %zu %zu\n", sizeof(flgpu_plan), offsetof(flgpu_plan, out_w), offsetof(flgpu_plan, max_out_bytes));
  printf("flgpu_config %zu %zu %zu\n", sizeof(flgpu_config), offsetof(flgpu_config, profile), offsetof(flgpu_config, devices));
  printf("flgpu_stats %zu %zu %zu\n", sizeof(flgpu_stats), offsetof(flgpu_stats, resample_ms), offsetof(flgpu_stats, frontend_ms));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)  # header must be plain C
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    got = {l.split()[0]: tuple(int(x) for x in l.split()[1:]) for l in out.strip().splitlines()}
    S = fl
    want = {
        "flgpu_image": (C.sizeof(S.flgpu_image), S.flgpu_image.width.offset, S.flgpu_image.flags.offset),
        "flgpu_query": (C.sizeof(S.flgpu_query), S.flgpu_query.w.offset, S.flgpu_query.rgb.offset),
        "flgpu_params": (C.sizeof(S.flgpu_params), S.flgpu_params.blur_sigma.offset, S.flgpu_params.front_end.offset),
        "flgpu_plan": (C.sizeof(S.flgpu_plan), S.flgpu_plan.out_w.offset, S.flgpu_plan.max_out_bytes.offset),
        "flgpu_config": (C.sizeof(S.flgpu_config), S.flgpu_config.profile.offset, S.flgpu_config.devices.offset),
        "flgpu_stats": (C.sizeof(S.flgpu_stats), S.flgpu_stats.resample_ms.offset, S.flgpu_stats.frontend_ms.offset),
    }
    assert got == want


def test_no_cpu_fallback_without_a_device(fl):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible")
    with pytest.raises(fl.FanlinError) as e:
        fl.State()
    assert e.value.status == 3  # FLGPU_ERR_NO_DEVICE


def test_product_library_does_not_link_the_oracle(fl):
    out = subprocess.run(["ldd", fl.LIB_PATH], capture_output=True, text=True).stdout
    assert "fanlin_oracle" not in out
    syms = subprocess.run(["nm", "-D", fl.LIB_PATH], capture_output=True, text=True).stdout
    assert " fo_" not in syms


def test_error_strings(fl):
    lib = fl.load_library()
    seen = set()
    for code in range(0, 9):
        s = lib.flgpu_strerror(code).decode()
        assert s and s != "unknown status"
        seen.add(s)
    assert len(seen) == 9 and lib.flgpu_strerror(99).decode() == "unknown status"


def test_rust_shim_mirrors_the_structs(fl):
    # shim/handler_gpu.rs is the reference-side binding as Rust source; its #[repr(C)] structs must list the header's
    # fields in the header's order (the ctypes mirrors are already checked against the header above)
    import re
    text = open(os.path.join(ROOT, "shim", "handler_gpu.rs")).read()
    assert "flgpu_abi_version() }, 6" in text
    mirrors = {"FlImage": fl.flgpu_image, "FlParams": fl.flgpu_params, "FlPlan": fl.flgpu_plan, "FlConfig": fl.flgpu_config, "FlJpegInfo": fl.flgpu_jpeg_info}
    declared = set(re.findall(r"#\[repr\(C\)\](?:\s*#\[derive\([^)]*\)\])?\s*pub struct (\w+)", text))
    assert declared == set(mirrors), declared                      # every #[repr(C)] struct of the shim is checked
    for rust, cls in mirrors.items():
        m = re.search(r"pub struct %s \{(.*?)\}" % rust, text, re.S)
        assert m, rust
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        names = [f.split(":")[0].strip() for f in body.split(",") if ":" in f]
        assert names == [n for n, _ in cls._fields_], (rust, names)
    # every entry point the shim binds exists in the header (and so in the library: test_library_exports_every_declared_symbol)
    header = open(os.path.join(ROOT, "include", "fanlin_gpu.h")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', text, re.S).group(1)
    bound = re.findall(r"fn (flgpu_\w+)\(", block)
    assert {"flgpu_transform", "flgpu_process_jpeg", "flgpu_process_jpeg_plan", "flgpu_jpeg_info_of", "flgpu_transform_batch", "flgpu_cmyk_to_rgb"} <= set(bound)
    for name in bound:
        assert re.search(r"\b%s\(" % name, header), name
    # constants the shim restates
    for rust_name, c_name in (("IMG_JPEG_SOURCE", "FLGPU_IMG_JPEG_SOURCE"), ("ACCEPT_WEBP", "FLGPU_ACCEPT_WEBP"), ("ACCEPT_AVIF", "FLGPU_ACCEPT_AVIF")):
        rv = int(re.search(r"const %s: u32 = (\d+);" % rust_name, text).group(1))
        cv = int(re.search(r"#define %s\s+(\d+)u" % c_name, header).group(1))
        assert rv == cv, rust_name
    assert int(re.search(r"const ERR_UNSUPPORTED: c_int = (\d+);", text).group(1)) == fl.ERR_UNSUPPORTED
    kinds = dict(re.findall(r"FLGPU_(RESULT_\w+) = (\d+)", header))
    for k, v in kinds.items():
        assert int(re.search(r"pub const %s: c_int = (\d+);" % k, text).group(1)) == int(v), k


def test_loaded_library_was_built_from_the_sources_beside_it(fl):
    # build provenance: the hash of the sources is compiled into the library (and written to build_info.json); a stale
    # binary -- built from other sources than the ones in the tree -- is caught here, on the CPU box and on the GPU box
    import json
    info = fl.build_info()
    assert info.startswith("sources ") and "gfx950" in info
    assert info.split()[1].rstrip(";") == fl.source_hash(), "libfanlin_gpu.so is stale: rebuild with __graft_entry__.build()"
    rec = json.load(open(os.path.join(ROOT, "fanlin-rs_amd", "build_info.json")))
    assert rec["sources_sha256_16"] == fl.source_hash()
