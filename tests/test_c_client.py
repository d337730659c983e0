"""The C ABI used from plain C -- no Python, no PyTorch, no second HIP runtime in the process (the situation of the
Rust shim in INTEGRATION.md).  tests/c_client.c is compiled with gcc against include/fanlin_gpu.h and linked to
fanlin-rs_amd/libfanlin_gpu.so; on the GPU box it runs requests given as query strings."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_client.c")
LIBDIR = os.path.join(ROOT, "fanlin-rs_amd")


def build(tmp_path, src=SRC, name="c_client"):
    exe = str(tmp_path / name)
    subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-std=c11", src, "-I", os.path.join(ROOT, "include"), "-L", LIBDIR,
                    "-lfanlin_gpu", "-lpthread", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def lcg_image(h, w, c):
    out = np.empty(h * w * c, np.uint8)
    s = 0xFA171200
    for i in range(out.size):                     # small pictures only
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = s >> 24
    return out.reshape(h, w, c)


def test_c_client_compiles_and_links_against_the_header(fl, tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe, "w=20&h=oops", "8", "8", "3", str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 3 and "query rejected" in r.stderr            # axum would answer 400 (host only, no device needed)
    r = subprocess.run([exe, "rgb=1,2,3&quality=80", "8", "8", "3", str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 4                                               # query.rs:80-87 as_is: the image is not touched


@pytest.mark.gpu
@pytest.mark.parametrize("query,shape", [("w=120&h=90", (96, 160, 3)), ("w=64&h=64&crop=true&grayscale=true", (80, 120, 4)),
                                          ("w=100&h=100&blur=12&rgb=200,10,10", (60, 90, 3)), ("inverse=true", (33, 17, 1))])
def test_c_client_matches_oracle(fl, oracle, tmp_path, query, shape):
    import oracle_lib
    exe = build(tmp_path)
    out = str(tmp_path / "out.bin")
    h, w, c = shape
    r = subprocess.run([exe, query, str(w), str(h), str(c), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = open(out, "rb").read()
    head, _, body = raw.partition(b"\n")
    ow, oh, oc, nbytes = (int(x) for x in head.split())
    got = np.frombuffer(body, np.uint8).reshape(oh, ow, oc)
    assert nbytes == got.size
    q = fl.Query.parse(query)
    dims = q.dimensions()
    kw = dict(w=dims[0] if dims else None, h=dims[1] if dims else None, fill=q.fill_color(), crop=q.cropping(), blur_sigma=q.blur(),
              grayscale=q.grayscale(), inverse=q.inverse())
    img = lcg_image(h, w, c)
    import parity
    parity.check_pixels_any_kernel(oracle, got, img, **kw)


def test_c_stress_compiles(fl, tmp_path):
    build(tmp_path, os.path.join(ROOT, "tests", "c_stress.c"), "c_stress")


@pytest.mark.gpu
def test_mixed_concurrent_requests_equal_sequential_ones(fl, tmp_path):
    # 24 threads x 12 mixed requests through the multi-lane queue, then the same 288 requests one at a time
    import json
    exe = build(tmp_path, os.path.join(ROOT, "tests", "c_stress.c"), "c_stress")
    r = subprocess.run([exe, "24", "12"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["requests"] == 288 and out["mismatches"] == 0
    assert out["queue_flushes"] < 2 * 288          # the concurrent half really was batched


def build_cpp(tmp_path):
    exe = str(tmp_path / "cpp_host")
    subprocess.run(["g++", "-O1", "-Wall", "-Wextra", "-Werror", "-std=c++17", os.path.join(ROOT, "tests", "cpp_host.cpp"), "-I", os.path.join(ROOT, "include"),
                    "-L", LIBDIR, "-lfanlin_gpu", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_cpp_mirror_of_query_and_format(fl, tmp_path):
    # include/fanlin_gpu.hpp: query::Query / content::Format / handler::State with the reference's names
    r = subprocess.run([build_cpp(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "host ok" in r.stdout, r.stderr


@pytest.mark.gpu
def test_cpp_state_process_image(fl, tmp_path):
    r = subprocess.run([build_cpp(tmp_path), "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gpu ok" in r.stdout, r.stderr
