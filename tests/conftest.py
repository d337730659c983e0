"""pytest configuration: the `gpu` marker, the package loader (the package directory is
named after the reference, `fanlin-rs_amd`, which is not a Python identifier) and the
CPU-oracle fixture."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    if "fanlin_rs_amd" in sys.modules:
        return sys.modules["fanlin_rs_amd"]
    pkg_dir = os.path.join(ROOT, "fanlin-rs_amd")
    spec = importlib.util.spec_from_file_location("fanlin_rs_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["fanlin_rs_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def fl():
    mod = load_package()
    if not os.path.exists(mod.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return mod


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


def require_device():
    """On the GPU box (gpurun exports GRAFT_REPO_ROOT) a gpu-marked test without a device FAILS -- a silent skip there
    would hide that the HIP path never ran; on a CPU-only machine a plain `pytest tests` skips those tests."""
    import torch
    if torch.cuda.is_available():
        return
    if os.environ.get("GRAFT_REPO_ROOT") or os.environ.get("FLGPU_REQUIRE_DEVICE"):
        pytest.fail("this test is marked gpu but no HIP device is visible")
    pytest.skip("needs a HIP device (run with -m gpu on the GPU box)")


_SESSION_STATE = None


@pytest.fixture(scope="session")
def gpu_state(fl):
    global _SESSION_STATE
    require_device()
    st = fl.State(device=0, profile=True)
    _SESSION_STATE = st
    import parity
    parity.STATE = st   # (parity.packed_arithmetic asks the context which arithmetic it runs)
    yield st
    _SESSION_STATE = None
    parity.STATE = None
    st.close()


@pytest.fixture(autouse=True)
def _switches_back_to_their_defaults():
    """Tests flip the context's switches with gpu_state.debug_set (the library reads no environment after flgpu_create); whatever a
    test set is undone here, so the session's one context starts every test as flgpu_create left it."""
    yield
    if _SESSION_STATE is not None:
        _SESSION_STATE.debug_set("reset")
