import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "rust-pathtracer_amd")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def load_package():
    """The package directory is `rust-pathtracer_amd` (not an identifier): register it
    as `rust_pathtracer_amd`."""
    name = "rust_pathtracer_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# The tests run on the TEST BUILD: the product's own objects linked with the hooks of include/rpt_test.h (probes, dispatch
# read-outs, the knob reload).  tests/test_shipped_library.py checks the product library itself: its exports, and that it renders
# the same frames.  RPT_LIB overrides (experiment builds).
os.environ.setdefault("RPT_LIB", os.path.join(PKG_DIR, "librpt_hip_test.so"))
load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_oracle():
    # always: make is a no-op when the libraries are newer than the oracle's sources, and a stale one would test yesterday's oracle
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)


@pytest.fixture(scope="session")
def oracle():
    _build_oracle()
    import oracle_lib
    return oracle_lib.Oracle("liboracle.so")


@pytest.fixture(scope="session")
def oracle_libm():
    _build_oracle()
    import oracle_lib
    return oracle_lib.Oracle("liboracle_libm.so")


@pytest.fixture(scope="session")
def oracle_opcount():
    _build_oracle()
    import oracle_lib
    return oracle_lib.Oracle("liboracle_opcount.so")


@pytest.fixture(scope="session")
def rpt():
    return load_package()


@pytest.fixture(autouse=True)
def _knobs_follow_the_environment():
    """The library reads its environment knobs ONCE per process (csrc/knobs.h).  Tests that change one call the test build's
    rpt_debug_reload_knobs themselves; behind every test the knobs are read again, so that a change undone by monkeypatch (or a
    `finally`) does not leak into the next test."""
    yield
    pkg = load_package()
    if pkg._lib._lib is not None and pkg._lib._lib.rpt_build_has_test_hooks():
        pkg._lib._lib.rpt_debug_reload_knobs()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
