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


def ab_built():
    """Does the library under test hold the kernel forms kept for A/B runs (include/rpt.h, rpt_build_has_ab_kernels)?  The shipped
    library does not; `python rust-pathtracer_amd/build.py --ab` + RPT_LIB=rust-pathtracer_amd/librpt_hip_ab.so runs the tests on one that does."""
    return bool(load_package().lib().rpt_build_has_ab_kernels())


def only_in_ab_builds(*items):
    """`items` when the library holds the A/B kernel forms, nothing otherwise: for loops over kernel forms."""
    return tuple(items) if ab_built() else ()


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    """A test that asks the shipped library for an A/B-only kernel form is skipped from there on (what it checked before stands)."""
    outcome = yield
    exc = outcome.excinfo
    if exc and exc[0].__name__ == "RptError" and "-DRPT_AB_KERNELS" in str(exc[1]):
        outcome.force_exception(pytest.skip.Exception("needs a library with the A/B kernel forms (build.py --ab): %s" % exc[1]))


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
