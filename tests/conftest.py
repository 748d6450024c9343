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
    libdir = os.path.join(ROOT, "oracle", "build")
    need = ["liboracle.so", "liboracle_libm.so", "liboracle_opcount.so"]
    if not all(os.path.exists(os.path.join(libdir, n)) for n in need):
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


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
