"""Loads the HIP extension.  There is no CPU fallback: if librpt_hip.so is missing or
cannot be loaded, importing the product API fails loudly."""
import ctypes as C
import os

from . import _abi

HERE = os.path.dirname(os.path.abspath(__file__))
# librpt_hip.so is the product (exactly include/rpt.h).  RPT_LIB selects another build of it: librpt_hip_test.so (the same objects
# linked with the hooks of include/rpt_test.h — what tests/conftest.py loads), or an experiment build (tools/build_variants.py).
LIB_PATH = os.environ.get("RPT_LIB", os.path.join(HERE, "librpt_hip.so"))
TEST_LIB_PATH = os.path.join(HERE, "librpt_hip_test.so")

_lib = None


class RptError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("rpt status %d: %s" % (status, message))
        self.status = status


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python rust-pathtracer_amd/build.py` or __graft_entry__.build(); "
                "the product path has no CPU fallback" % LIB_PATH)
        # torch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.  If this library were
        # loaded first it would bind to /opt/rocm's copies and torch would then bring in a second
        # HSA runtime, which cannot initialise (one /dev/kfd client per process): "no
        # ROCm-capable device".  Importing torch first makes its runtime the process's only one.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _abi.SYMBOLS.items():
            fn = getattr(l, name)      # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if l.rpt_build_has_test_hooks():
            for name, (res, args) in _abi.TEST_SYMBOLS.items():
                fn = getattr(l, name)
                fn.restype = res
                fn.argtypes = args
        _lib = l
        # Profiling runs (tools/collect_profiles.sh) ask which library the profiled process REALLY loaded: counters are stamped with
        # that file's hash, not with what the collecting script assumes.
        record = os.environ.get("RPT_LOADED_LIB_RECORD")
        if record:
            import hashlib
            with open(record, "w") as f:
                f.write("%s %s\n" % (hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest(), os.path.abspath(LIB_PATH)))
    return _lib


def check(status, ctx=None):
    if status != _abi.RPT_OK:
        msg = lib().rpt_last_error(ctx)
        raise RptError(status, msg.decode() if msg else "")
