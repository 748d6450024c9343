"""MI355X-native drop-in for rust-pathtracer's Tracer::render hot path.

The directory is named `rust-pathtracer_amd`; import it as `rust_pathtracer_amd`
(tests/conftest.py and __graft_entry__.py register it under that name).
"""
from . import _abi  # noqa: F401
from ._lib import RptError, lib  # noqa: F401
from .api import (AnalyticalLight, AnalyticalScene, ColorBuffer, DeviceColorBuffer, Material, Pinhole, Scene,  # noqa: F401
                  Tracer, comm_unique_id)
