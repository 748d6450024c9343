#!/usr/bin/env python3
"""Generate tests/golden/*.npy with the CPU oracle (run in the dev container).

The reference ships no fixtures and cannot be built here, so the golden
vectors are the oracle's own output: they pin the oracle AND the HIP path against
silent drift, not against the real Rust binary ("parity unpinned")."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402  (registers the package)
import oracle_lib  # noqa: E402

conftest._build_oracle()
o = oracle_lib.Oracle()
desc = o.scene_analytical()
out = os.path.join(ROOT, "tests", "golden")
os.makedirs(out, exist_ok=True)
np.save(os.path.join(out, "analytical_64x48_spp4_seed1.npy"), o.render(desc, 64, 48, 4, seed=1))
np.save(os.path.join(out, "analytical_32x24_spp16_seed7.npy"), o.render(desc, 32, 24, 16, seed=7))
np.save(os.path.join(out, "rng_seed1_frame0_pixel0_u32x16.npy"), o.rng_u32(1, 0, 0, 16))
print("wrote", sorted(os.listdir(out)))
