"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU code (never on the GPU): `make -C oracle asan`
builds the oracle and the host harness (tests/host_harness.cpp: the product's grid builder csrc/host_grid.h and its tiling
arithmetic csrc/tile_plan.h) with -fsanitize=address,undefined; the harness runs as it is, the oracle through ctypes in a child
interpreter with libasan preloaded: the stock scene, a fuzzed small scene, a 200-sphere scene, the SDF scene, media, the
denoiser, the u8 conversions and the ray log."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "oracle", "build")


@pytest.fixture(scope="module")
def asan_build():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return BUILD


def _env():
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    env = dict(os.environ)
    env.update({"LD_PRELOAD": libasan, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=23",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1:exitcode=24", "OMP_NUM_THREADS": "4"})
    return env


def test_host_grid_builder_and_tiling_arithmetic_are_clean(asan_build):
    r = subprocess.run([os.path.join(asan_build, "host_harness_asan"), "histogram"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_harness: ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr


CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import conftest
rpt = conftest.load_package()
import oracle_lib
from rust_pathtracer_amd import scenes
import test_gpu_parity
o = oracle_lib.Oracle("liboracle_asan.so")
A = rpt._abi
stock = o.scene_analytical()
img = o.render(stock, 48, 36, 3, seed=1)
assert np.isfinite(img).all() and img[..., :3].mean() > 0.05
assert np.array_equal(o.render(rpt.AnalyticalScene().describe(), 48, 36, 3, seed=1), img)
rng = np.random.default_rng(11)
for k in range(3):
    s = test_gpu_parity._random_small_scene(rpt, rng)
    o.render(s.describe(), 40, 28, 2, seed=3 + k)
big = scenes.random_spheres_scene(200, 4)
o.render(big.describe(), 32, 24, 2, seed=1)
o.render(scenes.sdf_scene().describe(), 40, 28, 2, seed=1)
o.render(scenes.media_scene().describe(), 40, 28, 3, seed=1)
deep = rpt.AnalyticalScene(); deep.max_depth = 12
o.render(deep.describe(), 32, 24, 2, seed=1, render_flags=A.RPT_RENDER_RUSSIAN_ROULETTE)
o.render(stock, 33, 17, 2, seed=1, rows=(3, 9))
o.sample_pixels(stock, np.array([0, 5, 47], dtype=np.uint32), np.array([0, 7, 35], dtype=np.uint32), np.array([0, 1, 2], dtype=np.uint64), 48, 36)
o.sample_rays(stock, 20, 30, 0, 48, 36)
o.convert_to_u8(img, 48, 36)
frame = np.zeros((50, 60, 4), dtype=np.uint8)
o.convert_to_u8_at(img, 48, 36, frame, (3, 4, 60, 50))
o.denoise(img, 48, 36, 3, 2.0)
o.math(A.RPT_PROBE_POW, np.linspace(0.0, 4.0, 1000, dtype=np.float32), np.full(1000, 2.2, dtype=np.float32))
o.rng_f32(1, 0, 17, 64)
print("oracle under asan+ubsan: ok")
'''


def test_oracle_is_clean_under_asan_and_ubsan(asan_build):
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, env=_env(), timeout=600, cwd=ROOT)
    assert r.returncode == 0, "exit %d\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
    assert "oracle under asan+ubsan: ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
