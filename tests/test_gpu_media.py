"""Participating media on the device against the oracle, bit for bit, in every kernel form (needs an MI355X).
The behaviour is PROJECT-DEFINED (include/rpt.h, "participating media"; known answers: tests/test_oracle_media.py)."""
import ctypes as C
import os

import numpy as np
import pytest

import conftest

from test_gpu_parity import _probe, _random_floats, _random_small_scene, assert_bit_identical, torch_cuda, tracer  # noqa: F401

pytestmark = pytest.mark.gpu


def test_probe_exp_log(rpt, torch_cuda, tracer, oracle):
    rng = np.random.default_rng(21)
    a = np.concatenate([_random_floats(rng, 2_000_000), rng.uniform(-100, 100, 1_000_000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, np.inf, -np.inf, np.nan, 88.8, -104.0, 1e-45], dtype=np.float32)])
    assert_bit_identical(_probe(rpt, torch_cuda, tracer, rpt._abi.RPT_PROBE_EXP, a), oracle.math(7, a), "exp")
    b = np.concatenate([_random_floats(rng, 2_000_000), rng.uniform(0, 1, 1_000_000).astype(np.float32), a[-9:]])
    assert_bit_identical(_probe(rpt, torch_cuda, tracer, rpt._abi.RPT_PROBE_LOG, b), oracle.math(8, b), "log")


FORMS = [("megakernel", 0, 5), ("compacting (default at 1 spp)", 0, 1),
         ("compacting, forced", "RPT_RENDER_SMALL_COMPACT", 4)]


@pytest.mark.parametrize("form,flag,spp", FORMS)
@pytest.mark.parametrize("rr", [False, True])
def test_media_scene_matches_oracle(rpt, oracle, form, flag, spp, rr):
    from rust_pathtracer_amd import scenes
    A = rpt._abi
    s = scenes.media_scene()
    flags = (getattr(A, flag) if flag else 0) | (A.RPT_RENDER_RUSSIAN_ROULETTE if rr else 0)
    w, h = 96, 72
    t = rpt.Tracer(s, device=0, seed=9)
    t.flags = flags
    buf = rpt.ColorBuffer(w, h)
    t.render_n(buf, spp)
    t.render_n(buf, 1)                                                   # resumed accumulation
    want = oracle.render(s.describe(), w, h, spp + 1, seed=9, render_flags=A.RPT_RENDER_RUSSIAN_ROULETTE if rr else 0)
    assert_bit_identical(buf.image(), want, "media scene, %s, roulette %s" % (form, rr))
    # the media do something: the same scene with the flag off is another image
    s.media = False
    assert not np.array_equal(want, oracle.render(s.describe(), w, h, spp + 1, seed=9, render_flags=A.RPT_RENDER_RUSSIAN_ROULETTE if rr else 0))
    t.close()


def _add_random_media(rpt, s, rng):
    """Random media on a random small scene: most materials transmissive so that paths get inside, every medium type,
    densities from thin to opaque (0 and huge included), anisotropies beyond the clamp, partial patches that carry only a medium."""
    s.media = True
    for m in s.materials:
        if rng.random() < 0.6:
            m.fields["spec_trans"] = float(rng.choice([1.0, 1.0, rng.uniform(0.3, 1.0)]))
            m.fields["metallic"] = 0.0
            m.fields.setdefault("rgb", tuple(rng.uniform(0.5, 1, 3)))
        if rng.random() < 0.7:
            m.medium = dict(type=str(rng.choice(["scatter", "scatter", "absorb", "emissive", "none"])),
                            density=float(rng.choice([0.0, 0.3, 1.5, 8.0, 1e4, rng.uniform(0.1, 3)])),
                            color=tuple(rng.uniform(0, 1, 3) * (rng.random() < 0.9)), anisotropy=float(rng.uniform(-1.2, 1.2)))
    s.max_depth = int(rng.integers(2, 14))


@pytest.mark.parametrize("seed", range(36))
def test_random_small_scenes_with_media_match_oracle(rpt, oracle, seed):
    rng = np.random.default_rng(5000 + seed)
    s = _random_small_scene(rpt, rng)
    while not s.spheres and not s.planes:
        s = _random_small_scene(rpt, rng)
    _add_random_media(rpt, s, rng)
    A = rpt._abi
    w, h, spp = int(rng.integers(8, 70)), int(rng.integers(8, 50)), int(rng.integers(1, 4))
    rr = A.RPT_RENDER_RUSSIAN_ROULETTE if seed % 4 == 3 else 0
    t = rpt.Tracer(s, device=0, seed=seed)
    t.flags = (0, 0, A.RPT_RENDER_SMALL_COMPACT)[seed % 3] | rr
    buf = rpt.ColorBuffer(w, h)
    t.render_n(buf, spp)
    want = oracle.render(s.describe(), w, h, spp, seed=seed, render_flags=rr)
    assert_bit_identical(buf.image(), want, "media fuzz seed %d (%dx%d x%d, %d spheres %d planes %d lights depth %d)" %
                         (seed, w, h, spp, len(s.spheres), len(s.planes), len(s.lights), s.max_depth))
    t.close()


def test_sdf_object_full_of_fog_matches_oracle_in_every_sdf_form(rpt, oracle):
    from rust_pathtracer_amd import scenes
    A = rpt._abi
    s = scenes.sdf_scene()
    s.media = True
    s.max_depth = 8
    s.any_hit_uses_max_dist = True
    s.materials[0] = rpt.Material(rgb=(0.9, 0.95, 1.0), roughness=0.1, spec_trans=1.0, ior=1.25,
                                  medium=dict(type="scatter", density=2.5, color=(0.7, 0.85, 1.0), anisotropy=-0.3))
    s.materials[1] = rpt.Material(rgb=(1.0, 0.6, 0.3), roughness=0.1, spec_trans=1.0, ior=1.4, medium=dict(type="absorb", density=2.0, color=(1.0, 0.4, 0.1)))
    w, h, spp = 80, 60, 3
    want = oracle.render(s.describe(), w, h, spp, seed=4)
    for name, flags in (("march kernel (two rooms)", 0),):
        t = rpt.Tracer(s, device=0, seed=4)
        t.flags = flags
        buf = rpt.ColorBuffer(w, h)
        t.render_n(buf, spp)
        assert_bit_identical(buf.image(), want, "SDF scene with media, " + name)
        t.close()


@pytest.mark.parametrize("form", ["megakernel"])
@pytest.mark.parametrize("n_spheres,rr", [(300, False), (700, True)])
def test_large_scene_with_media_matches_oracle(rpt, oracle, form, n_spheres, rr):
    from rust_pathtracer_amd import scenes
    A = rpt._abi
    s = scenes.random_spheres_scene(n_spheres=n_spheres, n_lights=5, media=True, n_palette=24)
    s.max_depth = 9
    rflag = A.RPT_RENDER_RUSSIAN_ROULETTE if rr else 0
    w, h, spp = 72, 40, 2
    t = rpt.Tracer(s, device=0, seed=6)
    t.flags = rflag
    buf = rpt.ColorBuffer(w, h)
    t.render_n(buf, spp)
    t.render_n(buf, 1)
    want = oracle.render(s.describe(), w, h, spp + 1, seed=6, render_flags=rflag)
    assert_bit_identical(buf.image(), want, "large scene with media, %s, %d spheres" % (form, n_spheres))
    s.media = False
    assert not np.array_equal(want, oracle.render(s.describe(), w, h, spp + 1, seed=6, render_flags=rflag))
    t.close()


def test_media_on_virtual_ranks(rpt, oracle, torch_cuda):
    from rust_pathtracer_amd import scenes
    s = scenes.media_scene()
    w, h = 64, 45
    os.environ["RPT_GATHER"] = "p2p"
    try:
        t = rpt.Tracer(s, devices=[0, 0, 0], seed=2)
    finally:
        os.environ.pop("RPT_GATHER", None)
    t.render_resident(w, h, 3)
    assert_bit_identical(t.resident_to_host(w, h).image(), oracle.render(s.describe(), w, h, 3, seed=2), "media on 3 virtual ranks")
    t.close()


def test_media_error_paths(rpt, torch_cuda):
    from rust_pathtracer_amd import scenes
    A = rpt._abi
    t = rpt.Tracer(scenes.media_scene(), device=0, seed=1)
    t.flags = A.RPT_RENDER_FAST_MATH
    with pytest.raises(rpt.RptError) as e:
        t.render_n(rpt.ColorBuffer(16, 16), 1)
    assert e.value.status == A.RPT_ERR_UNSUPPORTED and "media" in str(e.value)
    t.close()
    s = scenes.media_scene()
    s.materials[0].medium["density"] = -1.0
    with pytest.raises(rpt.RptError) as e:
        rpt.Tracer(s, device=0)
    assert e.value.status == A.RPT_ERR_INVALID_ARG
    # a large scene with media needs the medium on every sphere material
    big = scenes.random_spheres_scene(n_spheres=100, n_lights=2, media=True, n_palette=12)
    big.materials[4] = scenes.full_material(rgb=(0.5, 0.5, 0.5))
    with pytest.raises(rpt.RptError) as e:
        rpt.Tracer(big, device=0)
    assert e.value.status == A.RPT_ERR_UNSUPPORTED and "RPT_MAT_MEDIUM" in str(e.value)
