"""Participating media in the oracle (CPU).  PROJECT-DEFINED behaviour (include/rpt.h, "participating media"): the reference
declares Medium and never reads it, so nothing here is a reference result — these are known answers of the specification
itself: Beer-Lambert transmittance along a known chord, the emission integral, the free-flight distribution, a furnace, and
the Henyey-Greenstein pair (phase function normalised, sampling matches it).  With the scene flag off every medium field is
ignored, exactly like the reference."""
import ctypes as C

import numpy as np
import pytest


def ball_scene(rpt, medium, bg=(1.0, 1.0, 1.0), depth=8, media=True, ior=1.001):
    """An almost index-matched, perfectly transmissive unit sphere (rgb 1, spec_trans 1, ior 1.001 — at exactly 1 the refraction
    Jacobian of tracer.rs:397 is 0/0: every path goes practically straight through with
    weight ~1) filled with `medium`, seen against a constant background, no lights."""
    s = rpt.Scene()
    s.camera = rpt.Pinhole((0.0, 0.0, 4.0), (0.0, 0.0, 0.0), 40.0)
    s.background = dict(kind=rpt._abi.RPT_BG_CONSTANT, colour_a=bg, colour_b=(0.0, 0.0, 0.0), gamma=2.2, scale=1.0)
    s.materials = [rpt.Material(rgb=(1.0, 1.0, 1.0), spec_trans=1.0, ior=ior, roughness=0.0, medium=medium)]
    s.spheres = [((0.0, 0.0, 0.0), 1.0, 0)]
    s.max_depth = depth
    s.media = media
    return s


def inside_segment(rays):
    """Length of the path's segment inside the unit sphere: from the origin of its second ray (just inside the surface) along
    its direction to the far intersection."""
    o, d = rays[1, :3].astype(np.float64), rays[1, 3:6].astype(np.float64)
    b = o @ d
    return -b + np.sqrt(b * b - (o @ o - 1.0))


def test_media_are_ignored_without_the_scene_flag(rpt, oracle):
    med = dict(type="scatter", density=3.0, color=(0.9, 0.5, 0.2), anisotropy=0.4)
    a = oracle.render(ball_scene(rpt, med, media=False, ior=1.3).describe(), 40, 30, 3, seed=2)
    b = oracle.render(ball_scene(rpt, None, media=False, ior=1.3).describe(), 40, 30, 3, seed=2)
    c = oracle.render(ball_scene(rpt, None, media=True, ior=1.3).describe(), 40, 30, 3, seed=2)      # flag on, nothing to act on
    d = oracle.render(ball_scene(rpt, dict(type="none", density=3.0, color=(0.9, 0.5, 0.2)), media=True, ior=1.3).describe(), 40, 30, 3, seed=2)
    assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d)
    e = oracle.render(ball_scene(rpt, med, media=True, ior=1.3).describe(), 40, 30, 3, seed=2)
    assert not np.array_equal(a, e)


def test_absorbing_ball_is_beer_lambert_along_the_chord(rpt, oracle):
    """An absorbing medium draws no random numbers, so a path is the same with and without it and its radiance differs by
    exactly the transmittance over the segment inside: exp(-(1 - color) * density * length) per channel."""
    color, density = (0.8, 0.5, 0.2), 1.5
    w = h = 33
    on = ball_scene(rpt, dict(type="absorb", density=density, color=color))
    off = ball_scene(rpt, None)
    img_on = oracle.render(on.describe(), w, h, 1, seed=3)
    img_off = oracle.render(off.describe(), w, h, 1, seed=3)
    desc = on.describe()
    checked = 0
    for col, row in [(16, 16), (12, 16), (16, 11), (20, 19), (9, 16), (16, 24), (13, 13), (22, 16)]:
        rays = oracle.sample_rays(desc, col, row, 0, w, h, seed=3)
        assert len(rays) == 3, "camera ray, the ray inside, the ray behind"        # enter, leave, miss
        seg = inside_segment(rays)
        want = np.exp(-(1.0 - np.array(color)) * density * seg)
        got = img_on[row, col, :3].astype(np.float64) / img_off[row, col, :3].astype(np.float64)
        assert np.allclose(got, want, rtol=2e-5), (col, row, got, want)
        checked += 1
    assert checked == 8
    assert np.array_equal(img_on[0, 0], img_off[0, 0])                              # a path that misses the ball


def test_emissive_ball_adds_color_times_length_times_density(rpt, oracle):
    color, density = (0.3, 0.6, 0.9), 0.7
    w = h = 33
    on = ball_scene(rpt, dict(type="emissive", density=density, color=color), bg=(0.0, 0.0, 0.0))
    glow = oracle.render(on.describe(), w, h, 1, seed=5)
    # the weight of the path's first surface crossing: what a constant background of 1 behind an EMPTY ball shows is that
    # weight times the second crossing's; both are within a few 1e-3 of 1 for this material, so the square root will do
    through = oracle.render(ball_scene(rpt, None).describe(), w, h, 1, seed=5)
    desc = on.describe()
    for col, row in [(16, 16), (12, 16), (16, 11), (20, 19)]:
        seg = inside_segment(oracle.sample_rays(desc, col, row, 0, w, h, seed=5))
        w1 = np.sqrt(through[row, col, :3].astype(np.float64))
        want = np.array(color) * seg * density * w1
        assert np.allclose(glow[row, col, :3], want, rtol=5e-3), (col, row, glow[row, col, :3], want)
    assert not glow[0, 0, :3].any()


def test_black_scattering_ball_shows_the_unscattered_fraction(rpt, oracle):
    """Albedo 0: a path that scatters is lost, so what comes through is the probability of crossing the ball without an event,
    exp(-density * length) — the free-flight sampling d = -ln(r) / density against the segment length."""
    density = 0.6
    w = h = 9
    spp = 6000
    on = oracle.render(ball_scene(rpt, dict(type="scatter", density=density, color=(0.0, 0.0, 0.0))).describe(), w, h, spp, seed=7)
    off = oracle.render(ball_scene(rpt, None).describe(), w, h, spp, seed=7)
    got = on[4, 4, 0] / off[4, 4, 0]
    # the central pixel's rays pass within 0.05 of the centre: segments of 2 * sqrt(1 - b^2) - eps
    want = np.exp(-density * 1.99)
    assert abs(got - want) < 3.5 * np.sqrt(want * (1 - want) / spp) + 0.004, (got, want)


@pytest.mark.parametrize("g", [0.0, 0.7, -0.5])
def test_white_scattering_ball_is_a_furnace(rpt, oracle, g):
    """Albedo 1, constant surroundings, no absorption anywhere: whatever the density and the phase function, every path leaves
    the ball sooner or later and picks up the same background — the ball stays invisible (up to the surface weights, which
    differ from 1 by a few 1e-3, and the paths cut off at max_depth)."""
    w = h = 15
    spp = 400
    med = dict(type="scatter", density=2.0, color=(1.0, 1.0, 1.0), anisotropy=g)
    on = oracle.render(ball_scene(rpt, med, depth=200).describe(), w, h, spp, seed=11)
    off = oracle.render(ball_scene(rpt, None, depth=200).describe(), w, h, spp, seed=11)
    disc = (slice(5, 10), slice(5, 10), slice(0, 3))
    assert abs(on[disc].mean() / off[disc].mean() - 1.0) < 0.02
    assert abs(off[disc].mean() - 1.0) < 0.02


def test_light_inside_fog_is_attenuated_and_scattered(rpt, oracle):
    """A light INSIDE a scattering ball (any_hit honours max_dist, so it is visible from inside): brighter fog in front of a black
    background than the same ball with black (albedo 0) fog — in-scattering reaches the camera only when the albedo is not 0."""
    def scene(color):
        s = ball_scene(rpt, dict(type="scatter", density=1.2, color=color, anisotropy=0.3), bg=(0.0, 0.0, 0.0), depth=12)
        s.lights = [rpt.AnalyticalLight.spherical((0.0, 0.0, 0.0), 0.15, (20.0, 20.0, 20.0))]
        s.any_hit_uses_max_dist = True
        return s
    lit = oracle.render(scene((0.9, 0.9, 0.9)).describe(), 21, 21, 64, seed=13)
    dark = oracle.render(scene((0.0, 0.0, 0.0)).describe(), 21, 21, 64, seed=13)
    rim = (slice(8, 13), slice(4, 7), slice(0, 3))                                 # next to the light's disc, inside the ball's
    assert lit[rim].mean() > 4.0 * dark[rim].mean() + 1e-3
    assert np.isfinite(lit).all() and np.isfinite(dark).all()


def test_phase_function_is_normalised_and_sampling_matches_it(oracle):
    lib = oracle.lib
    lib.oracle_phase_hg.restype = C.c_float
    lib.oracle_phase_hg.argtypes = [C.c_float, C.c_float]
    lib.oracle_sample_hg.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
    rng = np.random.default_rng(3)
    for g in (0.0, 0.0005, 0.3, -0.6, 0.9, -0.9):
        c = np.linspace(-1.0, 1.0, 40001)
        p = np.array([lib.oracle_phase_hg(float(x), g) for x in c])
        integral = 2.0 * np.pi * np.trapezoid(p, c)
        assert abs(integral - 1.0) < 2e-3, (g, integral)
        v = rng.normal(size=3)
        v = (v / np.linalg.norm(v)).astype(np.float32)
        out = np.zeros(3, dtype=np.float32)
        cos = []
        for _ in range(20000):
            lib.oracle_sample_hg(v.ctypes.data, g, float(rng.random()), float(rng.random()), out.ctypes.data)
            assert abs(np.linalg.norm(out) - 1.0) < 1e-4
            cos.append(float(out @ v))
        # against v = the direction BACK along the ray the mean cosine of Henyey-Greenstein is -g
        assert abs(np.mean(cos) + g) < 0.02, (g, np.mean(cos))
        # and the sampled density is the phase function: P(cos < 0) from both
        frac = np.mean(np.array(cos) < 0.0)
        want = 2.0 * np.pi * np.trapezoid(p[c <= 0.0], c[c <= 0.0])
        assert abs(frac - want) < 0.02, (g, frac, want)
