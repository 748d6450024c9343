"""Hand-derived known-answer tests for the CPU oracle's leaf functions.

The reference has no tests, so these answers are worked out by hand from
the formulas in the cited reference lines; they pin the oracle's transcription.  CPU only."""
import math

import numpy as np
import pytest


def test_sphere_front_hit(oracle):
    # analytical.rs:166-190: origin (0,0,3) looking down -z at the unit sphere: tca = 3, d2 = 0, thc = 1 -> t = 2
    assert oracle.sphere((0, 0, 3), (0, 0, -1), (0, 0, 0), 1.0) == (True, 2.0)


def test_sphere_origin_inside_returns_far_root(oracle):
    # t0 = -1 < 0 -> t0 = t1 = 1 (analytical.rs:182-187)
    assert oracle.sphere((0, 0, 0), (0, 0, -1), (0, 0, 0), 1.0) == (True, 1.0)


def test_sphere_miss_and_behind(oracle):
    assert oracle.sphere((0, 2, 3), (0, 0, -1), (0, 0, 0), 1.0)[0] is False      # d2 = 4 > 1
    assert oracle.sphere((0, 0, 3), (0, 0, 1), (0, 0, 0), 1.0)[0] is False       # both roots negative
    # grazing: d2 == radius2 is a hit (the test is `d2 > radius2`)
    assert oracle.sphere((1, 0, 3), (0, 0, -1), (0, 0, 0), 1.0) == (True, 3.0)


def test_plane(oracle, rpt):
    p = rpt._abi.rpt_plane()
    p.normal = rpt._abi.F3(0, 1, 0); p.point = rpt._abi.F3(0, -1, 0); p.min_denom = 0.0001
    assert oracle.plane((0, 0, 3), (0, -1, 0), p) == (True, 1.0)                 # (-1 - 0) / -1
    assert oracle.plane((0, 0, 3), (0, 1, 0), p)[0] is False                     # t = -1 < 0
    assert oracle.plane((0, 0, 3), (1, 0, 0), p)[0] is False                     # denom = 0
    assert oracle.plane((0, 0, 3), (0, np.float32(-0.0001), 0), p)[0] is False   # |denom| must EXCEED 1e-4
    hit, t = oracle.plane((0, 1, 0), (0, -0.5, 0), p)
    assert hit and t == 4.0                                                      # (-1 - 1) / -0.5


def test_power_heuristic(oracle):
    L = oracle.lib
    assert L.oracle_power_heuristic(2.0, 2.0) == 0.5
    assert L.oracle_power_heuristic(1.0, 0.0) == 1.0
    assert L.oracle_power_heuristic(0.0, 1.0) == 0.0
    assert math.isnan(L.oracle_power_heuristic(0.0, 0.0))                        # 0/0: the reference has no guard (tracer.rs:223-226)
    assert L.oracle_power_heuristic(1.0, 3.0) == np.float32(1.0) / np.float32(10.0)


def test_schlick_fresnel(oracle):
    L = oracle.lib
    assert L.oracle_schlick_fresnel(0.0) == 1.0
    assert L.oracle_schlick_fresnel(1.0) == 0.0
    assert L.oracle_schlick_fresnel(0.5) == 0.03125                              # 0.5^5
    assert L.oracle_schlick_fresnel(2.0) == 0.0                                  # clamp(1-u, 0, 1)
    assert L.oracle_schlick_fresnel(-1.0) == 1.0


def test_dielectric_fresnel(oracle):
    L = oracle.lib
    # normal incidence, eta = 1/1.5: rs = rp = (eta-1)/(eta+1) = -0.2 -> 0.04
    assert abs(L.oracle_dielectric_fresnel(1.0, 1.0 / 1.5) - 0.04) < 1e-7
    # total internal reflection: eta^2 (1 - cos^2) = 2.25 > 1
    assert L.oracle_dielectric_fresnel(0.0, 1.5) == 1.0
    # eta = 1: no interface
    assert L.oracle_dielectric_fresnel(0.7, 1.0) == 0.0


def test_gtr1_uses_log2(oracle):
    L = oracle.lib
    # tracer.rs:233-240 with a = 0.5, ndoth = 1: a2 = .25, t = .25, (a2-1)/(pi*log2(.25)*t) = -.75/(pi*-2*.25)
    assert abs(L.oracle_gtr1(1.0, 0.5) - 0.75 / (math.pi * 0.5)) < 1e-6
    assert L.oracle_gtr1(0.3, 1.0) == np.float32(1.0) / np.float32(math.pi)      # a >= 1 -> INV_PI
    # the textbook GTR1 (natural log) is larger by 1/ln 2
    assert abs(L.oracle_gtr1(1.0, 0.5) / math.log(2) - (0.25 - 1) / (math.pi * math.log(0.25) * 0.25)) < 1e-6


def test_smith_and_gtr2(oracle):
    L = oracle.lib
    assert L.oracle_smithg(1.0, 0.25) == 1.0                                     # 2 / (1 + sqrt(a + 1 - a))
    assert abs(L.oracle_gtr2aniso(1.0, 0.0, 0.0, 0.2, 0.5) - 1.0 / (math.pi * 0.2 * 0.5)) < 1e-5
    assert abs(oracle.luminance((1.0, 1.0, 1.0)) - 1.0) < 1e-7
    assert abs(oracle.luminance((1.0, 0.0, 0.0)) - 0.212671) < 1e-7


def test_material_defaults_and_finalize(oracle):
    m = oracle.material_defaults()
    assert list(m[:3]) == [1.5, 1.5, 1.5]                                        # material.rs:85 (sic)
    assert m[8] == 0.5 and m[16] == np.float32(1.45) and m[7] == 0.0
    m2 = m.copy(); m2[8] = 0.05
    r, ccr, ax, ay = oracle.material_finalize(m2)
    assert r == np.float32(0.05) and ax == np.float32(0.05) and ay == np.float32(0.05)      # aspect = sqrt(1 - 0) = 1
    assert ccr == np.float32(0.1)                                                # mix(.1, .001, gloss = 0)
    m3 = m.copy(); m3[14] = 1.0; m3[8] = 0.001
    r, ccr, ax, ay = oracle.material_finalize(m3)
    assert r == np.float32(0.01)                                                 # roughness clamp
    assert ccr == np.float32(0.001)                                              # (1-1)*.1 + .001*1
    m4 = m.copy(); m4[6] = 1.0; m4[8] = 0.5                                      # anisotropic = 1: aspect = sqrt(.1)
    r, ccr, ax, ay = oracle.material_finalize(m4)
    assert abs(ax - 0.5 / math.sqrt(0.1)) < 1e-6 and abs(ay - 0.5 * math.sqrt(0.1)) < 1e-6


def test_gen_ray_centre_and_corner(oracle):
    cam = [0, 0, 3, 0, 0, 0, 80.0]
    r = oracle.gen_ray(cam, 0.5, 0.5, 0.0, 0.0, 800.0, 600.0)
    assert list(r[:3]) == [0, 0, 3]
    assert abs(r[3]) < 1e-6 and abs(r[4]) < 1e-6 and abs(r[5] + 1.0) < 1e-6
    # lower-left corner: rd = (-tan 40deg, -tan 40deg / (4/3), -1) normalised (pinhole.rs:43-56)
    hw = math.tan(math.radians(80.0) / 2)
    d = np.array([-hw, -hw / (800.0 / 600.0), -1.0])
    d /= np.linalg.norm(d)
    r = oracle.gen_ray(cam, 0.0, 0.0, 0.0, 0.0, 800.0, 600.0)
    assert np.allclose(r[3:], d, atol=2e-7)
    # the jitter moves the ray by one pixel at most: offset (1,1) from p equals offset (0,0) from p + pixel_size
    a = oracle.gen_ray(cam, 0.25, 0.25, 1.0, 1.0, 800.0, 600.0)
    b = oracle.gen_ray(cam, 0.25 + 1 / 800.0, 0.25 + 1 / 600.0, 0.0, 0.0, 800.0, 600.0)
    assert np.allclose(a, b, atol=2e-7)


def _empty_scene(rpt, bg):
    s = rpt.Scene()
    s.background = bg
    return s


def test_background_gradient_and_image_orientation(oracle, rpt):
    """No geometry: every pixel is background(ray) (analytical.rs:28-32).  The top row of the
    buffer must look up (more blue), the bottom row down: tracer.rs:29-46 maps j = 0 to the LAST
    memory row, coord.y = 0 = lower edge of the frustum."""
    s = _empty_scene(rpt, dict(kind=rpt._abi.RPT_BG_GRADIENT_Y, colour_a=(1, 1, 1), colour_b=(0.5, 0.7, 1.0), gamma=2.2, scale=0.5))
    w, h = 32, 24
    img = oracle.render(s.describe(), w, h, 1, seed=1)
    assert np.all(img[..., 3] == 1.0)
    assert img[0, w // 2, 0] < img[h - 1, w // 2, 0]            # red falls as the ray points up
    col = 5
    for row in (0, h - 1):
        j = h - 1 - row
        offs = oracle.rng_f32(1, 0, row * w + col, 2)
        yy = np.float32(np.float32(h) - np.float32(j)) / np.float32(h)
        r = oracle.gen_ray([0, 0, 3, 0, 0, 0, 80.0], np.float32(col) / np.float32(w), np.float32(1.0) - yy, offs[0], offs[1], w, h)
        t = 0.5 * (np.float64(r[4]) + 1.0)
        c = (1.0 - t) * np.ones(3) + t * np.array([0.5, 0.7, 1.0])
        want = c ** 2.2 * 0.5
        assert np.allclose(img[row, col, :3], want, rtol=2e-6), (row, img[row, col], want)


def test_running_mean_is_the_reference_expression(oracle, rpt):
    """pixel = (1-v)*pixel + v*color with v = 1/(frames+1) (tracer.rs:105-117); constant
    background -> the mean stays the constant, alpha becomes exactly 1 after the first frame."""
    s = _empty_scene(rpt, dict(kind=rpt._abi.RPT_BG_CONSTANT, colour_a=(0.25, 0.5, 0.75), colour_b=(0, 0, 0), gamma=2.2, scale=1.0))
    img = oracle.render(s.describe(), 8, 6, 1, seed=1)
    assert np.all(img == np.array([0.25, 0.5, 0.75, 1.0], dtype=np.float32))
    img7 = oracle.render(s.describe(), 8, 6, 7, seed=1)
    # replay the f32 recurrence by hand
    p = np.zeros(4, dtype=np.float32)
    for f in range(7):
        v = np.float32(1.0) / np.float32(f + 1)
        p = (np.float32(1.0) - v) * p + np.array([0.25, 0.5, 0.75, 1.0], dtype=np.float32) * v
    assert np.all(img7 == p)


def test_directly_viewed_light_is_invisible_quirk(oracle, rpt):
    """scene.rs:66: sample_lights starts from state.hit_dist, which is -1 when the camera ray hit no
    geometry, so a light seen directly against the sky contributes nothing."""
    s = _empty_scene(rpt, dict(kind=rpt._abi.RPT_BG_CONSTANT, colour_a=(0.1, 0.1, 0.1), colour_b=(0, 0, 0), gamma=2.2, scale=1.0))
    s.lights = [rpt.AnalyticalLight.spherical((0.0, 0.0, 0.0), 1.0, (5.0, 5.0, 5.0))]      # dead centre of the view
    img = oracle.render(s.describe(), 16, 12, 2, seed=1)
    assert np.all(img[..., :3] == np.float32(0.1))


def test_material_layering_quirk(oracle, rpt):
    """analytical.rs:56-58 then :82-85: a ray whose line also crosses the LEFT sphere but hits the RIGHT
    one first gets the right sphere's rgb/clearcoat/roughness and KEEPS the left one's metallic = 1.
    With metallic = 1 the diffuse weight is 0 (tracer.rs:423); a grey metallic=0 right sphere would
    differ.  Compared on single samples from a camera placed on the +x side looking along -x."""
    base = rpt.AnalyticalScene()
    base.camera = rpt.Pinhole((6.0, 0.0, 0.0), (0.0, 0.0, 0.0), 20.0)
    swapped = rpt.AnalyticalScene()
    swapped.camera = rpt.Pinhole((6.0, 0.0, 0.0), (0.0, 0.0, 0.0), 20.0)
    swapped.spheres = [swapped.spheres[1], swapped.spheres[0]]          # test the right sphere first: no layering on it
    w, h = 16, 16
    a = oracle.render(base.describe(), w, h, 8, seed=2)
    b = oracle.render(swapped.describe(), w, h, 8, seed=2)
    centre = (slice(6, 10), slice(6, 10))
    assert not np.allclose(a[centre][..., :3], b[centre][..., :3])


def test_rng_golden_and_float_conversion(oracle):
    import os
    want = np.load(os.path.join(os.path.dirname(__file__), "golden", "rng_seed1_frame0_pixel0_u32x16.npy"))
    got = oracle.rng_u32(1, 0, 0, 16)
    assert np.array_equal(got, want)
    f = oracle.rng_f32(1, 0, 0, 16)
    assert np.array_equal(f, (got >> 8).astype(np.float32) * np.float32(2.0 ** -24))   # rand 0.8.5 Standard f32
    assert f.min() >= 0.0 and f.max() < 1.0
    # the PCG hash itself, spelled out (Jarzynski & Olano 2020)
    def pcg(v):
        state = (v * 747796405 + 2891336453) & 0xFFFFFFFF
        word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
        return ((word >> 22) ^ word) & 0xFFFFFFFF
    k = pcg(0); k = pcg(k ^ 1); k = pcg(k ^ 0); k0 = pcg(k ^ 0)            # frame_key(seed=1, frame=0): two folds
    j = pcg(0 ^ 0x85EBCA6B); j = pcg(j ^ 1); j = pcg(j ^ 0); k1 = pcg(j ^ 0)
    a = pcg(0); b = pcg(a)                                                # pixel 0
    state, inc = pcg(a ^ k0), pcg(b ^ k1) | 1
    # PCG-RXS-M-XS-32 with the path's own increment: an LCG step, then the output permutation
    def step(st):
        st = (st * 747796405 + inc) & 0xFFFFFFFF
        word = (((st >> ((st >> 28) + 4)) ^ st) * 277803737) & 0xFFFFFFFF
        return st, ((word >> 22) ^ word) & 0xFFFFFFFF
    for i in range(16):
        state, out = step(state)
        assert int(got[i]) == out, i


def test_rng_streams_of_neighbouring_paths_do_not_overlap(oracle):
    """Round 2's generator drew pcg_hash(key + counter): two paths whose 32-bit keys differed by less than a path's ~40 draws read
    overlapping windows of ONE sequence (about 5e-8 of all path pairs; ~1e10 pairs per full-HD frame).  Per-path streams
    share draws only when state AND increment coincide.  Over 200 000 paths x 48 draws — the pixels of a frame, consecutive
    frames, two seeds — no 4-draw window of one path may show up in another."""
    seen = {}
    clashes = 0
    paths = [(seed, frame, pixel) for seed in (1, 2) for frame in range(20) for pixel in range(5000)]
    for seed, frame, pixel in paths:
        d = oracle.rng_u32(seed, frame, pixel, 48)
        key = (int(d[8]), int(d[9]), int(d[10]), int(d[11]))                # one window in the middle of the stream ...
        seen[key] = (seed, frame, pixel)
    for seed, frame, pixel in paths[::7]:
        d = oracle.rng_u32(seed, frame, pixel, 48)
        for o in range(0, 44):                                              # ... looked for at every offset of every 7th path
            if o == 8:
                continue
            k = (int(d[o]), int(d[o + 1]), int(d[o + 2]), int(d[o + 3]))
            if k in seen and seen[k] != (seed, frame, pixel):
                clashes += 1
    assert clashes == 0
    assert len(seen) == len(paths)


@pytest.mark.parametrize("name", ["analytical_64x48_spp4_seed1", "analytical_32x24_spp16_seed7"])
def test_oracle_reproduces_golden_frames(oracle, name):
    import os
    want = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npy"))
    h, w = want.shape[:2]
    spp = int(name.split("spp")[1].split("_")[0])
    seed = int(name.split("seed")[1])
    got = oracle.render(oracle.scene_analytical(), w, h, spp, seed=seed)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_oracle_is_deterministic_and_thread_count_independent(oracle):
    d = oracle.scene_analytical()
    a = oracle.render(d, 48, 36, 3, seed=9, threads=1)
    b = oracle.render(d, 48, 36, 3, seed=9, threads=8)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    c = oracle.render(d, 48, 36, 3, seed=10, threads=8)
    assert not np.array_equal(a, c)


def test_oracle_row_range_and_resume(oracle):
    d = oracle.scene_analytical()
    full = oracle.render(d, 40, 30, 2, seed=1)
    part = oracle.render(d, 40, 30, 2, seed=1, rows=(10, 20))
    assert np.array_equal(part[10:20], full[10:20]) and np.all(part[:10] == 0) and np.all(part[20:] == 0)
    one = oracle.render(d, 40, 30, 1, seed=1)
    two = oracle.render(d, 40, 30, 1, seed=1, frames_done=1, pixels=one.copy())
    assert np.array_equal(two, full)


def test_convert_to_u8(oracle):
    px = np.array([[0.0, 1.0, 0.5, 1.0], [-1.0, 2.0, np.nan, np.inf], [0.2176, 0.0031, 1e-8, 0.999]], dtype=np.float32).reshape(1, 3, 4)
    out = oracle.convert_to_u8(px, 3, 1).reshape(3, 4)
    assert list(out[0]) == [0, 255, int((0.5 ** 0.4545) * 255), 255]
    assert list(out[1]) == [0, 255, 0, 255]                       # pow(-1, .4545) = NaN -> 0; saturating casts
    assert out[2, 0] == int((np.float64(np.float32(0.2176)) ** np.float64(np.float32(0.4545))) * 255.0)
    assert out[2, 3] == int(np.float32(0.999) * np.float32(255.0))


def test_sdf_object_known_answers(oracle, rpt):
    """The project-defined SDF object (include/rpt.h rpt_sdf): a single unit sphere, no smoothing
    partner, seen head-on from z = 3 must be hit at t = 2 like the analytical sphere (to the march
    tolerance hit_eps * t), with the analytical normal, so both scenes render nearly the same image."""
    def scene(use_sdf):
        s = rpt.Scene()
        s.camera = rpt.Pinhole((0.0, 0.0, 3.0), (0.0, 0.0, 0.0), 40.0)
        s.background = dict(kind=rpt._abi.RPT_BG_CONSTANT, colour_a=(0.3, 0.3, 0.3), colour_b=(0, 0, 0), gamma=2.2, scale=1.0)
        s.materials = [rpt.Material(rgb=(0.8, 0.8, 0.8), roughness=1.0)]
        s.lights = [rpt.AnalyticalLight.spherical((0.0, 4.0, 3.0), 0.5, (20.0, 20.0, 20.0))]
        if use_sdf:
            s.sdf = dict(prims=[(rpt._abi.RPT_SDF_SPHERE, (0.0, 0.0, 0.0), (1.0, 0.0))], material=0, smooth_k=0.25,
                         max_steps=128, hit_eps=1e-4, max_t=50.0, normal_eps=1e-3)
        else:
            s.spheres = [((0.0, 0.0, 0.0), 1.0, 0)]
        return s
    a = oracle.render(scene(True).describe(), 48, 48, 64, seed=4)[..., :3]
    b = oracle.render(scene(False).describe(), 48, 48, 64, seed=4)[..., :3]
    assert not np.isnan(a).any()
    assert abs(a.mean() - b.mean()) < 0.02 * b.mean()
    assert np.abs(a[16:32, 16:32].mean(axis=(0, 1)) - b[16:32, 16:32].mean(axis=(0, 1))).max() < 0.03
    # outside the silhouette both are the constant background
    assert np.allclose(a[0, 0], 0.3, atol=1e-6) and np.array_equal(a[0, 0], b[0, 0])
