"""The only artefact of the REAL reference's output: its README screenshot (images/spheres.png, a
window capture of the 800x600 frame).  tests/golden/reference_screenshot_200x150_u8.npy is that frame
cropped and box-filtered (tools/make_screenshot_fixture.py).  The oracle's render of the same scene
must agree with it closely — a statistical pin on the oracle against the Rust binary.  It stays "parity
unpinned": the capture's sample count is unknown and its colours went through the window system.

What the window system did is separated from what the tracer did: the SKY of the frame is
analytical.rs:28-32 — a gradient with no sampling in it — so a colour matrix fitted on sky rows alone
(white-preserving: rows sum to 1, six free numbers) is the capture's colour management and nothing else;
applied to the whole oracle frame it leaves per region what the tracers disagree on.  CPU only."""
import os

import numpy as np

GAMMA = 0.4545                                                    # ColorBuffer::convert_to_u8, buffer.rs:59
REGIONS = {                                                       # rows, columns of the 200x150 frame
    "left sphere": (slice(60, 100), slice(45, 85)),               # metal, roughness .05: reflects sky and floor
    "right sphere": (slice(60, 100), slice(115, 155)),            # orange, clearcoat
    "far floor": (slice(100, 118), slice(0, 200)),
    "near floor": (slice(125, 150), slice(0, 200)),
}


def _frames(oracle):
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_screenshot_200x150_u8.npy")).astype(np.float64) / 255.0
    img = oracle.render(oracle.scene_analytical(), 200, 150, 256, seed=1)[..., :3].astype(np.float64)
    return ref, np.clip(img, 0.0, 1.0)


def _sky_matrix(ref, lin):
    """Least squares on the sky rows, in linear light: ref_lin ~ M lin with every row of M summing to 1."""
    sky = (slice(0, 40), slice(0, 200))
    X = lin[sky].reshape(-1, 3)
    Y = (ref[sky] ** (1.0 / GAMMA)).reshape(-1, 3)
    B = np.stack([X[:, 0] - X[:, 1], X[:, 1] - X[:, 2]], axis=1)
    M = np.eye(3)
    for c in range(3):
        ab = np.linalg.lstsq(B, Y[:, c] - X[:, c], rcond=None)[0]
        M[c] += ab[0] * np.array([1.0, -1.0, 0.0]) + ab[1] * np.array([0.0, 1.0, -1.0])
    return M


def test_oracle_matches_the_reference_screenshot(oracle):
    ref, lin = _frames(oracle)
    mine = lin ** GAMMA
    for c in range(3):
        corr = np.corrcoef(ref[..., c].ravel(), mine[..., c].ravel())[0, 1]
        assert corr > 0.98, (c, corr)
    d = np.abs(ref - mine)
    assert d.mean() < 0.025
    assert np.percentile(d, 99) < 0.10
    assert np.abs(ref.mean(axis=(0, 1)) - mine.mean(axis=(0, 1))).max() < 0.015
    # structure: sky above, the two spheres left/right of centre, checker floor below
    assert mine[10, 100, 2] > mine[10, 100, 0]                   # blue sky
    assert mine[75, 150, 0] > 2 * mine[75, 150, 2]               # the orange clearcoat sphere


def test_regions_agree_once_the_captures_colour_management_is_taken_out(oracle):
    ref, lin = _frames(oracle)
    M = _sky_matrix(ref, lin)
    # a mild desaturation (the capture's display profile), nothing wild: |M - I| small, rows sum to 1
    assert np.allclose(M.sum(axis=1), 1.0) and np.abs(M - np.eye(3)).max() < 0.25, M
    adj = np.clip(lin @ M.T, 0.0, None) ** GAMMA
    sky = adj[:40] - ref[:40]
    assert np.abs(sky).mean() < 0.004, "the matrix does not even explain the sky"
    # Before the matrix the sky — no Monte Carlo in it — is off by 0.017 / 0.005 / 0.012 (R / G / B, gamma space): that much of
    # the old tolerance was the window system's.
    assert np.abs((lin[:40] ** GAMMA - ref[:40]).mean(axis=(0, 1))).max() > 0.01
    for name, (ys, xs) in REGIONS.items():
        d = adj[ys, xs] - ref[ys, xs]
        mean_abs = np.abs(d).mean()
        signed = d.mean(axis=(0, 1))
        # mean |d| carries the capture's own noise and the box filter's edges (checker, silhouettes); the SIGNED mean is the
        # energy of the region and is what a mis-weighted lobe or light would move
        assert mean_abs < (0.014 if name == "right sphere" else 0.011), (name, mean_abs)
        if name == "right sphere":
            # orange (1, .186, 0): its blue is ~0 in linear light and outside the span of the sky colours the matrix was
            # identified from, so blue (and a little of green) is extrapolation; red is not
            assert abs(signed[0]) < 0.004 and abs(signed[1]) < 0.007 and abs(signed[2]) < 0.018, (name, signed)
        else:
            assert np.abs(signed).max() < 0.005, (name, signed)
        # what the bound is worth: 5 % more energy in the region would break it
        brighter = np.clip(lin[ys, xs] * 1.05 @ M.T, 0.0, None) ** GAMMA - ref[ys, xs]
        assert np.abs(brighter.mean(axis=(0, 1))).max() > (0.0045 if name != "right sphere" else 0.0045), (name, brighter.mean(axis=(0, 1)))


# ---- per object (VERDICT r5, next #8c): statistics a wrong light, lobe, material or shadow would move --------------------------
def _render_lin(oracle, scene, seed=1):
    return np.clip(oracle.render(scene.describe(), 200, 150, 256, seed=seed)[..., :3].astype(np.float64), 0.0, 1.0)


def _object_stats(img, faithful, shadow):
    """Four numbers per frame, each tied to one object of renderer/src/analytical.rs; pixel sets are chosen on the FAITHFUL oracle frame.
      metal     mean colour of the brightest 5 % of the left sphere's box: the sky and the light mirrored in the metal (roughness .05)
      coat      the same for the right sphere: the light's reflection in the clearcoat over the orange base
      checker   mean of the near floor's brighter half minus its darker half: the contrast of analytical.rs:107-116's 0.25 / 0.1
      shadow    mean colour where the spheres shadow the floor (the set where a frame WITHOUT the spheres is much brighter)"""
    out = {}
    for name, key in (("metal", "left sphere"), ("coat", "right sphere")):
        reg = REGIONS[key]
        lum = faithful[reg].mean(axis=-1)
        out[name] = img[reg][lum >= np.percentile(lum, 95)].mean(axis=0)
    reg = REGIONS["near floor"]
    lum = faithful[reg].mean(axis=-1)
    v = img[reg].mean(axis=-1)
    out["checker"] = np.array([v[lum > np.median(lum)].mean() - v[lum <= np.median(lum)].mean()])
    out["shadow"] = img[shadow].mean(axis=0)
    return out


# what each statistic may differ by between the reference's capture and the oracle: ~2x what another seed moves it (0.004 on the
# highlights at 256 spp) plus the capture's own colour management; the coat's blue is extrapolated by the sky matrix (see above)
OBJECT_TOL = {"metal": (0.008, 0.008, 0.008), "coat": (0.008, 0.008, 0.03), "checker": (0.008,), "shadow": (0.008, 0.008, 0.01)}


def _violations(stats, ref_stats):
    return [(k, c) for k in OBJECT_TOL for c, tol in enumerate(OBJECT_TOL[k]) if abs(stats[k][c] - ref_stats[k][c]) > tol]


def test_objects_agree_and_wrong_scenes_would_not(rpt, oracle):
    """Per-object statistics of the oracle frame against the reference's screenshot — and, so that the bounds mean something, the
    same statistics for scenes that are WRONG in one respect each: every one of them must break a bound.  What the capture cannot
    tell apart is stated too, with its reason asserted: quirk Q3 (any_hit ignoring max_dist, analytical.rs:130) changes no pixel of
    this view — no shadow ray of a visible point meets geometry beyond the light — and quirk Q5 (log2 in GTR1, tracer.rs:239) moves
    the clearcoat highlight by less than another seed does.  Those two stay pinned by citation only."""
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_screenshot_200x150_u8.npy")).astype(np.float64) / 255.0
    lin = _render_lin(oracle, rpt.AnalyticalScene())
    M = _sky_matrix(ref, lin)
    adj = lambda x: np.clip(x @ M.T, 0.0, None) ** GAMMA      # noqa: E731

    def variant(edit):
        s = rpt.AnalyticalScene()
        edit(s)
        return adj(_render_lin(oracle, s))

    faithful = adj(lin)
    bare = variant(lambda s: setattr(s, "spheres", []))                     # the floor under the same light, nothing to shadow it
    rows = np.arange(150)[:, None] * np.ones((1, 200), dtype=int)
    shadow = (bare.mean(axis=-1) - faithful.mean(axis=-1) > 0.08) & (rows > 100)
    assert 1500 < shadow.sum() < 3500
    ref_stats = _object_stats(ref, faithful, shadow)
    assert _violations(_object_stats(faithful, faithful, shadow), ref_stats) == []
    # another seed of the faithful scene stays inside as well (the bounds are not a fit to seed 1's noise)
    assert _violations(_object_stats(adj(_render_lin(oracle, rpt.AnalyticalScene(), seed=2)), faithful, shadow), ref_stats) == []

    wrong = {
        "light 10 % brighter (analytical.rs:15-16)": lambda s: setattr(s, "lights", [rpt.AnalyticalLight.spherical((3.0, 2.0, 2.0), 1.0, (3.3, 3.3, 3.3))]),
        "metal sphere at roughness 0.2 (analytical.rs:57: 0.05)": lambda s: s.materials.__setitem__(0, rpt.Material(rgb=(1.0, 1.0, 1.0), roughness=0.2, metallic=1.0)),
        "left sphere not metallic (analytical.rs:58)": lambda s: s.materials.__setitem__(0, rpt.Material(rgb=(1.0, 1.0, 1.0), roughness=0.05)),
        "checker 0.2 / 0.15 (analytical.rs:113-115: 0.25 / 0.1)": lambda s: s.materials.__setitem__(2, rpt.Material(roughness=1.0, checker_dir=(0.5, 100.0, 0.2, 0.15))),
        "orange sphere at roughness 0.5, no gloss (analytical.rs:83-85)": lambda s: s.materials.__setitem__(1, rpt.Material(rgb=(1.0, 0.186, 0.0), clearcoat=1.0, clearcoat_gloss=0.0, roughness=0.5)),
        "no occluders: nothing shadows the floor (analytical.rs:130-145)": lambda s: setattr(s, "spheres", []),
    }
    for what, edit in wrong.items():
        broken = _violations(_object_stats(variant(edit), faithful, shadow), ref_stats)
        assert broken, "the screenshot comparison would not notice: " + what

    # what it cannot see, and why
    d = oracle.scene_analytical()
    d.flags |= rpt._abi.RPT_SCENE_ANYHIT_USES_MAX_DIST
    q3 = oracle.render(d, 200, 150, 64, seed=1)
    assert np.array_equal(q3.view(np.uint32), oracle.render(oracle.scene_analytical(), 200, 150, 64, seed=1).view(np.uint32)), \
        "Q3 does change this view: pin it with the screenshot"
    oracle.lib.oracle_undo_quirks(1)
    try:
        q5 = _object_stats(adj(_render_lin(oracle, rpt.AnalyticalScene())), faithful, shadow)
    finally:
        oracle.lib.oracle_undo_quirks(0)
    f = _object_stats(faithful, faithful, shadow)
    assert max(np.abs(q5[k] - f[k]).max() for k in OBJECT_TOL) < 0.002, "Q5 does move an object's statistic: pin it with the screenshot"
