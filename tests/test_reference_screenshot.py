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
