"""The only artefact of the REAL reference's output: its README screenshot (images/spheres.png, a
window capture of the 800x600 frame).  tests/golden/reference_screenshot_200x150_u8.npy is that frame
cropped and box-filtered (tools/make_screenshot_fixture.py).  The oracle's render of the same scene
must agree with it closely — a statistical pin on the oracle against the Rust binary (the capture's
sample count and colour management are unknown, so this cannot be exact).  CPU only."""
import os

import numpy as np


def test_oracle_matches_the_reference_screenshot(oracle):
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_screenshot_200x150_u8.npy")).astype(np.float64) / 255.0
    img = oracle.render(oracle.scene_analytical(), 200, 150, 96, seed=1)[..., :3].astype(np.float64)
    mine = np.clip(img, 0.0, 1.0) ** 0.4545                      # ColorBuffer::convert_to_u8's gamma, buffer.rs:59
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
