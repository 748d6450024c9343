#!/usr/bin/env python3
"""tests/golden/reference_screenshot_200x150_u8.npy from the reference's README screenshot.

/root/reference/images/spheres.png is a macOS window capture (1824x1480) of the reference renderer:
the 800x600 frame shown at 2x inside a title bar and a black margin.  This script crops the frame
(rows 134..1333, columns 112..1711), box-filters it 8x8 to 200x150 and stores the gamma-encoded u8
RGB.  It is DATA derived from an image the reference ships (the only artefact of its output that
exists); it is used for a qualitative check only (tests/test_reference_screenshot.py).
Run in the dev container (the reference tree does not exist on the GPU box)."""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
im = np.asarray(Image.open("/root/reference/images/spheres.png").convert("RGB")).astype(np.float64)
crop = im[134:1334, 112:1712]
assert crop.shape == (1200, 1600, 3)
small = crop.reshape(150, 8, 200, 8, 3).mean(axis=(1, 3))
out = np.clip(np.round(small), 0, 255).astype(np.uint8)
np.save(os.path.join(ROOT, "tests", "golden", "reference_screenshot_200x150_u8.npy"), out)
print("wrote", out.shape, out.dtype)
