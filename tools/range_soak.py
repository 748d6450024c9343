"""Differential soak of the range trackers (csrc/dev_math.h, RPT_MATH_MODE 2) against the tests per operation: random SMALL scenes — 1-8
spheres with random full materials (metal, clearcoat, glass), 1-4 lights, depth 1-8, with and without roulette, every length scaled
by a random power of two between 2^-33 and 2^33 so that none / some / all samples leave the short sequences' range — rendered by the
library RPT_LIB names, one hash per scene.  Run it once with the shipped library and once with the `perop` variant of tools/build_variants.py (per-operation tests
throughout: -DRPT_GUARD_PER_OP) and compare the outputs: they must be identical.   python tools/range_soak.py [n_scenes] [first_seed] [ref]
`ref`: every scene has the reference scene's table sizes (2 spheres, 1 plane, 1 light) and takes the kernels that know them — run once
as is and once with RPT_NO_SIZED_KERNELS=1."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest  # noqa: E402
import torch  # noqa: E402

rpt = conftest.load_package()
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ref_sizes = len(sys.argv) > 3 and sys.argv[3] == "ref"

from scene_fuzz import random_small_scene  # noqa: E402  (tests/)

for seed in range(first, first + n_scenes):
    s, log2_k, flags, rng = random_small_scene(rpt, seed, 2, 1) if ref_sizes else random_small_scene(rpt, seed)
    w, h = int(rng.integers(200, 700)), int(rng.integers(120, 400))
    steps = [int(x) for x in rng.integers(1, 12, size=int(rng.integers(1, 4)))]
    t = rpt.Tracer(s, device=0, seed=seed)
    t.flags = flags
    buf = rpt.DeviceColorBuffer(w, h)
    for n in steps:
        t.render_n(buf, n)
    torch.cuda.synchronize()
    img = buf.pixels.cpu().numpy()
    t.close()
    print("scene %3d  x 2^%-3d %4dx%-4d steps %-12s depth %d  nan pixels %5d  %s" % (
        seed, log2_k, w, h, steps, s.max_depth, int(np.isnan(img).any(axis=2).sum()), hashlib.sha1(img.tobytes()).hexdigest()[:16]), flush=True)
