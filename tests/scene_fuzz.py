"""Random SMALL scenes for differential tests of the render kernels (tests/test_gpu_range_guards.py, tools/range_soak.py): 1-8 spheres with
random full materials (metal, clearcoat, glass, sheen, subsurface, anisotropy), 1-4 spherical lights, the reference's checker floor,
depth 1-8, and EVERY LENGTH SCALED by a power of two between 2^-33 and 2^33 — so that none, some or all samples leave the range the
library's short divide / square root are proven for (csrc/dev_math.h)."""
import numpy as np


def random_small_scene(rpt, seed, n_spheres=None, n_lights=None):
    """-> (scene, log2 of its scale, render flags, rng).  n_spheres / n_lights: fixed table sizes (the reference's are 2 and 1: scenes that
    take the kernels which know those sizes, kernels.hip sized_scene); None: random."""
    from rust_pathtracer_amd import scenes
    from rust_pathtracer_amd.api import Pinhole, Scene
    A = rpt._abi
    rng = np.random.default_rng(seed)
    log2_k = int(rng.integers(-33, 34))
    k = float(2.0 ** log2_k)
    f = lambda x: float(np.float32(x) * np.float32(k))      # noqa: E731
    s = Scene()
    s.camera = Pinhole((f(rng.uniform(-1, 1)), f(rng.uniform(0.5, 2)), f(rng.uniform(3, 5))), (0.0, 0.0, 0.0), float(rng.uniform(50, 90)))
    s.background = dict(kind=A.RPT_BG_GRADIENT_Y, colour_a=(1.0, 1.0, 1.0), colour_b=(0.5, 0.7, 1.0), gamma=2.2, scale=0.5)
    s.materials = []
    for _ in range(int(rng.integers(1, 9))):
        glass = rng.random() < 0.25
        s.materials.append(scenes.full_material(rgb=tuple(float(x) for x in rng.uniform(0.0, 1.0, 3)), roughness=float(rng.uniform(0.0, 1.0)),
                                                metallic=float(rng.random() < 0.3), clearcoat=float(rng.random() < 0.3), clearcoat_gloss=float(rng.uniform(0, 1)),
                                                spec_trans=1.0 if glass else 0.0, ior=float(rng.uniform(1.1, 1.8)), anisotropic=float(rng.uniform(0, 1)),
                                                sheen=float(rng.uniform(0, 1)), subsurface=float(rng.uniform(0, 1))))
    s.materials.append(rpt.Material(roughness=1.0, checker_dir=(0.5, 100.0, 0.25, 0.1)))
    s.spheres = [((f(rng.uniform(-2.5, 2.5)), f(rng.uniform(-0.5, 1.5)), f(rng.uniform(-2.5, 1.0))), f(rng.uniform(0.3, 1.0)), int(rng.integers(0, len(s.materials) - 1)))
                 for _ in range(n_spheres if n_spheres else int(rng.integers(1, 9)))]
    s.planes = [((0.0, 1.0, 0.0), (0.0, f(-1.0), 0.0), 0.0001, len(s.materials) - 1)]
    s.lights = [rpt.AnalyticalLight.spherical((f(rng.uniform(-4, 4)), f(rng.uniform(2, 5)), f(rng.uniform(-2, 4))), f(rng.uniform(0.3, 1.2)),
                                              tuple(float(x) for x in rng.uniform(1.0, 6.0, 3))) for _ in range(n_lights if n_lights else int(rng.integers(1, 5)))]
    s.eps = f(0.005)
    s.max_depth = int(rng.integers(1, 9))
    s.any_hit_uses_max_dist = bool(rng.random() < 0.5)
    flags = A.RPT_RENDER_RUSSIAN_ROULETTE if rng.random() < 0.4 else 0
    return s, log2_k, flags, rng
