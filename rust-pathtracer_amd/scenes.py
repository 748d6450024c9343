"""Project-defined benchmark scenes (BASELINE.json configs 4-5; the reference has only AnalyticalScene).

Everything is generated from the project's PCG hash so the scenes are reproducible in any
language: u(i) = (pcg(seed + i) >> 8) * 2^-24."""
from . import _abi
from .api import AnalyticalLight, Material, Pinhole, Scene

_DEFAULTS = dict(rgb=(1.5, 1.5, 1.5), emission=(0.0, 0.0, 0.0), anisotropic=0.0, metallic=0.0, roughness=0.5,
                 subsurface=0.0, specular_tint=0.0, sheen=0.0, sheen_tint=0.0, clearcoat=0.0, clearcoat_gloss=0.0,
                 spec_trans=0.0, ior=1.45)                      # Material::new, material.rs:82-114


def full_material(medium=None, **fields):
    """A patch that sets every field (mask == RPT_MAT_ALL): what large scenes require for spheres.  `medium`: also the
    Medium (large scenes with media need it on every sphere material; dict(type="none") for none)."""
    d = dict(_DEFAULTS)
    d.update(fields)
    return Material(medium=medium, **d)


def pcg_hash(v):
    state = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


class _U:
    def __init__(self, seed):
        self.seed, self.i = seed & 0xFFFFFFFF, 0

    def __call__(self, lo=0.0, hi=1.0):
        u = (pcg_hash((self.seed + self.i) & 0xFFFFFFFF) >> 8) * 2.0 ** -24
        self.i += 1
        return lo + (hi - lo) * u


def media_scene():
    """Participating media (project-defined, include/rpt.h) in the reference's scene: the left sphere becomes a glass ball
    full of forward-scattering fog, the right one a ball of absorbing amber, a third small one glows (an emissive medium), and
    a small light sits inside the fog (visible from inside it: any_hit honours max_dist)."""
    from .api import AnalyticalScene
    s = AnalyticalScene()
    s.media = True
    s.any_hit_uses_max_dist = True
    s.max_depth = 12
    s.materials[0] = Material(rgb=(1.0, 1.0, 1.0), roughness=0.05, spec_trans=1.0, ior=1.2,
                              medium=dict(type="scatter", density=1.6, color=(0.95, 0.9, 0.8), anisotropy=0.6))
    s.materials[1] = Material(rgb=(1.0, 0.9, 0.7), roughness=0.1, spec_trans=1.0, ior=1.33,
                              medium=dict(type="absorb", density=1.2, color=(0.9, 0.55, 0.1)))
    s.materials.append(Material(rgb=(1.0, 1.0, 1.0), roughness=0.02, spec_trans=1.0, ior=1.05,
                                medium=dict(type="emissive", density=0.8, color=(0.2, 0.6, 1.0), anisotropy=2.0)))
    s.spheres.append(((0.0, -0.6, 1.2), 0.4, 3))
    s.lights.append(AnalyticalLight.spherical((-1.1, 0.0, 0.0), 0.12, (12.0, 12.0, 12.0)))
    return s


def random_spheres_scene(n_spheres=10000, n_lights=16, seed=0x5EED0005, n_palette=64, media=False):
    """spheres uniform in [-60,60]x[0,12]x[-120,0], radii U[0.3,1.2], a
    palette of full materials (rgb U[.05,1]^3, roughness U[.02,1], metallic 1 with p=.3, clearcoat 1
    with p=.2), a checker plane y=-1, spherical lights (r=1, emission 5) on a grid at y=15."""
    u = _U(seed)
    s = Scene()
    s.camera = Pinhole((0.0, 6.0, 14.0), (0.0, 2.0, -40.0), 70.0)
    s.background = dict(kind=_abi.RPT_BG_GRADIENT_Y, colour_a=(1.0, 1.0, 1.0), colour_b=(0.5, 0.7, 1.0), gamma=2.2, scale=0.5)
    s.any_hit_uses_max_dist = True
    s.media = media
    s.materials = []
    for _ in range(n_palette):
        rgb = (u(0.05, 1.0), u(0.05, 1.0), u(0.05, 1.0))
        rough = u(0.02, 1.0)
        metallic = 1.0 if u() < 0.3 else 0.0
        coat = 1.0 if u() < 0.2 else 0.0
        medium = None
        if media:                                     # a third of the palette: glass full of fog / coloured absorber / glow
            kind = (len(s.materials) % 6)
            medium = (dict(type="scatter", density=0.9, color=rgb, anisotropy=0.5), dict(type="absorb", density=1.5, color=rgb),
                      dict(type="emissive", density=0.3, color=rgb), dict(type="none"), dict(type="none"), dict(type="none"))[kind]
            if kind < 3:
                s.materials.append(full_material(medium=medium, rgb=(1.0, 1.0, 1.0), roughness=0.05, spec_trans=1.0, ior=1.3))
                continue
        s.materials.append(full_material(medium=medium, rgb=rgb, roughness=rough, metallic=metallic, clearcoat=coat, clearcoat_gloss=coat))
    s.materials.append(Material(roughness=1.0, checker_dir=(0.5, 100.0, 0.25, 0.1)))     # the reference's floor
    floor = len(s.materials) - 1
    s.spheres = []
    for _ in range(n_spheres):
        c = (u(-60.0, 60.0), u(0.0, 12.0), u(-120.0, 0.0))
        r = u(0.3, 1.2)
        s.spheres.append((c, r, int(u() * n_palette) % n_palette))
    # The floor reaches 400 units along a ray: beyond that the f32 ray/sphere test of the reference is
    # noise (d2 = l.l - tca^2 cancels), which would force brute-force tests for rays that start there.
    s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, floor, 400.0)]
    side = max(1, int(round(n_lights ** 0.5)))
    s.lights = []
    for i in range(n_lights):
        gx, gz = i % side, i // side
        x = -45.0 + 90.0 * (gx + 0.5) / side
        z = -105.0 + 90.0 * (gz + 0.5) / side
        s.lights.append(AnalyticalLight.spherical((x, 15.0, z), 1.0, (5.0, 5.0, 5.0)))
    return s


def six_primitive_scene():
    """Five spheres with whole materials on the reference's checker floor: a small scene of more than four primitives — what
    bench.py's `six_primitives` leg renders (the material table by class of accepted set, csrc/launch.h MatClassMap) and
    tests/test_gpu_dispatch.py checks against the oracle ("five spheres on a floor")."""
    from .api import AnalyticalScene
    s = AnalyticalScene()
    s.materials = [full_material(rgb=(0.9, 0.3, 0.2), clearcoat=1.0, clearcoat_gloss=0.7, roughness=0.4),
                   full_material(rgb=(0.8, 0.8, 0.9), roughness=0.15, metallic=1.0, anisotropic=0.6),
                   full_material(rgb=(0.95, 0.95, 1.0), roughness=0.05, spec_trans=1.0, ior=1.5),
                   full_material(rgb=(0.2, 0.7, 0.3), roughness=0.6, sheen=0.8, subsurface=0.4),
                   full_material(rgb=(0.9, 0.8, 0.1), roughness=0.3, metallic=1.0),
                   Material(roughness=1.0, checker_dir=(0.5, 100.0, 0.25, 0.1))]
    s.spheres = [((0.2, 0.0, -0.6), 1.0, 0), ((-1.3, -0.3, 0.4), 0.7, 1), ((1.4, -0.4, 0.5), 0.6, 2), ((-0.4, -0.6, 1.1), 0.4, 3), ((0.6, -0.65, 1.3), 0.35, 4)]
    s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 5)]
    return s


def sdf_scene():
    """BASELINE.json configs[3] (project-defined; the reference has no SDF scene, Readme.md:18): a
    sphere-marched blob — the polynomial smooth union of two spheres and a torus — over the reference's
    checker plane, next to one analytical clearcoat sphere, lit by the reference's spherical light."""
    s = Scene()
    s.camera = Pinhole((0.0, 0.6, 3.6), (0.0, 0.0, 0.0), 70.0)
    s.background = dict(kind=_abi.RPT_BG_GRADIENT_Y, colour_a=(1.0, 1.0, 1.0), colour_b=(0.5, 0.7, 1.0), gamma=2.2, scale=0.5)
    s.lights = [AnalyticalLight.spherical((3.0, 2.0, 2.0), 1.0, (3.0, 3.0, 3.0))]
    s.materials = [
        Material(rgb=(0.2, 0.55, 0.9), roughness=0.35, clearcoat=1.0, clearcoat_gloss=0.8),      # the blob
        Material(rgb=(1.0, 0.186, 0.0), clearcoat=1.0, clearcoat_gloss=1.0, roughness=0.1),      # analytical sphere
        Material(roughness=1.0, checker_dir=(0.5, 100.0, 0.25, 0.1)),                           # floor
    ]
    s.spheres = [((1.9, -0.4, -0.3), 0.6, 1)]
    s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 2)]
    s.sdf = dict(prims=[(_abi.RPT_SDF_SPHERE, (-0.9, -0.2, 0.0), (0.8, 0.0)),
                        (_abi.RPT_SDF_SPHERE, (0.1, 0.15, 0.2), (0.55, 0.0)),
                        (_abi.RPT_SDF_TORUS_Y, (-0.3, -0.55, 0.1), (1.25, 0.22))],
                 material=0, smooth_k=0.35, max_steps=128, hit_eps=1e-3, max_t=60.0, normal_eps=1e-3)
    return s
