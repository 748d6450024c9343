"""The Python mirror of rust/gpu_tracer.rs's scene adapter — `SceneDescBuilder`, `RptMaterial::patch / full / with_medium /
with_checker_dir / zero_unmasked`, `From<&AnalyticalLight> for RptLight` — method for method, and of rust/analytical_gpu.rs's
`impl GpuScene for AnalyticalScene` statement for statement.  The Rust files cannot be compiled here (no rustc); this mirror is what
tests/test_rust_binding.py runs: the descriptor it builds for the stock scene is byte-identical to the library's
`rpt_scene_analytical`, and its method list and the literals of `analytical_describe` are compared with the Rust sources
mechanically.  Test scaffolding: the product path never imports it (api.Scene.describe is the package's own builder)."""
import ctypes as C
import math

import numpy as np

from rust_pathtracer_amd import _abi


class RefMaterial:
    """`Material::new()` (material.rs:82-114): the reference-side value a patch is taken from."""

    def __init__(self):
        self.rgb = (1.5, 1.5, 1.5)
        self.emission = (0.0, 0.0, 0.0)
        self.anisotropic = 0.0
        self.metallic = 0.0
        self.roughness = 0.5
        self.subsurface = 0.0
        self.specular_tint = 0.0
        self.sheen = 0.0
        self.sheen_tint = 0.0
        self.clearcoat = 0.0
        self.clearcoat_gloss = 0.0
        self.spec_trans = 0.0
        self.ior = 1.45
        self.medium = dict(medium_type=_abi.RPT_MEDIUM_NONE, density=0.0, color=(0.0, 0.0, 0.0), anisotropy=0.0)   # Medium::new(), material.rs:24-34


class RefLight:
    """`AnalyticalLight::spherical` (light.rs:13-28): `.light` is the `Light` of globals.rs:76-84."""

    def __init__(self, position, radius, emission):
        r = np.float32(radius)
        self.light = dict(light_type=_abi.RPT_LIGHT_SPHERICAL, position=tuple(position), emission=tuple(emission), u=(0.0, 0.0, 0.0), v=(0.0, 0.0, 0.0),
                          radius=float(r), area=float(np.float32(4.0) * np.float32(math.pi) * r * r))


def light_record(l):
    """`impl From<&AnalyticalLight> for RptLight`."""
    L = l.light
    out = _abi.rpt_light()
    out.type = L["light_type"]
    out.position = _abi.F3(*L["position"]); out.emission = _abi.F3(*L["emission"])
    out.u = _abi.F3(*L["u"]); out.v = _abi.F3(*L["v"])
    out.radius = L["radius"]; out.area = L["area"]
    return out


_SCALARS = (("anisotropic", _abi.RPT_MAT_ANISOTROPIC), ("metallic", _abi.RPT_MAT_METALLIC), ("roughness", _abi.RPT_MAT_ROUGHNESS),
            ("subsurface", _abi.RPT_MAT_SUBSURFACE), ("specular_tint", _abi.RPT_MAT_SPECULAR_TINT), ("sheen", _abi.RPT_MAT_SHEEN),
            ("sheen_tint", _abi.RPT_MAT_SHEEN_TINT), ("clearcoat", _abi.RPT_MAT_CLEARCOAT), ("clearcoat_gloss", _abi.RPT_MAT_CLEARCOAT_GLOSS),
            ("spec_trans", _abi.RPT_MAT_SPEC_TRANS), ("ior", _abi.RPT_MAT_IOR))


class RptMaterial:
    """The methods of `impl RptMaterial` (rust/gpu_tracer.rs), on a ctypes rpt_material."""

    def __init__(self, c):
        self.c = c

    @staticmethod
    def patch(m, mask):
        c = _abi.rpt_material()
        c.mask = mask
        c.proc_kind = _abi.RPT_PROC_NONE
        c.rgb = _abi.F3(*m.rgb); c.emission = _abi.F3(*m.emission)
        for name, _ in _SCALARS:
            setattr(c, name, getattr(m, name))
        c.medium_type = m.medium["medium_type"]; c.medium_density = m.medium["density"]
        c.medium_color = _abi.F3(*m.medium["color"]); c.medium_anisotropy = m.medium["anisotropy"]
        return RptMaterial(c)

    @staticmethod
    def full(m):
        return RptMaterial.patch(m, _abi.RPT_MAT_ALL)

    def with_medium(self):
        self.c.mask |= _abi.RPT_MAT_MEDIUM
        return self

    def with_checker_dir(self, scale, offset, a, b):
        self.c.proc_kind = _abi.RPT_PROC_CHECKER_DIR
        self.c.proc_params = _abi.F4(scale, offset, a, b)
        return self

    def zero_unmasked(self):
        c, out = self.c, _abi.rpt_material()
        out.mask, out.proc_kind = c.mask, c.proc_kind
        if c.mask & _abi.RPT_MAT_RGB:
            out.rgb = c.rgb
        if c.mask & _abi.RPT_MAT_EMISSION:
            out.emission = c.emission
        for name, bit in _SCALARS:
            if c.mask & bit:
                setattr(out, name, getattr(c, name))
        out.proc_params = c.proc_params
        if c.mask & _abi.RPT_MAT_MEDIUM:
            out.medium_type, out.medium_density, out.medium_color, out.medium_anisotropy = c.medium_type, c.medium_density, c.medium_color, c.medium_anisotropy
        return RptMaterial(out)


class SceneDescBuilder:
    """`SceneDescBuilder` of rust/gpu_tracer.rs: owns the tables, lends a descriptor."""

    def __init__(self):                                              # `new()`
        self.flags = 0
        self.camera_ = ((0.0, 0.0, 3.0), (0.0, 0.0, 0.0), 80.0)
        self.background = (_abi.RPT_BG_CONSTANT, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 1.0, 1.0)
        self.eps = 0.005
        self.max_depth_ = 4
        self.spheres, self.planes, self.lights, self.materials, self.sdf_prims = [], [], [], [], []

    def camera(self, origin, center, fov_deg):
        self.camera_ = (tuple(origin), tuple(center), fov_deg)
        return self

    def background_gradient_y(self, a, b, gamma, scale):
        self.background = (_abi.RPT_BG_GRADIENT_Y, tuple(a), tuple(b), gamma, scale)
        return self

    def background_constant(self, c, scale):
        self.background = (_abi.RPT_BG_CONSTANT, tuple(c), (0.0, 0.0, 0.0), 1.0, scale)
        return self

    def max_depth(self, depth):
        self.max_depth_ = int(depth)
        return self

    def set_flags(self, flags):                                      # `flags()` (the attribute has the name here)
        self.flags = flags
        return self

    def material(self, m):
        self.materials.append(m)
        return len(self.materials) - 1

    def sphere(self, center, radius, material):
        self.spheres.append((tuple(center), radius, material))
        return self

    def plane(self, normal, point, min_denom, material, max_t):
        self.planes.append((tuple(normal), tuple(point), min_denom, material, max_t))
        return self

    def light(self, l):
        self.lights.append(light_record(l))
        return self

    def lights_of(self, scene):
        for i in range(scene.number_of_lights()):
            self.lights.append(light_record(scene.light_at(i)))
        return self

    def with_desc(self, f):
        d = _abi.rpt_scene_desc()
        d.abi_version = _abi.RPT_ABI_VERSION
        d.flags = self.flags
        d.camera.origin = _abi.F3(*self.camera_[0]); d.camera.center = _abi.F3(*self.camera_[1]); d.camera.fov_deg = self.camera_[2]
        d.background.kind = self.background[0]
        d.background.colour_a = _abi.F3(*self.background[1]); d.background.colour_b = _abi.F3(*self.background[2])
        d.background.gamma, d.background.scale = self.background[3], self.background[4]
        d.eps, d.max_depth = self.eps, self.max_depth_
        sph = (_abi.rpt_sphere * max(1, len(self.spheres)))()
        for i, (c, r, m) in enumerate(self.spheres):
            sph[i].center = _abi.F3(*c); sph[i].radius = r; sph[i].material = m
        pl = (_abi.rpt_plane * max(1, len(self.planes)))()
        for i, (n, p, md, m, mt) in enumerate(self.planes):
            pl[i].normal = _abi.F3(*n); pl[i].point = _abi.F3(*p); pl[i].min_denom = md; pl[i].material = m; pl[i].max_t = mt
        li = (_abi.rpt_light * max(1, len(self.lights)))(*self.lights)
        ma = (_abi.rpt_material * max(1, len(self.materials)))(*[m.c for m in self.materials])
        d.n_spheres = len(self.spheres); d.spheres = C.cast(sph, C.POINTER(_abi.rpt_sphere)) if self.spheres else None
        d.n_planes = len(self.planes); d.planes = C.cast(pl, C.POINTER(_abi.rpt_plane)) if self.planes else None
        d.n_lights = len(self.lights); d.lights = C.cast(li, C.POINTER(_abi.rpt_light)) if self.lights else None
        d.n_materials = len(self.materials); d.materials = C.cast(ma, C.POINTER(_abi.rpt_material)) if self.materials else None
        return f(d)


class RefAnalyticalScene:
    """What rust/analytical_gpu.rs reads of `AnalyticalScene` through the `Scene` trait (analytical.rs:13-22, 149-155; scene.rs:28-30)."""

    def __init__(self):
        em = 3.0
        self.lights = [RefLight((3.0, 2.0, 2.0), 1.0, (em, em, em))]

    def number_of_lights(self):
        return len(self.lights)

    def light_at(self, index):
        return self.lights[index]

    def recursion_depth(self):
        return 4


def analytical_describe(self):
    """rust/analytical_gpu.rs, `impl GpuScene for AnalyticalScene`, statement for statement."""
    b = SceneDescBuilder()
    b.max_depth(self.recursion_depth())
    b.background_gradient_y((1.0, 1.0, 1.0), (0.5, 0.7, 1.0), 2.2, 0.5)
    left = RefMaterial()
    left.rgb = (1.0, 1.0, 1.0)
    left.roughness = 0.05
    left.metallic = 1.0
    left = b.material(RptMaterial.patch(left, _abi.RPT_MAT_RGB | _abi.RPT_MAT_ROUGHNESS | _abi.RPT_MAT_METALLIC))
    right = RefMaterial()
    right.rgb = (1.0, 0.186, 0.0)
    right.clearcoat = 1.0
    right.clearcoat_gloss = 1.0
    right.roughness = 0.1
    right = b.material(RptMaterial.patch(right, _abi.RPT_MAT_RGB | _abi.RPT_MAT_CLEARCOAT | _abi.RPT_MAT_CLEARCOAT_GLOSS | _abi.RPT_MAT_ROUGHNESS))
    floor = RefMaterial()
    floor.roughness = 1.0
    floor = b.material(RptMaterial.patch(floor, _abi.RPT_MAT_ROUGHNESS).with_checker_dir(0.5, 100.0, 0.25, 0.1))
    b.materials = [m.zero_unmasked() for m in b.materials]
    b.sphere((-1.1, 0.0, 0.0), 1.0, left)
    b.sphere((1.1, 0.0, 0.0), 1.0, right)
    b.plane((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, floor, 0.0)
    b.lights_of(self)
    return b
