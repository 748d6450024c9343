"""Host-side mirror of the reference's API for the render path, over the C ABI.

Names and argument meaning follow the reference so code reads the same:
    reference (Rust)                                   here
    ------------------------------------------------   ---------------------------------
    ColorBuffer::new(w, h)       buffer.rs:18          ColorBuffer(w, h) / DeviceColorBuffer(w, h)
    Tracer::new(scene)           tracer.rs:13          Tracer(scene)
    Tracer::render(&mut buffer)  tracer.rs:22          Tracer.render(buffer)     (1 spp, frames += 1)
    Tracer::scene()              tracer.rs:629         Tracer.scene()
    AnalyticalScene::new()       analytical.rs:13      AnalyticalScene()
    Pinhole::new/set/set_fov     pinhole.rs:14-34      Pinhole(...)
    AnalyticalLight::spherical   light.rs:13           AnalyticalLight.spherical(...)
    buffer.convert_to_u8(frame)  buffer.rs:55          buffer.convert_to_u8()
All computation happens in the HIP library; nothing here falls back to the CPU.
"""
import ctypes as C
import math

import numpy as np

from . import _abi
from ._lib import RptError, check, lib


class Pinhole:
    """camera/pinhole.rs:6-34"""

    def __init__(self, origin=(0.0, 0.0, 3.0), center=(0.0, 0.0, 0.0), fov=80.0):
        self.origin = tuple(origin)
        self.center = tuple(center)
        self.fov = float(fov)

    def set(self, origin, center):
        self.origin = tuple(origin)
        self.center = tuple(center)

    def set_fov(self, fov):
        self.fov = float(fov)


class AnalyticalLight:
    """light.rs:6-28 (globals.rs:76-84 for the fields)"""

    def __init__(self, light_type, position, emission, radius, area, u=(0.0, 0.0, 0.0), v=(0.0, 0.0, 0.0)):
        self.light_type = light_type
        self.position = tuple(position)
        self.emission = tuple(emission)
        self.radius = float(radius)
        self.area = float(area)
        self.u = tuple(u)
        self.v = tuple(v)

    @staticmethod
    def spherical(position, radius, emission):
        r = np.float32(radius)
        area = np.float32(4.0) * np.float32(math.pi) * r * r          # light.rs:22, f32 left to right
        return AnalyticalLight(_abi.RPT_LIGHT_SPHERICAL, position, emission, radius, float(area))

    @staticmethod
    def rectangular(position, u, v, emission):
        """LightType::Rectangular (globals.rs:70): the parallelogram position + a*u + b*v.  The reference declares the type
        and has no constructor or sampling code for it; sampled only in scenes with sample_all_light_types (include/rpt.h)."""
        c = np.cross(np.asarray(u, dtype=np.float64), np.asarray(v, dtype=np.float64))
        return AnalyticalLight(_abi.RPT_LIGHT_RECTANGULAR, position, emission, 0.0, float(np.float32(np.sqrt((c * c).sum()))), u, v)

    @staticmethod
    def distant(position, emission):
        """LightType::Distant (globals.rs:72): light from the direction normalize(position); area 0 = no MIS (tracer.rs:158)."""
        return AnalyticalLight(_abi.RPT_LIGHT_DISTANT, position, emission, 0.0, 0.0)


class Material:
    """A material PATCH: only the fields passed are written over Material::new()
    (material.rs:82-114), the way analytical.rs:56-58 assigns individual fields."""

    _FIELDS = {
        "rgb": _abi.RPT_MAT_RGB, "emission": _abi.RPT_MAT_EMISSION, "anisotropic": _abi.RPT_MAT_ANISOTROPIC,
        "metallic": _abi.RPT_MAT_METALLIC, "roughness": _abi.RPT_MAT_ROUGHNESS, "subsurface": _abi.RPT_MAT_SUBSURFACE,
        "specular_tint": _abi.RPT_MAT_SPECULAR_TINT, "sheen": _abi.RPT_MAT_SHEEN, "sheen_tint": _abi.RPT_MAT_SHEEN_TINT,
        "clearcoat": _abi.RPT_MAT_CLEARCOAT, "clearcoat_gloss": _abi.RPT_MAT_CLEARCOAT_GLOSS,
        "spec_trans": _abi.RPT_MAT_SPEC_TRANS, "ior": _abi.RPT_MAT_IOR,
    }

    MEDIUM_TYPES = {"none": _abi.RPT_MEDIUM_NONE, "absorb": _abi.RPT_MEDIUM_ABSORB, "scatter": _abi.RPT_MEDIUM_SCATTER,
                    "emissive": _abi.RPT_MEDIUM_EMISSIVE}

    def __init__(self, checker_dir=None, medium=None, **fields):
        for k in fields:
            if k not in self._FIELDS:
                raise TypeError("unknown material field %r" % k)
        self.fields = fields
        self.checker_dir = checker_dir       # (scale, offset, colour_a, colour_b) or None
        # Material.medium (material.rs:16-21): dict(type="absorb"|"scatter"|"emissive"|"none", density, color, anisotropy);
        # read only by scenes with `media = True` (project-defined, include/rpt.h)
        self.medium = medium

    def to_c(self):
        m = _abi.rpt_material()
        for k, v in self.fields.items():
            m.mask |= self._FIELDS[k]
            if k in ("rgb", "emission"):
                setattr(m, k, _abi.F3(*v))
            else:
                setattr(m, k, float(v))
        if self.checker_dir is not None:
            m.proc_kind = _abi.RPT_PROC_CHECKER_DIR
            m.proc_params = _abi.F4(*self.checker_dir)
        if self.medium is not None:
            m.mask |= _abi.RPT_MAT_MEDIUM
            t = self.medium.get("type", "none")
            m.medium_type = self.MEDIUM_TYPES[t] if isinstance(t, str) else int(t)
            m.medium_density = float(self.medium.get("density", 0.0))
            m.medium_color = _abi.F3(*self.medium.get("color", (0.0, 0.0, 0.0)))
            m.medium_anisotropy = float(self.medium.get("anisotropy", 0.0))
        return m


class Scene:
    """Data-driven counterpart of trait Scene (scene.rs:5-90): instead of callbacks the
    scene describes itself (spheres, planes, lights, material patches, camera)."""

    def __init__(self):
        self.camera = Pinhole()
        self.spheres = []        # (center, radius, material_index)
        self.planes = []         # (normal, point, min_denom, material_index[, max_t])
        self.lights = []         # AnalyticalLight
        self.materials = []      # Material
        self.background = dict(kind=_abi.RPT_BG_CONSTANT, colour_a=(0.0, 0.0, 0.0), colour_b=(0.0, 0.0, 0.0), gamma=2.2, scale=1.0)
        self.eps = 0.005         # tracer.rs:16
        self.max_depth = 4       # scene.rs:28-30
        self.any_hit_uses_max_dist = False
        self.sample_all_light_types = False   # rectangular / distant lights do something (project-defined; off = the reference)
        self.media = False       # Material.medium is read: participating media (project-defined, include/rpt.h; off = the reference)
        self.sdf = None          # dict(prims=[(kind, center, (p0, p1))], material, smooth_k, max_steps, hit_eps, max_t, normal_eps)
        self._keep = None

    def recursion_depth(self):
        return self.max_depth

    def number_of_lights(self):
        return len(self.lights)

    def light_at(self, index):
        return self.lights[index]

    def describe(self):
        """-> rpt_scene_desc (keeps the backing arrays alive on self)."""
        d = _abi.rpt_scene_desc()
        d.abi_version = _abi.RPT_ABI_VERSION
        d.flags = (_abi.RPT_SCENE_ANYHIT_USES_MAX_DIST if self.any_hit_uses_max_dist else 0) | \
                  (_abi.RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES if self.sample_all_light_types else 0) | \
                  (_abi.RPT_SCENE_MEDIA if self.media else 0)
        d.camera.origin = _abi.F3(*self.camera.origin)
        d.camera.center = _abi.F3(*self.camera.center)
        d.camera.fov_deg = self.camera.fov
        bg = self.background
        d.background.kind = bg["kind"]
        d.background.colour_a = _abi.F3(*bg["colour_a"])
        d.background.colour_b = _abi.F3(*bg["colour_b"])
        d.background.gamma = bg["gamma"]
        d.background.scale = bg["scale"]
        d.eps = self.eps
        d.max_depth = self.max_depth
        sph = (_abi.rpt_sphere * max(1, len(self.spheres)))()
        for i, (c, r, m) in enumerate(self.spheres):
            sph[i].center = _abi.F3(*c); sph[i].radius = r; sph[i].material = m
        pl = (_abi.rpt_plane * max(1, len(self.planes)))()
        for i, pln in enumerate(self.planes):
            n, p, md, m = pln[:4]
            pl[i].normal = _abi.F3(*n); pl[i].point = _abi.F3(*p); pl[i].min_denom = md; pl[i].material = m
            pl[i].max_t = pln[4] if len(pln) > 4 else 0.0
        li = (_abi.rpt_light * max(1, len(self.lights)))()
        for i, L in enumerate(self.lights):
            li[i].type = L.light_type; li[i].position = _abi.F3(*L.position); li[i].emission = _abi.F3(*L.emission)
            li[i].radius = L.radius; li[i].area = L.area; li[i].u = _abi.F3(*L.u); li[i].v = _abi.F3(*L.v)
        ma = (_abi.rpt_material * max(1, len(self.materials)))()
        for i, M in enumerate(self.materials):
            ma[i] = M.to_c()
        d.n_spheres = len(self.spheres); d.spheres = C.cast(sph, C.POINTER(_abi.rpt_sphere))
        d.n_planes = len(self.planes); d.planes = C.cast(pl, C.POINTER(_abi.rpt_plane))
        d.n_lights = len(self.lights); d.lights = C.cast(li, C.POINTER(_abi.rpt_light))
        d.n_materials = len(self.materials); d.materials = C.cast(ma, C.POINTER(_abi.rpt_material))
        sd = None
        if self.sdf:
            sd = (_abi.rpt_sdf_prim * len(self.sdf["prims"]))()
            for i, (kind, c, prm) in enumerate(self.sdf["prims"]):
                sd[i].kind = kind; sd[i].center = _abi.F3(*c); sd[i].params = (C.c_float * 2)(*prm)
            d.sdf.n_prims = len(self.sdf["prims"]); d.sdf.prims = C.cast(sd, C.POINTER(_abi.rpt_sdf_prim))
            d.sdf.material = self.sdf["material"]; d.sdf.smooth_k = self.sdf.get("smooth_k", 0.5)
            d.sdf.max_steps = self.sdf.get("max_steps", 128); d.sdf.hit_eps = self.sdf.get("hit_eps", 1e-3)
            d.sdf.max_t = self.sdf.get("max_t", 100.0); d.sdf.normal_eps = self.sdf.get("normal_eps", 1e-3)
        self._keep = (sph, pl, li, ma, sd)
        return d


class AnalyticalScene(Scene):
    """renderer/src/analytical.rs:4-205 as data."""

    def __init__(self):
        super().__init__()
        em = 3.0
        self.lights = [AnalyticalLight.spherical((3.0, 2.0, 2.0), 1.0, (em, em, em))]            # analytical.rs:15-16
        self.materials = [
            Material(rgb=(1.0, 1.0, 1.0), roughness=0.05, metallic=1.0),                           # analytical.rs:56-58
            Material(rgb=(1.0, 0.186, 0.0), clearcoat=1.0, clearcoat_gloss=1.0, roughness=0.1),    # analytical.rs:82-85
            Material(roughness=1.0, checker_dir=(0.5, 100.0, 0.25, 0.1)),                          # analytical.rs:107-116
        ]
        self.spheres = [((-1.1, 0.0, 0.0), 1.0, 0), ((1.1, 0.0, 0.0), 1.0, 1)]                     # analytical.rs:41,70
        self.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 2)]                             # analytical.rs:194-198
        self.background = dict(kind=_abi.RPT_BG_GRADIENT_Y, colour_a=(1.0, 1.0, 1.0), colour_b=(0.5, 0.7, 1.0), gamma=2.2, scale=0.5)


class ColorBuffer:
    """buffer.rs:6-32: host RGBA f32 buffer, row 0 = top, plus the frame counter."""

    def __init__(self, width, height):
        self.width = int(width)
        self.height = int(height)
        self.pixels = np.zeros(self.width * self.height * 4, dtype=np.float32)
        self.frames = 0

    def at(self, x, y):                                            # buffer.rs:29-32
        i = y * self.width * 4 + x * 4
        return [float(v) for v in self.pixels[i:i + 4]]

    def image(self):
        return self.pixels.reshape(self.height, self.width, 4)

    def convert_to_u8(self, frame=None):                           # buffer.rs:55-64, on the device
        import torch
        dev = torch.from_numpy(self.pixels).to("cuda")
        out = _convert_to_u8_tensor(dev, self.width, self.height)
        res = out.cpu().numpy().reshape(-1)
        if frame is not None:
            frame[:] = res
        return res

    def to_u8_vec(self):                                           # buffer.rs:37-52 (same arithmetic)
        return self.convert_to_u8()

    def denoise(self, iterations=3, edge_k=2.0, device=0):
        """A denoised copy (project-defined, include/rpt.h "denoiser"; the reference lists one as a Todo): a new ColorBuffer."""
        out = ColorBuffer(self.width, self.height)
        ctx = _ctx_for(device)
        check(lib().rpt_denoise(ctx, self.pixels.ctypes.data, out.pixels.ctypes.data, self.width, self.height, iterations, edge_k), ctx)
        out.frames = self.frames
        return out


class DeviceColorBuffer:
    """ColorBuffer whose pixels live in HBM (a torch CUDA tensor): what the render loop
    uses when frames are accumulated on the device and fetched once at the end."""

    def __init__(self, width, height, device="cuda:0"):
        import torch
        self.width = int(width)
        self.height = int(height)
        self.pixels = torch.zeros(self.height, self.width, 4, dtype=torch.float32, device=device)
        self.frames = 0

    def to_host(self):
        b = ColorBuffer(self.width, self.height)
        b.pixels = self.pixels.detach().cpu().numpy().reshape(-1).copy()
        b.frames = self.frames
        return b

    def convert_to_u8(self):
        return _convert_to_u8_tensor(self.pixels, self.width, self.height)

    def denoise(self, iterations=3, edge_k=2.0, out=None):
        """A denoised copy on the device (include/rpt.h "denoiser"): a new DeviceColorBuffer, or `out` (same size)."""
        import torch
        if out is None:
            out = DeviceColorBuffer(self.width, self.height, device=self.pixels.device)
        assert out.width == self.width and out.height == self.height
        ctx = _ctx_for(self.pixels.device.index or 0)
        stream = torch.cuda.current_stream(self.pixels.device).cuda_stream
        check(lib().rpt_denoise_device(ctx, self.pixels.data_ptr(), out.pixels.data_ptr(), self.width, self.height, iterations, edge_k,
                                       C.c_void_p(stream)), ctx)
        out.frames = self.frames
        return out

    def convert_to_u8_at(self, frame, at):
        """buffer.rs:67-89: blit into `frame` (a CUDA uint8 tensor [at[3], at[2], 4]) at offset (at[0], at[1])."""
        import torch
        assert frame.is_cuda and frame.dtype == torch.uint8 and frame.is_contiguous() and frame.numel() == at[2] * at[3] * 4
        ctx = _ctx_for(self.pixels.device.index or 0)
        stream = torch.cuda.current_stream(self.pixels.device).cuda_stream
        check(lib().rpt_convert_to_u8_at_device(ctx, self.pixels.data_ptr(), self.width, self.height, frame.data_ptr(),
                                                at[0], at[1], at[2], at[3], C.c_void_p(stream)), ctx)
        return frame


_default_ctx = {}


def _ctx_for(device_index):
    """A bare context (no scene) for stateless helpers such as convert_to_u8."""
    if device_index not in _default_ctx:
        h = C.c_void_p()
        check(lib().rpt_create(C.byref(h), device_index))
        _default_ctx[device_index] = h
    return _default_ctx[device_index]


def _convert_to_u8_tensor(pixels, width, height):
    import torch
    assert pixels.is_cuda and pixels.dtype == torch.float32 and pixels.is_contiguous()
    idx = pixels.device.index or 0
    out = torch.empty(height, width, 4, dtype=torch.uint8, device=pixels.device)
    ctx = _ctx_for(idx)
    stream = torch.cuda.current_stream(pixels.device).cuda_stream
    check(lib().rpt_convert_to_u8_device(ctx, pixels.data_ptr(), out.data_ptr(), width, height, C.c_void_p(stream)), ctx)
    return out


def comm_unique_id():
    """128 opaque bytes from rank 0 that every rank passes to Tracer(..., rank=, world=, unique_id=) (ncclGetUniqueId)."""
    uid = _abi.rpt_unique_id()
    check(lib().rpt_comm_unique_id(C.byref(uid)))
    return C.string_at(C.byref(uid), _abi.RPT_UNIQUE_ID_BYTES)


class Tracer:
    """tracer.rs:5-19.  Owns the scene; render() is the boundary into the HIP library.

    Tracer(scene, device=0)                       one GPU
    Tracer(scene, devices=[0, 1, ...])            the GPUs of a node driven by this process: render() fans out over them
                                                  inside the call, where the reference fans out over rayon threads
    Tracer(scene, device=d, rank=r, world=n,      one process per GPU (torch.distributed.run): collective construction;
           unique_id=comm_unique_id() of rank 0)  the resident_* calls are then collective too"""

    def __init__(self, scene, device=0, seed=1, devices=None, rank=None, world=None, unique_id=None):
        self._scene = scene
        self.seed = int(seed)
        self.flags = 0                 # RPT_RENDER_* bits (0 = the strict, bit-exact regenerating kernel)
        self._frame = None             # resident_to_u8's page-locked frame
        self._h = C.c_void_p()
        if devices is not None:
            ids = (C.c_int * len(devices))(*devices)
            self.device = int(devices[0])
            check(lib().rpt_create_multi(C.byref(self._h), ids, len(devices)))
        elif world is not None:
            uid = _abi.rpt_unique_id()
            C.memmove(C.byref(uid), unique_id, _abi.RPT_UNIQUE_ID_BYTES)
            self.device = int(device)
            check(lib().rpt_create_rank(C.byref(self._h), self.device, int(rank), int(world), C.byref(uid)))
        else:
            self.device = int(device)
            check(lib().rpt_create(C.byref(self._h), self.device))
        self.upload_scene()

    def world(self):
        """(rank of this context's first device, ranks in all, devices this context drives)"""
        r, w, n = C.c_int(), C.c_int(), C.c_int()
        check(lib().rpt_world(self._h, C.byref(r), C.byref(w), C.byref(n)), self._h)
        return r.value, w.value, n.value

    def set_tile_rows(self, tile_rows):
        check(lib().rpt_set_tile_rows(self._h, int(tile_rows)), self._h)

    def set_dispatch(self, cost_order=1, unit_rounds=12, unit_min_spp=64, unit_slots=0):
        """Scheduling of the launches (include/rpt.h, rpt_set_dispatch): changes when a sample is computed, never its value."""
        check(lib().rpt_set_dispatch(self._h, int(cost_order), int(unit_rounds), int(unit_min_spp), int(unit_slots)), self._h)

    def upload_scene(self):
        """Call after mutating the scene returned by scene()."""
        desc = self._scene.describe()
        check(lib().rpt_upload_scene(self._h, C.byref(desc)), self._h)

    def scene(self):                                               # tracer.rs:629-631
        return self._scene

    def render(self, buffer):
        """Render one frame and accumulate into the pixels buffer (tracer.rs:21-123)."""
        self.render_n(buffer, 1)

    def render_n(self, buffer, spp):
        """`spp` consecutive render() calls folded into one launch; bit-identical to them."""
        if isinstance(buffer, ColorBuffer):
            assert buffer.pixels.dtype == np.float32 and buffer.pixels.size == buffer.width * buffer.height * 4
            check(lib().rpt_render(self._h, buffer.pixels.ctypes.data, buffer.width, buffer.height, buffer.frames,
                                   spp, self.seed, self.flags), self._h)
        else:
            import torch
            px = buffer.pixels
            assert px.is_cuda and px.is_contiguous() and px.dtype == torch.float32
            assert (px.device.index or 0) == self.device
            stream = torch.cuda.current_stream(px.device).cuda_stream
            check(lib().rpt_render_device(self._h, px.data_ptr(), buffer.width, buffer.height, buffer.frames, spp,
                                          self.seed, self.flags, buffer.height, 0, 1, C.c_void_p(stream)), self._h)
        buffer.frames += spp                                       # tracer.rs:121

    # --- resident ColorBuffer: the interactive loop of renderer/src/main.rs:113-124 without the PCIe round trip
    def render_resident(self, width, height, spp=1):
        check(lib().rpt_resident_render(self._h, width, height, spp, self.seed, self.flags), self._h)

    def resident_frames(self):
        f = C.c_uint64(0)
        check(lib().rpt_resident_frames(self._h, C.byref(f)), self._h)
        return f.value

    def resident_to_host(self, width, height):
        """Collective on a multi-process tracer; every rank gets a ColorBuffer, only rank 0's is filled."""
        b = ColorBuffer(width, height)
        check(lib().rpt_resident_download(self._h, b.pixels.ctypes.data), self._h)
        b.frames = self.resident_frames()
        return b

    def resident_to_u8(self, width, height, frame=None):
        """buffer.convert_to_u8(frame) on the device + the 4 B per pixel over PCIe.  `frame`: the caller's uint8 array
        (renderer/src/main.rs:122 passes the window's); None: the tracer's own page-locked frame, REUSED by the next call."""
        if frame is None:
            n = width * height * 4
            if self._frame is None or self._frame.size != n:
                self._drop_frame()
                self._frame = np.zeros(n, dtype=np.uint8)
                check(lib().rpt_host_pin(self._frame.ctypes.data, n))
            frame = self._frame
        assert frame.dtype == np.uint8 and frame.size == width * height * 4
        check(lib().rpt_resident_download_u8(self._h, frame.ctypes.data), self._h)
        return frame

    def _drop_frame(self):
        if getattr(self, "_frame", None) is not None:
            lib().rpt_host_unpin(self._frame.ctypes.data)
            self._frame = None

    def resident_reset(self):
        check(lib().rpt_resident_reset(self._h), self._h)

    def resident_upload(self, buffer):
        """Start the resident buffer from a host ColorBuffer (resume; each rank takes the rows it owns)."""
        assert buffer.pixels.dtype == np.float32 and buffer.pixels.size == buffer.width * buffer.height * 4
        check(lib().rpt_resident_upload(self._h, buffer.pixels.ctypes.data, buffer.width, buffer.height, buffer.frames), self._h)

    def resident_gather(self, image=None):
        """Enqueue the assembly of the resident image on the root device (RCCL gather over xGMI + scatter kernel).
        image: a [h, w, 4] f32 CUDA tensor on rank 0's device, or None (the library's own staging image)."""
        check(lib().rpt_resident_gather_device(self._h, image.data_ptr() if image is not None else None), self._h)

    def resident_sync(self):
        check(lib().rpt_resident_sync(self._h), self._h)

    def resident_kernel_ms(self):
        ms = C.c_float(0.0)
        check(lib().rpt_resident_kernel_ms(self._h, C.byref(ms)), self._h)
        return ms.value

    def render_tile(self, tile_pixels, width, height, frames_done, spp, tile_rows, rank, world):
        """Render this rank's rows of a row-tiled image into its compact tile tensor."""
        import torch
        assert tile_pixels.is_cuda and tile_pixels.is_contiguous() and tile_pixels.dtype == torch.float32
        stream = torch.cuda.current_stream(tile_pixels.device).cuda_stream
        check(lib().rpt_render_device(self._h, tile_pixels.data_ptr(), width, height, frames_done, spp, self.seed, self.flags,
                                      tile_rows, rank, world, C.c_void_p(stream)), self._h)

    def close(self):
        self._drop_frame()
        if self._h:
            lib().rpt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
