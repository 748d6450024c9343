"""Dispatch of the render launches (include/rpt.h, rpt_set_dispatch; kernels.hip, "Dispatch: units, their order, their hand-off"):
the order of the tiles and the cutting of a launch's samples into chunks that are handed from workgroup to workgroup decide when
and where a sample is computed — never its value.  Every case is compared bit for bit with the oracle or with the undivided
launch.  Needs an MI355X."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import conftest
from test_gpu_parity import assert_bit_identical

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


# (cost_order, unit_rounds, unit_min_spp, unit_slots): pretending the device holds 8 workgroups makes small frames many rounds
PLAIN = (0, 0, 64, 0)                 # bottom rows first, one unit per tile
MANY_CHUNKS = (1, 100000, 1, 8)       # cost order, every sample its own chunk
SOME_CHUNKS = (1, 40, 3, 8)


def _render(rpt, torch, scene, w, h, spp, dispatch, steps=1, seed=1, flags=0):
    t = rpt.Tracer(scene, device=0, seed=seed)
    t.flags = flags
    t.set_dispatch(*dispatch)
    buf = rpt.DeviceColorBuffer(w, h)
    for _ in range(steps):
        t.render_n(buf, spp)
    torch.cuda.synchronize()
    img = buf.pixels.cpu().numpy()
    t.close()
    return img


@pytest.mark.parametrize("dispatch", [PLAIN, MANY_CHUNKS, SOME_CHUNKS], ids=["plain", "one-sample-chunks", "chunks-of-3"])
@pytest.mark.parametrize("size", [(160, 96), (333, 77), (64, 200)], ids=["160x96", "333x77-ragged", "64x200"])
def test_chunked_launches_match_the_oracle(rpt, oracle, torch_cuda, dispatch, size):
    """Ragged frames have tiles with waves that own no pixel at all: they must still count for the hand-off."""
    w, h = size
    spp = 10
    s = rpt.AnalyticalScene()
    want = oracle.render(s.describe(), w, h, spp, seed=1)
    got = _render(rpt, torch_cuda, s, w, h, spp, dispatch)
    assert_bit_identical(got, want, "%dx%d, dispatch %r" % (w, h, dispatch))


def test_progressive_steps_learn_an_order_and_stay_exact(rpt, oracle, torch_cuda):
    """Step k is dispatched in the order of step k - 1's costs; three steps of 7 samples are 21 render() calls."""
    w, h = 208, 144
    s = rpt.AnalyticalScene()
    want = oracle.render(s.describe(), w, h, 21, seed=1)
    for dispatch in (PLAIN, MANY_CHUNKS, SOME_CHUNKS):
        got = _render(rpt, torch_cuda, s, w, h, 7, dispatch, steps=3)
        assert_bit_identical(got, want, "3 x 7 spp, dispatch %r" % (dispatch,))


def test_more_samples_than_the_kernel_tables_hold_is_one_launch(rpt, oracle, torch_cuda):
    """The state-machine kernels keep a chunk's samples in LDS tables of 512 entries; 1 100 samples are chunks of one launch."""
    w, h, spp = 48, 32, 1100
    s = rpt.AnalyticalScene()
    want = oracle.render(s.describe(), w, h, spp, seed=1)
    for dispatch in (PLAIN, (1, 12, 64, 0), (1, 100, 100, 2)):
        got = _render(rpt, torch_cuda, s, w, h, spp, dispatch)
        assert_bit_identical(got, want, "1 100 spp, dispatch %r" % (dispatch,))


def test_every_kernel_family_takes_chunks(rpt, oracle, torch_cuda):
    from rust_pathtracer_amd import scenes
    A = rpt._abi
    cases = [("sdf two rooms", scenes.sdf_scene(), 0), ("300 spheres", scenes.random_spheres_scene(300, 5), 0),
             ("media", scenes.media_scene(), 0), ("roulette", rpt.AnalyticalScene(), A.RPT_RENDER_RUSSIAN_ROULETTE)]
    w, h, spp = 112, 80, 6
    for name, scene, flags in cases:
        oflags = flags & A.RPT_RENDER_RUSSIAN_ROULETTE
        want = oracle.render(scene.describe(), w, h, spp, seed=1, render_flags=oflags)
        for dispatch in (PLAIN, MANY_CHUNKS):
            got = _render(rpt, torch_cuda, scene, w, h, spp, dispatch, flags=flags)
            assert_bit_identical(got, want, "%s, dispatch %r" % (name, dispatch))


def test_a_scene_without_bounces_with_chunks(rpt, oracle, torch_cuda):
    s = rpt.AnalyticalScene()
    s.max_depth = 0
    w, h, spp = 100, 52, 9
    want = oracle.render(s.describe(), w, h, spp, seed=1)
    got = _render(rpt, torch_cuda, s, w, h, spp, MANY_CHUNKS)
    assert_bit_identical(got, want, "max_depth 0, chunked")


def test_rank_tiles_with_chunks(rpt, oracle, torch_cuda):
    """One rank's rows of a row-tiled image (cyclic 2-row blocks), chunked: the rows of the full oracle frame it owns."""
    from rust_pathtracer_amd import tiling
    torch = torch_cuda
    w, h, spp, world, tile_rows = 200, 120, 8, 3, 2
    s = rpt.AnalyticalScene()
    want = oracle.render(s.describe(), w, h, spp, seed=1)
    t = rpt.Tracer(s, device=0, seed=1)
    t.set_dispatch(*MANY_CHUNKS)
    for rank in range(world):
        rows = tiling.tile_global_rows(h, tile_rows, rank, world)
        tile = torch.zeros(len(rows), w, 4, dtype=torch.float32, device="cuda")
        t.render_tile(tile, w, h, 0, spp, tile_rows, rank, world)
        torch.cuda.synchronize()
        assert_bit_identical(tile.cpu().numpy(), want[rows], "rank %d of %d" % (rank, world))
    t.close()


def test_full_size_frame_in_32_chunks_equals_the_undivided_launch(rpt, torch_cuda):
    """The hand-off under load: 8 160 tiles x 32 chunks of one sample on the real device geometry (every chunk's workgroups
    wait for, acquire and re-read what another compute unit — usually another XCD — has just written), three steps so that the
    order is the learned one; the whole frame must hash like the plain launches'."""
    w, h, spp = 1920, 1080, 32
    s = rpt.AnalyticalScene()
    ref = _render(rpt, torch_cuda, s, w, h, spp, PLAIN, steps=3)
    got = _render(rpt, torch_cuda, s, w, h, spp, (1, 1000, 1, 0), steps=3)
    assert hashlib.sha1(got.tobytes()).hexdigest() == hashlib.sha1(ref.tobytes()).hexdigest()
    got = _render(rpt, torch_cuda, s, w, h, spp, (1, 12, 8, 0), steps=3)
    assert hashlib.sha1(got.tobytes()).hexdigest() == hashlib.sha1(ref.tobytes()).hexdigest()


def test_the_dispatch_order_is_a_permutation_that_puts_expensive_tiles_first(rpt, torch_cuda):
    torch = torch_cuda
    w, h, spp = 640, 368, 32
    t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
    buf = rpt.DeviceColorBuffer(w, h)
    n = (w // 16) * (h // 16)
    raw = np.zeros(10 * n, dtype=np.uint32)
    nt = C.c_uint32(0)
    for step in range(3):
        t.render_n(buf, spp)
        rpt._lib.check(rpt.lib().rpt_debug_sched_read(t._h, raw.ctypes.data_as(C.POINTER(C.c_uint32)), n, C.byref(nt)), t._h)
        assert nt.value == n
        order = raw[4 * n:5 * n]
        assert sorted(order.tolist()) == list(range(n)), "step %d: the dispatch order is not a permutation" % step
        cost = raw[:4 * n].reshape(n, 4).max(axis=1).astype(np.float64)
        assert (cost > 0).all(), "every tile of this frame has pixels in every wave"
        along = cost[order]                                           # costs in dispatch order: non-increasing up to the bucket width
        assert along[: n // 8].mean() > 3.0 * along[-n // 8:].mean(), "floor and spheres before sky"
        assert (np.diff(along) <= cost.max() / 1024.0 * 2.0 + 1.0).all(), "step %d: not sorted by cost" % step
    t.close()


def test_set_dispatch_validates(rpt, torch_cuda):
    A = rpt._abi
    t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
    assert rpt.lib().rpt_set_dispatch(t._h, 7, 0, 0, 0) == A.RPT_ERR_INVALID_ARG
    assert rpt.lib().rpt_set_dispatch(None, 1, 12, 64, 0) == A.RPT_ERR_INVALID_ARG
    t.close()


def test_kernels_that_know_the_table_sizes_equal_the_general_ones(rpt, oracle, torch_cuda):
    """Scenes with the reference scene's table sizes (2 spheres, 1 plane, 1 light) take instantiations of the megakernel and of the
    compacting kernel that know those sizes, SDF objects of 1-4 primitives over one plane under one light instantiations of the march
    kernel (kernels.hip, sized_scene); RPT_NO_SIZED_KERNELS=1 takes the general kernels.  Same frames — for the reference's scene, for
    one that shares nothing with it but the sizes, for SDF objects of 1 ... 5 primitives (5: general either way) — and the oracle's."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = r"""
import hashlib, sys
sys.path.insert(0, %r)
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
def other():
    s = rpt.AnalyticalScene()
    s.materials = [scenes.full_material(rgb=(0.9, 0.9, 1.0), roughness=0.03, spec_trans=1.0, ior=1.45),
                   scenes.full_material(rgb=(0.2, 0.7, 0.3), roughness=0.6, sheen=0.8, subsurface=0.4, emission=(0.3, 0.1, 0.0)),
                   rpt.Material(rgb=(0.7, 0.7, 0.7), roughness=0.4, metallic=1.0)]
    s.spheres = [((-0.7, 0.2, 0.4), 0.8, 1), ((0.9, -0.3, -0.2), 0.7, 0)]
    s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 2, 30.0)]
    s.lights = [rpt.AnalyticalLight.spherical((-2.0, 3.0, 1.0), 0.6, (6.0, 5.0, 4.0))]
    s.max_depth = 6
    s.any_hit_uses_max_dist = True
    return s
def blob(n_prims):
    s = scenes.sdf_scene()
    prims = list(s.sdf["prims"])
    A = rpt._abi
    more = [(A.RPT_SDF_SPHERE, (0.6, 0.5, -0.4), (0.4, 0.0)), (A.RPT_SDF_TORUS_Y, (0.2, 0.3, 0.0), (0.7, 0.12)), (A.RPT_SDF_SPHERE, (-0.2, 0.8, 0.3), (0.3, 0.0))]
    s.sdf["prims"] = (prims + more)[:n_prims]
    return s
cases = [("reference", rpt.AnalyticalScene()), ("other", other())] + [("sdf " + str(n), blob(n)) for n in (1, 2, 3, 4, 5)]
for name, scene in cases:
    t = rpt.Tracer(scene, device=0, seed=3)
    buf = rpt.DeviceColorBuffer(208, 112)
    for n in (1, 1, 6, 9):
        t.render_n(buf, n)
    torch.cuda.synchronize()
    print("HASH", name, hashlib.sha1(buf.pixels.cpu().numpy().tobytes()).hexdigest())
    t.close()
""" % here
    out = {}
    for no_sized in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, RPT_NO_SIZED_KERNELS=no_sized), timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        out[no_sized] = [l for l in r.stdout.splitlines() if l.startswith("HASH")]
        assert len(out[no_sized]) == 7
    assert out["0"] == out["1"], "kernels with and without the table sizes differ: %r vs %r" % (out["0"], out["1"])
    # and against the oracle, in this process (the sized kernels: the default)
    from rust_pathtracer_amd import scenes
    s = rpt.AnalyticalScene()
    s.materials = [scenes.full_material(rgb=(0.9, 0.9, 1.0), roughness=0.03, spec_trans=1.0, ior=1.45),
                   scenes.full_material(rgb=(0.2, 0.7, 0.3), roughness=0.6, sheen=0.8, subsurface=0.4, emission=(0.3, 0.1, 0.0)),
                   rpt.Material(rgb=(0.7, 0.7, 0.7), roughness=0.4, metallic=1.0)]
    s.spheres = [((-0.7, 0.2, 0.4), 0.8, 1), ((0.9, -0.3, -0.2), 0.7, 0)]
    s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 2, 30.0)]
    s.lights = [rpt.AnalyticalLight.spherical((-2.0, 3.0, 1.0), 0.6, (6.0, 5.0, 4.0))]
    s.max_depth = 6
    s.any_hit_uses_max_dist = True
    want = oracle.render(s.describe(), 120, 72, 7, seed=1)
    t = rpt.Tracer(s, device=0, seed=1)
    buf = rpt.DeviceColorBuffer(120, 72)
    for n in (1, 6):
        t.render_n(buf, n)
    torch_cuda.cuda.synchronize()
    assert_bit_identical(buf.pixels.cpu().numpy(), want, "a scene of the reference's table sizes and nothing else in common")
    t.close()
    for n_prims in (1, 2, 4):                                         # (3: scenes.sdf_scene itself, in test_gpu_parity)
        s = scenes.sdf_scene()
        A = rpt._abi
        s.sdf["prims"] = (list(s.sdf["prims"]) + [(A.RPT_SDF_SPHERE, (0.6, 0.5, -0.4), (0.4, 0.0))])[:n_prims]
        want = oracle.render(s.describe(), 96, 56, 4, seed=1)
        t = rpt.Tracer(s, device=0, seed=1)
        buf = rpt.DeviceColorBuffer(96, 56)
        t.render_n(buf, 4)
        torch_cuda.cuda.synchronize()
        assert_bit_identical(buf.pixels.cpu().numpy(), want, "SDF object of %d primitives" % n_prims)
        t.close()


def _table_scene(rpt, which):
    """Scenes for the material table (dev_integrator.h, MaterialTable): what a row depends on, and what rules a table out."""
    from rust_pathtracer_amd import scenes
    from scene_fuzz import random_small_scene
    if isinstance(which, int):                                        # random materials, random scale (2^-33 ... 2^33), roulette or not
        s, _, flags, _ = random_small_scene(rpt, which, n_spheres=2, n_lights=1)
        return s, flags
    if isinstance(which, str) and which.startswith("sdf"):
        # the SDF march kernel's table: the plane, the object and at most one analytical sphere (3 bits + colour + side = 32 rows)
        s = scenes.sdf_scene()
        if which == "sdf no sphere":
            s.spheres = []
        elif which == "sdf two spheres":                              # 4 primitives: no table
            s.spheres = list(s.spheres) + [((-1.8, -0.5, 0.9), 0.5, 0)]
        elif which == "sdf two lights":                               # not the sizes the march kernel is instantiated for: the table's shape as data
            s.lights = list(s.lights) + [rpt.AnalyticalLight.spherical((2.0, 2.5, -1.0), 0.4, (5.0, 6.0, 7.0))]
        elif which == "sdf two planes":                               # the object and two planes, no sphere: three primitives
            s.spheres = []
            s.planes = list(s.planes) + [((0.0, 0.0, 1.0), (0.0, 0.0, -3.0), 0.0001, 0)]
        elif which == "sdf checker object":                           # the one procedural material on the object, glass beside it
            s.materials = [rpt.Material(roughness=0.4, checker_dir=(3.0, 11.0, 0.8, 0.1)),
                           scenes.full_material(rgb=(0.95, 0.95, 1.0), roughness=0.05, spec_trans=1.0, ior=1.5),
                           rpt.Material(rgb=(0.5, 0.5, 0.5), roughness=0.7, metallic=1.0)]
        else:
            assert which == "sdf"
        return s, 0
    s = rpt.AnalyticalScene()
    if which == "overlapping patches":
        # two spheres through each other whose patches write DIFFERENT fields: a hit on the second after the first was accepted
        # keeps the first one's fields (analytical.rs:56-58 writes field by field), i.e. rows with both sphere bits set
        s.materials = [rpt.Material(rgb=(0.9, 0.3, 0.2), clearcoat=1.0, clearcoat_gloss=0.7, emission=(0.05, 0.0, 0.1)),
                       rpt.Material(roughness=0.15, metallic=1.0, anisotropic=0.6, sheen=0.5, sheen_tint=0.4, specular_tint=0.7),
                       rpt.Material(roughness=0.8, checker_dir=(0.5, 100.0, 0.25, 0.1))]
        s.spheres = [((0.2, 0.0, -0.6), 1.0, 0), ((-0.3, 0.1, 0.2), 0.9, 1)]
        s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 2)]
    elif which == "camera inside glass":
        # the first hit comes from inside (normal . ray > 0: eta = ior, the rows of the other side)
        s.materials = [scenes.full_material(rgb=(0.95, 0.95, 1.0), roughness=0.02, spec_trans=1.0, ior=1.5),
                       rpt.Material(rgb=(0.8, 0.6, 0.1), roughness=0.3, subsurface=0.5),
                       rpt.Material(roughness=1.0, checker_dir=(0.5, 100.0, 0.25, 0.1))]
        o = s.camera.origin
        s.spheres = [((o[0], o[1], o[2] - 0.5), 1.2, 0), ((0.0, 0.0, 0.0), 0.8, 1)]
        s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 2)]
        s.max_depth = 7
    elif which == "checker on a sphere":
        # the one procedural material on a sphere, a plain floor
        s.materials = [rpt.Material(roughness=0.5, checker_dir=(2.0, 7.0, 0.9, 0.05)), rpt.Material(rgb=(0.2, 0.5, 0.8), metallic=1.0, roughness=0.2),
                       rpt.Material(rgb=(0.6, 0.6, 0.6), roughness=0.9)]
        s.spheres = [((-0.8, 0.0, 0.0), 1.0, 0), ((0.9, 0.0, 0.3), 0.8, 1)]
        s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 2)]
    elif which == "extreme materials":
        # values that push State::finalize and get_spec_color out of the short divide's range (1 / ior = 1e20, aspect = sqrt(1e-7),
        # roughness 1e-30): a row is built with hipcc's own divide, a hit's material with the tracked one + the second computation
        s.materials = [scenes.full_material(rgb=(0.9, 0.9, 1.0), roughness=1e-30, spec_trans=1.0, ior=1e-20),
                       scenes.full_material(rgb=(1e-30, 0.7, 1e20), roughness=0.3, anisotropic=0.9999999 / 0.9, metallic=1.0, clearcoat=1.0, clearcoat_gloss=1.0),
                       rpt.Material(roughness=1.0, checker_dir=(0.5, 100.0, 0.25, 0.1))]
        s.spheres = [((-0.9, 0.0, 0.2), 0.9, 0), ((0.9, -0.1, -0.3), 0.8, 1)]
        s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 2)]
    elif which == "two checkers":
        # two procedural materials: the table has one bit for "the checker's second colour": the kernel without a table renders this one
        s.materials = [rpt.Material(roughness=0.5, checker_dir=(2.0, 7.0, 0.9, 0.05)), rpt.Material(rgb=(0.2, 0.5, 0.8), metallic=1.0, roughness=0.2),
                       rpt.Material(roughness=1.0, checker_dir=(0.5, 100.0, 0.25, 0.1))]
        s.spheres = [((-0.8, 0.0, 0.0), 1.0, 0), ((0.9, 0.0, 0.3), 0.8, 1)]
        s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 2)]
    elif which == "three spheres":
        # shapes other than the reference's 2 + 1 (round 5: the table's shape is data in render_*_table_kernel)
        s.materials = [rpt.Material(rgb=(0.9, 0.3, 0.2), clearcoat=1.0, clearcoat_gloss=0.7), rpt.Material(roughness=0.15, metallic=1.0, anisotropic=0.6),
                       scenes.full_material(rgb=(0.95, 0.95, 1.0), roughness=0.05, spec_trans=1.0, ior=1.5)]
        s.spheres = [((0.2, 0.0, -0.6), 1.0, 0), ((-0.9, 0.1, 0.4), 0.7, 1), ((1.1, -0.2, 0.5), 0.6, 2)]
        s.planes = []
    elif which == "one sphere two planes":
        s.materials = [rpt.Material(rgb=(0.2, 0.7, 0.3), roughness=0.3, sheen=0.6), rpt.Material(roughness=0.9, checker_dir=(0.5, 100.0, 0.25, 0.1)),
                       rpt.Material(rgb=(0.7, 0.7, 0.8), roughness=0.1, metallic=1.0)]
        s.spheres = [((0.0, 0.0, 0.0), 1.0, 0)]
        s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 1), ((0.0, 0.0, 1.0), (0.0, 0.0, -2.5), 0.0001, 2)]
    elif which == "three spheres on a floor":
        # four primitives: the megakernel's table has the rows for them (64), the compacting kernel's (one-sample launches) has not
        s.materials = [rpt.Material(rgb=(0.9, 0.3, 0.2), clearcoat=1.0, clearcoat_gloss=0.7), rpt.Material(roughness=0.15, metallic=1.0, anisotropic=0.6),
                       scenes.full_material(rgb=(0.95, 0.95, 1.0), roughness=0.05, spec_trans=1.0, ior=1.5), rpt.Material(roughness=1.0, checker_dir=(0.5, 100.0, 0.25, 0.1))]
        s.spheres = [((0.2, 0.0, -0.6), 1.0, 0), ((-1.3, -0.3, 0.4), 0.7, 1), ((1.4, -0.4, 0.5), 0.6, 2)]
        s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 3)]
    elif which == "two spheres two planes":
        s.materials = [rpt.Material(rgb=(0.2, 0.7, 0.3), roughness=0.3, sheen=0.6), rpt.Material(rgb=(0.8, 0.8, 0.2), roughness=0.05, metallic=1.0),
                       rpt.Material(roughness=0.9, checker_dir=(0.5, 100.0, 0.25, 0.1)), rpt.Material(rgb=(0.7, 0.7, 0.8), roughness=0.4)]
        s.spheres = [((-0.9, 0.0, 0.0), 1.0, 0), ((1.0, -0.2, 0.3), 0.8, 1)]
        s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 2), ((0.0, 0.0, 1.0), (0.0, 0.0, -2.5), 0.0001, 3)]
    elif which in ("five spheres on a floor", "six spheres two planes", "six primitives partial patches", "eight primitives many classes", "eight spheres four planes"):
        # five to eight primitives (round 6): the table by CLASS of accepted set (launch.h, MatClassMap) — whole materials: n + 1 classes;
        # patches that write different fields: more, up to 16; beyond that the material is built per hit
        full = scenes.full_material
        if which == "five spheres on a floor":
            s = scenes.six_primitive_scene()                          # (bench.py's `six_primitives` leg)
        elif which == "six spheres two planes":
            s.materials = [full(rgb=(0.9, 0.3, 0.2), roughness=0.4), full(rgb=(0.8, 0.8, 0.9), roughness=0.15, metallic=1.0),
                           full(rgb=(0.95, 0.95, 1.0), roughness=0.05, spec_trans=1.0, ior=1.5), full(rgb=(0.2, 0.7, 0.3), roughness=0.6, sheen=0.8),
                           full(rgb=(0.9, 0.8, 0.1), roughness=0.3, metallic=1.0), full(rgb=(0.3, 0.3, 0.9), roughness=0.2, clearcoat=1.0, clearcoat_gloss=1.0),
                           rpt.Material(roughness=1.0, checker_dir=(0.5, 100.0, 0.25, 0.1)), full(rgb=(0.7, 0.7, 0.8), roughness=0.5)]
            s.spheres = [((0.2, 0.0, -0.6), 1.0, 0), ((-1.3, -0.3, 0.4), 0.7, 1), ((1.4, -0.4, 0.5), 0.6, 2), ((-0.4, -0.6, 1.1), 0.4, 3), ((0.6, -0.65, 1.3), 0.35, 4),
                         ((0.1, 1.3, -0.2), 0.5, 5)]
            s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 6), ((0.0, 0.0, 1.0), (0.0, 0.0, -3.0), 0.0001, 7)]
        elif which == "six primitives partial patches":
            # spheres through each other whose patches write different fields (analytical.rs:56-58 writes field by field): sets with
            # the same last writers share a class, the others do not
            s.materials = [rpt.Material(rgb=(0.9, 0.3, 0.2), clearcoat=1.0, clearcoat_gloss=0.7), rpt.Material(roughness=0.15, metallic=1.0, anisotropic=0.6),
                           full(rgb=(0.2, 0.4, 0.9), roughness=0.5), full(rgb=(0.9, 0.9, 0.2), roughness=0.3, sheen=0.7),
                           full(rgb=(0.95, 0.95, 1.0), roughness=0.05, spec_trans=1.0, ior=1.5), rpt.Material(roughness=0.8, checker_dir=(0.5, 100.0, 0.25, 0.1))]      # 14 classes
            s.spheres = [((0.2, 0.0, -0.6), 1.0, 0), ((-0.3, 0.1, 0.2), 0.9, 1), ((1.4, -0.4, 0.5), 0.6, 2), ((-1.5, -0.5, 0.8), 0.5, 3), ((0.7, -0.6, 1.2), 0.4, 4)]
            s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 5)]
        elif which == "eight spheres four planes":
            # the kernarg tables full: twelve primitives with whole materials = 13 classes (4 096 accepted sets sorted on the host)
            cols = [(0.9, 0.3, 0.2), (0.8, 0.8, 0.9), (0.95, 0.95, 1.0), (0.2, 0.7, 0.3), (0.9, 0.8, 0.1), (0.3, 0.3, 0.9), (0.7, 0.2, 0.7), (0.6, 0.6, 0.6)]
            s.materials = [full(rgb=c, roughness=0.1 + 0.1 * i, metallic=float(i % 2), clearcoat=float(i % 3 == 0), clearcoat_gloss=0.5) for i, c in enumerate(cols)] + \
                          [full(rgb=(0.5, 0.5, 0.5), roughness=0.9), full(rgb=(0.7, 0.7, 0.8), roughness=0.5), full(rgb=(0.8, 0.6, 0.5), roughness=0.7), full(rgb=(0.4, 0.6, 0.5), roughness=0.3, metallic=1.0)]
            s.spheres = [((-1.6 + 0.45 * i, -0.3 + 0.25 * (i % 3), -0.4 + 0.35 * (i % 2)), 0.5, i) for i in range(8)]
            s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 8), ((0.0, 0.0, 1.0), (0.0, 0.0, -3.0), 0.0001, 9),
                        ((1.0, 0.0, 0.0), (-3.5, 0.0, 0.0), 0.0001, 10), ((-1.0, 0.0, 0.0), (3.5, 0.0, 0.0), 0.0001, 11)]
        else:
            # eight primitives whose patches each write one field of their own: 2^7 combinations of last writers, far more than 16 classes
            names = ["metallic", "roughness", "subsurface", "sheen", "clearcoat", "specular_tint", "anisotropic"]
            s.materials = [rpt.Material(**{n: 0.7}) for n in names] + [rpt.Material(roughness=0.9, checker_dir=(0.5, 100.0, 0.25, 0.1))]
            s.spheres = [((-1.5 + 0.5 * i, -0.2 + 0.1 * (i % 3), 0.3 * (i % 2)), 0.6, i) for i in range(7)]
            s.planes = [((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), 0.0001, 7)]
    elif which == "one plane two lights":
        s.spheres = []
        s.lights = list(s.lights) + [rpt.AnalyticalLight.spherical((-2.0, 1.5, 1.0), 0.5, (4.0, 4.0, 8.0))]
    else:
        assert which == "reference"
    return s, 0


_TABLE_CASES = ["reference", "overlapping patches", "camera inside glass", "checker on a sphere", "extreme materials", "two checkers", "sdf", "sdf no sphere", "sdf two spheres",
                "sdf checker object", "three spheres", "one sphere two planes", "one plane two lights", "three spheres on a floor", "two spheres two planes", "sdf two lights", "sdf two planes",
                "five spheres on a floor", "six spheres two planes", "six primitives partial patches", "eight primitives many classes", "eight spheres four planes", 2, 5, 9, 13, 17, 21, 26, 33]


def test_the_material_table_holds_what_every_hit_would_compute(rpt, oracle, torch_cuda):
    """The kernels for scenes of the reference's table sizes (megakernel, compacting kernel of one-sample launches) and for an SDF
    object over one plane read what the BSDF code needs of a hit's material — the finalized fields, eta, the specular / sheen
    colours, the lobe weights' numerators, gtr1's constants — from a table of the cases there are (accepted primitives x checker
    colour x side), built once per workgroup (dev_integrator.h, MaterialTable); RPT_NO_MATERIAL_TABLE=1 takes the kernels that
    compute them at every hit.  Same frames, and the oracle's: patches that write different fields on overlapping spheres, a camera
    inside a glass sphere, the procedural material on a sphere or on the SDF object, two of them or four primitives (no table),
    random materials at random scales with and without roulette."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = r"""
import hashlib, sys
sys.path.insert(0, %r)
import conftest, torch
rpt = conftest.load_package()
import test_gpu_dispatch as T
for which in T._TABLE_CASES:
    scene, flags = T._table_scene(rpt, which)
    t = rpt.Tracer(scene, device=0, seed=5)
    t.flags = flags
    buf = rpt.DeviceColorBuffer(176, 96)
    for n in (1, 2, 7):                                               # (1: the compacting kernel)
        t.render_n(buf, n)
    torch.cuda.synchronize()
    print("HASH", which, hashlib.sha1(buf.pixels.cpu().numpy().tobytes()).hexdigest())
    t.close()
""" % here
    out = {}
    for no_table in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, RPT_NO_MATERIAL_TABLE=no_table), timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        out[no_table] = [l for l in r.stdout.splitlines() if l.startswith("HASH")]
        assert len(out[no_table]) == len(_TABLE_CASES)
    assert out["0"] == out["1"], "frames with and without the material table differ: %r vs %r" % (out["0"], out["1"])
    for which in _TABLE_CASES:                                        # and against the oracle, in this process (the table: the default)
        s, flags = _table_scene(rpt, which)
        w, h, spp = 88, 56, 5
        want = oracle.render(s.describe(), w, h, spp, seed=2, render_flags=flags & rpt._abi.RPT_RENDER_RUSSIAN_ROULETTE)
        t = rpt.Tracer(s, device=0, seed=2)
        t.flags = flags
        buf = rpt.DeviceColorBuffer(w, h)
        for n in (1, 1, 3):
            t.render_n(buf, n)
        torch_cuda.cuda.synchronize()
        got = buf.pixels.cpu().numpy()
        t_handle = t._h
        assert_bit_identical(got, want, "material table, scene %r" % (which,))
        # the kernel aimed at is the one that ran (the last launch: 3 samples, the megakernel)
        choice = C.c_uint32()
        assert rpt.lib().rpt_debug_kernel_choice(t_handle, C.byref(choice)) == 0
        # (bits that must be set, bits that must not, classes: None = between 8 and 16)
        expect = {"reference": (1 | 2, 8, 0), "three spheres on a floor": (4, 2 | 8, 0), "five spheres on a floor": (8, 2 | 4, 12), "six spheres two planes": (8, 2 | 4, 15),
                  "six primitives partial patches": (8, 2 | 4, None), "eight primitives many classes": (0, 2 | 4 | 8, 0), "two checkers": (0, 2 | 4 | 8, 0), "eight spheres four planes": (8, 2 | 4, 13)}.get(which)
        if expect is not None:
            assert choice.value & expect[0] == expect[0] and choice.value & expect[1] == 0, (which, hex(choice.value))
            if expect[2] is None:
                assert 7 < (choice.value >> 8) & 0xFF <= 16, (which, hex(choice.value))
            else:
                assert (choice.value >> 8) & 0xFF == expect[2], (which, hex(choice.value))
        t.close()
