"""Per-function device probes against the oracle (needs an MI355X): the integrator's building blocks evaluated one
record per lane through rpt_probe_fn and compared bit for bit with the oracle's entry points for the same function, so
that a frame mismatch can be localised.  1e5 random records each, including the degenerate inputs the reference does not
guard (NaN hygiene): grazing views (v.z -> 0), a black dielectric with eta = 1 (total lobe weight 0 ->
NaN weights), zero-length vectors, rays starting inside spheres, parallel rays."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_parity import assert_bit_identical

pytestmark = pytest.mark.gpu
N = 100000


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


@pytest.fixture(scope="module")
def tracer(rpt, torch_cuda):
    t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
    yield t
    t.close()


def device_probe(rpt, torch, tracer, fn, records, params=None):
    A = rpt._abi
    rec = torch.from_numpy(np.ascontiguousarray(records, dtype=np.float32)).cuda()
    out = torch.empty(rec.shape[0], A.RPT_PROBE_OUT_STRIDE, dtype=torch.float32, device="cuda")
    p = np.ascontiguousarray(params, dtype=np.float32) if params is not None else None
    stream = torch.cuda.current_stream().cuda_stream
    rpt._lib.check(rpt.lib().rpt_probe_fn(tracer._h, fn, rec.data_ptr(), out.data_ptr(), rec.shape[0],
                                          p.ctypes.data if p is not None else None, C.c_void_p(stream)), tracer._h)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def records(n):
    return np.zeros((n, 32), dtype=np.float32)


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def u32_as_f32(a):
    return np.asarray(a, dtype=np.uint32).view(np.float32)


def materials(rng, n):
    """17 user-set material floats per record, with the corner cases mixed in."""
    m = np.zeros((n, 17), dtype=np.float32)
    m[:, 0:3] = rng.uniform(0.0, 1.0, size=(n, 3))
    m[:, 6] = rng.choice([0.0, 0.0, 0.5, 0.95, 1.0], size=n)            # anisotropic
    m[:, 7] = rng.choice([0.0, 0.0, 1.0, 0.4], size=n)                  # metallic
    m[:, 8] = rng.choice([0.0, 0.02, 0.1, 0.5, 1.0], size=n) * rng.uniform(0.5, 1.0, size=n)   # roughness
    m[:, 9] = rng.choice([0.0, 0.0, 0.7], size=n)                       # subsurface
    m[:, 10] = rng.choice([0.0, 0.5], size=n)                           # specular_tint
    m[:, 11] = rng.choice([0.0, 0.0, 1.0], size=n)                      # sheen
    m[:, 12] = rng.choice([0.0, 0.5], size=n)                           # sheen_tint
    m[:, 13] = rng.choice([0.0, 0.0, 1.0, 0.3], size=n)                 # clearcoat
    m[:, 14] = rng.uniform(0.0, 1.0, size=n)                            # clearcoat_gloss
    m[:, 15] = rng.choice([0.0, 0.0, 0.0, 1.0, 0.5], size=n)            # spec_trans
    m[:, 16] = rng.choice([1.45, 1.5, 1.0, 1.33], size=n)               # ior
    black = rng.uniform(size=n) < 0.03                                  # black dielectric: with eta = 1 every lobe weight is 0
    m[black, 0:3] = 0.0
    m[black, 7] = 0.0
    return m, black


def test_probe_gen_ray(rpt, oracle, torch_cuda, tracer):
    rng = np.random.default_rng(11)
    for cam, (w, h) in (((0, 0, 3, 0, 0, 0, 80), (800, 600)), ((0, 6, 14, 0, 2, -40, 70), (4096, 4096)), ((2, 1, -3, 0.5, 0, 1, 35), (1920, 1080))):
        s = rpt.AnalyticalScene()
        s.camera = rpt.Pinhole(cam[0:3], cam[3:6], cam[6])
        t = rpt.Tracer(s, device=0)
        rec = records(N)
        rec[:, 0:2] = rng.uniform(0.0, 1.0, size=(N, 2))
        rec[:, 2:4] = rng.uniform(0.0, 1.0, size=(N, 2))
        rec[:16, 0:4] = 0.0
        got = device_probe(rpt, torch_cuda, t, rpt._abi.RPT_PROBE_FN_GEN_RAY, rec, params=(w, h))
        want = oracle.probe_fn(rpt._abi.RPT_PROBE_FN_GEN_RAY, rec, cam=cam, params=(w, h))
        assert_bit_identical(got, want, "gen_ray %r %dx%d" % (cam, w, h))
        assert np.abs(np.linalg.norm(got[:, 3:6], axis=1) - 1.0).max() < 1e-5
        t.close()


def test_probe_hit_sphere_and_plane(rpt, oracle, torch_cuda, tracer):
    rng = np.random.default_rng(12)
    rec = records(N)
    rec[:, 0:3] = rng.uniform(-5, 5, size=(N, 3))
    rec[:, 3:6] = unit(rng, N)
    rec[:, 6:9] = rng.uniform(-5, 5, size=(N, 3))
    rec[:, 9] = rng.choice([0.3, 1.0, 4.0, 9.0, 0.0], size=N)          # large radii: origins inside the sphere (far root)
    far = rng.uniform(size=N) < 0.1                                      # far origins: d2 = l.l - tca^2 cancels
    rec[far, 0:3] *= 1000.0
    graze = rng.uniform(size=N) < 0.1                                    # aim at the silhouette
    c = rec[graze, 6:9] - rec[graze, 0:3]
    dist = np.linalg.norm(c, axis=1, keepdims=True)
    side = np.cross(c, unit(rng, int(graze.sum())))
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    aim = c + side * rec[graze, 9:10] * rng.choice([0.999999, 1.0, 1.000001], size=(int(graze.sum()), 1))
    rec[graze, 3:6] = (aim / np.linalg.norm(aim, axis=1, keepdims=True)).astype(np.float32)
    got = device_probe(rpt, torch_cuda, tracer, rpt._abi.RPT_PROBE_FN_HIT_SPHERE, rec)
    want = oracle.probe_fn(rpt._abi.RPT_PROBE_FN_HIT_SPHERE, rec)
    assert_bit_identical(got, want, "hit_sphere")
    assert 0.05 < got[:, 0].mean() < 0.95

    rec = records(N)
    rec[:, 0:3] = rng.uniform(-5, 5, size=(N, 3))
    rec[:, 3:6] = unit(rng, N)
    rec[:, 6:9] = unit(rng, N)
    rec[:, 9:12] = rng.uniform(-2, 2, size=(N, 3))
    rec[:, 12] = 0.0001
    rec[:, 13] = rng.choice([0.0, 0.0, 3.0, 400.0], size=N)
    par = rng.uniform(size=N) < 0.05                                     # (nearly) parallel rays: |n.d| around min_denom
    d = np.cross(rec[par, 6:9], unit(rng, int(par.sum())))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rec[par, 3:6] = (d + rec[par, 6:9] * rng.choice([0.0, 1e-4, 9.99e-5, 1.01e-4, -1e-4], size=(int(par.sum()), 1))).astype(np.float32)
    got = device_probe(rpt, torch_cuda, tracer, rpt._abi.RPT_PROBE_FN_HIT_PLANE, rec)
    want = oracle.probe_fn(rpt._abi.RPT_PROBE_FN_HIT_PLANE, rec)
    assert_bit_identical(got, want, "hit_plane")


def test_probe_sample_light(rpt, oracle, torch_cuda, tracer):
    A = rpt._abi
    rng = np.random.default_rng(13)
    rec = records(N)
    types = rng.choice([A.RPT_LIGHT_SPHERICAL, A.RPT_LIGHT_SPHERICAL, A.RPT_LIGHT_RECTANGULAR, A.RPT_LIGHT_DISTANT], size=N)
    rec[:, 0] = u32_as_f32(types)
    rec[:, 1:4] = rng.uniform(-4, 4, size=(N, 3))
    rec[:, 4:7] = rng.uniform(0, 5, size=(N, 3))
    rec[:, 7] = rng.choice([1.0, 0.25, 2.0], size=N)
    rec[:, 8] = (4.0 * np.pi * rec[:, 7] ** 2).astype(np.float32)
    rec[:, 9:12] = rng.uniform(-2, 2, size=(N, 3))
    rec[:, 12:15] = rng.uniform(-2, 2, size=(N, 3))
    rect = types == A.RPT_LIGHT_RECTANGULAR
    rec[rect, 8] = np.linalg.norm(np.cross(rec[rect, 9:12], rec[rect, 12:15]), axis=1).astype(np.float32)
    rec[types == A.RPT_LIGHT_DISTANT, 8] = 0.0
    rec[:, 15:18] = rng.uniform(-6, 6, size=(N, 3))
    on_axis = rng.uniform(size=N) < 0.05                                 # scatter point straight above / below the light: the onb's other branch
    rec[on_axis, 15:17] = rec[on_axis, 1:3]
    rec[:, 18] = rng.choice([1.0, 3.0, 16.0], size=N)
    rec[:, 19] = u32_as_f32(rng.choice([0, A.RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES, A.RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES], size=N))
    rec[:, 20] = u32_as_f32(rng.integers(0, 2 ** 32, size=N, dtype=np.uint64))
    rec[:, 21] = u32_as_f32(rng.integers(0, 2 ** 22, size=N, dtype=np.uint64))
    rec[:, 22] = u32_as_f32(rng.integers(0, 40, size=N, dtype=np.uint64))
    got = device_probe(rpt, torch_cuda, tracer, A.RPT_PROBE_FN_SAMPLE_LIGHT, rec)
    want = oracle.probe_fn(A.RPT_PROBE_FN_SAMPLE_LIGHT, rec)
    assert_bit_identical(got, want, "sample_light")
    flags_on = rec[:, 19].view(np.uint32) != 0
    assert np.all(got[(types == A.RPT_LIGHT_SPHERICAL), 11] == 2.0)
    assert np.all(got[rect & flags_on, 11] == 2.0) and np.all(got[rect & ~flags_on, 11] == 0.0)
    assert np.all(got[(types == A.RPT_LIGHT_DISTANT), 11] == 0.0)
    assert np.all(got[rect & ~flags_on, 0:11] == 0.0)                    # tracer.rs:217: nothing happens


def _bsdf_records(rng, n):
    rec = records(n)
    m, black = materials(rng, n)
    rec[:, 0:17] = m
    front = rng.uniform(size=n) < 0.85
    rec[:, 17] = np.where(front, 1.0 / m[:, 16], m[:, 16])              # State::finalize, globals.rs:60
    nrm = unit(rng, n)
    axis = rng.uniform(size=n) < 0.1                                     # |n.z| >= 0.999: the onb's other branch
    nrm[axis] = np.array([0.0, 0.0, 1.0], dtype=np.float32) * rng.choice([-1.0, 1.0], size=(int(axis.sum()), 1))
    v = unit(rng, n)
    flip = (v * nrm).sum(axis=1) < 0
    v[flip] = -v[flip]                                                   # the viewer is on the normal's side (ffnormal)
    graze = rng.uniform(size=n) < 0.08                                   # grazing view: v.z -> 0, and exactly 0
    tang = np.cross(nrm[graze], unit(rng, int(graze.sum())))
    tang /= np.linalg.norm(tang, axis=1, keepdims=True)
    v[graze] = (tang + nrm[graze] * rng.choice([0.0, 1e-7, 1e-4, 1e-3], size=(int(graze.sum()), 1))).astype(np.float32)
    v[graze] /= np.linalg.norm(v[graze], axis=1, keepdims=True)
    rec[:, 18:21] = v
    rec[:, 21:24] = nrm
    return rec, black


def test_probe_disney_eval(rpt, oracle, torch_cuda, tracer):
    A = rpt._abi
    rng = np.random.default_rng(14)
    rec, black = _bsdf_records(rng, N)
    rec[:, 24:27] = unit(rng, N)                                         # l: both hemispheres (refraction side too)
    degenerate = rng.uniform(size=N) < 0.02
    rec[degenerate, 24:27] = -rec[degenerate, 18:21]                     # l = -v: h = normalize(0) = NaN
    got = device_probe(rpt, torch_cuda, tracer, A.RPT_PROBE_FN_DISNEY_EVAL, rec)
    want = oracle.probe_fn(A.RPT_PROBE_FN_DISNEY_EVAL, rec)
    assert_bit_identical(got, want, "disney_eval")
    assert np.isnan(want[:, 3]).sum() > 0 and np.isfinite(want[:, 3]).mean() > 0.8     # the NaN cases are there, and are the exception


def test_probe_disney_sample(rpt, oracle, torch_cuda, tracer):
    A = rpt._abi
    rng = np.random.default_rng(15)
    rec, black = _bsdf_records(rng, N)
    stale = unit(rng, N)
    first = rng.uniform(size=N) < 0.4                                    # first bounce: the stale l is zeros (tracer.rs:55)
    stale[first] = 0.0
    rec[:, 24:27] = stale
    rec[:, 27] = u32_as_f32(rng.integers(0, 2 ** 32, size=N, dtype=np.uint64))
    rec[:, 28] = u32_as_f32(rng.integers(0, 2 ** 22, size=N, dtype=np.uint64))
    rec[:, 29] = u32_as_f32(rng.integers(0, 40, size=N, dtype=np.uint64))
    eta_one = black & (rng.uniform(size=N) < 0.5)                        # total lobe weight 0 -> NaN weights (tracer.rs:427-432)
    rec[eta_one, 16] = 1.0
    rec[eta_one, 17] = 1.0
    got = device_probe(rpt, torch_cuda, tracer, A.RPT_PROBE_FN_DISNEY_SAMPLE, rec)
    want = oracle.probe_fn(A.RPT_PROBE_FN_DISNEY_SAMPLE, rec)
    assert_bit_identical(got, want, "disney_sample")
    assert set(np.unique(want[:, 7])) <= {2.0, 3.0} and (want[:, 7] == 3.0).mean() > 0.1   # 2 draws, 3 in the specular arm
    dead = eta_one & (rec[:, 13] == 0.0)                                 # ... when there is no clearcoat lobe either
    assert dead.sum() > 100 and np.isnan(want[dead][:, 6]).mean() > 0.3    # the unguarded 0/0 cases are there (and bit-identical above)


def test_first_probe_launch_in_fresh_processes(rpt, torch_cuda):
    """once in ~25 runs of this suite the FIRST probe launch of the process (test_probe_gen_ray above: the launch
    that loads the probe kernels' code object) ended in SIGABRT, with nothing on stderr.  tools/probe_first_launch.py is exactly that
    launch in a process of its own; profiles/r6/abort_hunt.txt holds 250 of them in a row (150 plain, 50 with HIP_LAUNCH_BLOCKING=1, 50
    with AMD_LOG_LEVEL=3) without a failure, so the first launch by itself is not the cause.  This keeps 20 of them in the suite: an
    abort here would be that bug, caught with its stderr."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AMD_LOG_LEVEL="1")
    for i in range(20):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "probe_first_launch.py"), "gen_ray" if i % 2 == 0 else "math"],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0 and "FIRST-LAUNCH OK" in r.stdout, "run %d: rc %d\n%s\n%s" % (i, r.returncode, r.stdout[-2000:], r.stderr[-6000:])
