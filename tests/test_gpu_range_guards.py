"""The short divide / square root (csrc/dev_math.h) are correct for operands in [2^-60, 2^60) only.  Small scenes' megakernel does not
test the operands next to every operation: it tracks them and looks once per sample; a sample that saw an operand outside the range
is computed again from its camera ray with hipcc's own divide and sqrtf (kernels.hip, sample_guard; namespace rptplain).  These
scenes force that: the stock scene scaled so far down (up) that squared lengths leave the range — for every sample, or for some.
The oracle divides and takes roots in IEEE arithmetic everywhere, so the frames must still match it bit for bit.  Needs an MI355X."""
import numpy as np
import pytest

from test_gpu_parity import assert_bit_identical

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def scaled_stock_scene(rpt, k, media=False):
    """The reference's scene (analytical.rs) with every length multiplied by k (a power of two: the geometry is the same up to
    rounding at the ends of the exponent range, the arithmetic is not)."""
    k = float(k)
    s = rpt.AnalyticalScene()
    mul = lambda v: tuple(float(np.float32(x) * np.float32(k)) for x in v)    # noqa: E731
    s.camera.set(mul(s.camera.origin), mul(s.camera.center))
    s.spheres = [(mul(c), float(np.float32(r) * np.float32(k)), m) for c, r, m in s.spheres]
    s.planes = [(n, mul(p), md, m) for n, p, md, m in s.planes]
    s.lights = [rpt.AnalyticalLight.spherical(mul(L.position), float(np.float32(L.radius) * np.float32(k)), L.emission) for L in s.lights]
    s.eps = float(np.float32(s.eps) * np.float32(k))
    return s


def _render(rpt, torch, scene, w, h, spp, flags=0):
    t = rpt.Tracer(scene, device=0, seed=1)
    t.flags = flags
    buf = rpt.DeviceColorBuffer(w, h)
    t.render_n(buf, spp)
    torch.cuda.synchronize()
    img = buf.pixels.cpu().numpy()
    t.close()
    return img


# 2^-31 / 2^31: every squared length is outside [2^-60, 2^60): every sample is recomputed.  2^-30, 2^-29, 2^29, 2^30: some are.
# 2^-20: none (the control: the same code with the second computation never taken).
@pytest.mark.parametrize("log2_k", [-31, -30, -29, -20, 29, 30, 31])
def test_scaled_scenes_match_the_oracle(rpt, oracle, torch_cuda, log2_k):
    w, h, spp = 96, 64, 6
    s = scaled_stock_scene(rpt, 2.0 ** log2_k)
    want = oracle.render(s.describe(), w, h, spp, seed=1)
    assert np.isfinite(want).all() and want[..., :3].std() > 0.01, "the scaled scene still renders a picture"
    got = _render(rpt, torch_cuda, s, w, h, spp)
    assert_bit_identical(got, want, "stock scene x 2^%d, megakernel" % log2_k)


@pytest.mark.parametrize("log2_k", [-31, 30])
def test_scaled_scenes_in_the_other_small_scene_kernels(rpt, oracle, torch_cuda, log2_k):
    """The nested-loop kernel (same trackers, same second computation) and the compacting kernel of one-sample launches (tests next to
    every operation) on the same scenes, and progressive steps: a recomputed sample must leave the pixel's later samples alone."""
    A = rpt._abi
    w, h = 80, 48
    s = scaled_stock_scene(rpt, 2.0 ** log2_k)
    want = oracle.render(s.describe(), w, h, 5, seed=1)
    got = _render(rpt, torch_cuda, s, w, h, 5, flags=A.RPT_RENDER_NESTED_LOOPS)
    assert_bit_identical(got, want, "x 2^%d, nested loops" % log2_k)
    t = rpt.Tracer(s, device=0, seed=1)
    buf = rpt.DeviceColorBuffer(w, h)
    for n in (1, 1, 3):                                               # 1 spp: the compacting kernel; 3: the megakernel resumes the same pixels
        t.render_n(buf, n)
    torch_cuda.cuda.synchronize()
    assert_bit_identical(buf.pixels.cpu().numpy(), want, "x 2^%d, 1 + 1 + 3 samples" % log2_k)
    t.close()


def test_a_scaled_scene_with_chunks_and_roulette(rpt, oracle, torch_cuda):
    """The second computation inside chunked launches (units handed from workgroup to workgroup) and with Russian roulette's extra draws."""
    A = rpt._abi
    w, h, spp = 112, 80, 9
    s = scaled_stock_scene(rpt, 2.0 ** -30)
    want = oracle.render(s.describe(), w, h, spp, seed=1, render_flags=A.RPT_RENDER_RUSSIAN_ROULETTE)
    t = rpt.Tracer(s, device=0, seed=1)
    t.flags = A.RPT_RENDER_RUSSIAN_ROULETTE
    t.set_dispatch(1, 100000, 1, 8)                                   # every sample its own chunk
    buf = rpt.DeviceColorBuffer(w, h)
    t.render_n(buf, spp)
    torch_cuda.cuda.synchronize()
    assert_bit_identical(buf.pixels.cpu().numpy(), want, "x 2^-30, one-sample chunks, roulette")
    t.close()


@pytest.mark.parametrize("seed", [3, 7, 11, 14, 23, 31, 42, 57])
def test_random_scaled_scenes_match_the_oracle(rpt, oracle, torch_cuda, seed):
    """Random small scenes (glass, clearcoat, metal, several lights, depth up to 8, roulette) scaled by a random power of two in
    2^-33 ... 2^33 (tests/scene_fuzz.py): whatever share of the samples takes the second computation, the frame is the oracle's.
    (tools/range_soak.py runs 60 of these at larger sizes against the library with per-operation tests.)"""
    from scene_fuzz import random_small_scene
    s, log2_k, flags, rng = random_small_scene(rpt, seed)
    w, h, spp = 72, 48, 5
    want = oracle.render(s.describe(), w, h, spp, seed=seed, render_flags=flags & rpt._abi.RPT_RENDER_RUSSIAN_ROULETTE)
    t = rpt.Tracer(s, device=0, seed=seed)
    t.flags = flags
    buf = rpt.DeviceColorBuffer(w, h)
    t.render_n(buf, 2)
    t.render_n(buf, 3)
    torch_cuda.cuda.synchronize()
    got = buf.pixels.cpu().numpy()
    t.close()
    assert_bit_identical(got, want, "random scene %d x 2^%d" % (seed, log2_k))
