"""Project-defined extensions of the path, each defined in the oracle first and off by default (needs an MI355X):
Russian roulette (RPT_RENDER_RUSSIAN_ROULETTE) and sampling / intersection of the reference's declared-but-unimplemented
light types (RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES).  Bit-identical to the oracle like everything else."""
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


def _render(rpt, torch, scene, w, h, spp, seed=1, flags=0):
    t = rpt.Tracer(scene, device=0, seed=seed)
    t.flags = flags
    buf = rpt.DeviceColorBuffer(w, h)
    t.render_n(buf, spp)
    torch.cuda.synchronize()
    img = buf.pixels.cpu().numpy()
    t.close()
    return img


def _deep_scene(rpt, depth):
    s = rpt.AnalyticalScene()
    s.max_depth = depth
    # brighter, rougher surfaces so that deep bounces carry energy
    s.materials[0] = rpt.Material(rgb=(0.9, 0.9, 0.9), roughness=0.3, metallic=1.0)
    return s


@pytest.mark.parametrize("depth", [4, 16])
def test_russian_roulette_matches_oracle(rpt, oracle, torch_cuda, depth):
    A = rpt._abi
    s = _deep_scene(rpt, depth)
    w, h, spp = 96, 64, 6
    got = _render(rpt, torch_cuda, s, w, h, spp, flags=A.RPT_RENDER_RUSSIAN_ROULETTE)
    want = oracle.render(s.describe(), w, h, spp, seed=1, render_flags=A.RPT_RENDER_RUSSIAN_ROULETTE)
    assert_bit_identical(got, want, "russian roulette, depth %d" % depth)
    off = oracle.render(s.describe(), w, h, spp, seed=1)
    assert not np.array_equal(want, off), "roulette must change the samples"
    got_nested = _render(rpt, torch_cuda, s, w, h, spp, flags=A.RPT_RENDER_RUSSIAN_ROULETTE | A.RPT_RENDER_NESTED_LOOPS)
    assert_bit_identical(got_nested, want, "russian roulette, nested-loop kernel")


def test_russian_roulette_keeps_the_expectation(rpt, torch_cuda):
    """Same mean image with and without roulette: 2048 spp each on a small frame, compared per channel over the frame
    and per pixel within Monte-Carlo noise."""
    A = rpt._abi
    s = _deep_scene(rpt, 12)
    w, h, spp = 64, 48, 2048
    on = _render(rpt, torch_cuda, s, w, h, spp, seed=5, flags=A.RPT_RENDER_RUSSIAN_ROULETTE)[..., :3].astype(np.float64)
    off = _render(rpt, torch_cuda, s, w, h, spp, seed=6)[..., :3].astype(np.float64)
    off2 = _render(rpt, torch_cuda, s, w, h, spp, seed=7)[..., :3].astype(np.float64)
    noise = np.abs(off - off2).mean()                                   # what two independent estimates of the same image differ by
    assert np.abs(on - off).mean() < 1.5 * noise + 1e-4
    assert np.allclose(on.mean(axis=(0, 1)), off.mean(axis=(0, 1)), rtol=0.01)


def _light_zoo(rpt, flag):
    s = rpt.AnalyticalScene()
    s.sample_all_light_types = flag
    s.any_hit_uses_max_dist = True                                      # a shadow ray towards a distant light has max_dist = inf
    s.lights = [rpt.AnalyticalLight.spherical((3.0, 2.0, 2.0), 1.0, (3.0, 3.0, 3.0)),
                rpt.AnalyticalLight.rectangular((-2.0, 3.0, -1.0), (1.5, 0.0, 0.0), (0.0, 0.0, 1.5), (6.0, 5.0, 4.0)),
                rpt.AnalyticalLight.distant((-1.0, 2.0, 1.5), (0.6, 0.6, 0.7)),
                rpt.AnalyticalLight.rectangular((0.0, 0.5, -3.0), (0.0, 1.5, 0.0), (2.0, 0.0, 0.0), (2.0, 2.0, 4.0))]   # faces the camera: visible
    return s


def test_rectangular_and_distant_lights_match_oracle(rpt, oracle, torch_cuda):
    w, h, spp = 120, 80, 8
    s = _light_zoo(rpt, True)
    got = _render(rpt, torch_cuda, s, w, h, spp)
    want = oracle.render(s.describe(), w, h, spp, seed=1)
    assert_bit_identical(got, want, "light zoo, all types sampled")
    s_off = _light_zoo(rpt, False)
    got_off = _render(rpt, torch_cuda, s_off, w, h, spp)
    want_off = oracle.render(s_off.describe(), w, h, spp, seed=1)
    assert_bit_identical(got_off, want_off, "light zoo, reference behaviour (non-spherical types are no-ops)")
    assert got[..., :3].mean() > 1.1 * got_off[..., :3].mean()          # the extra lights do light the scene


def test_light_types_in_a_large_scene(rpt, oracle, torch_cuda):
    from rust_pathtracer_amd import scenes
    s = scenes.random_spheres_scene(n_spheres=400, n_lights=9)
    s.sample_all_light_types = True
    s.lights[1] = rpt.AnalyticalLight.rectangular((-10.0, 14.0, -40.0), (20.0, 0.0, 0.0), (0.0, 0.0, 20.0), (8.0, 8.0, 8.0))
    s.lights[2] = rpt.AnalyticalLight.distant((0.3, 1.0, 0.4), (1.0, 0.9, 0.8))
    w, h, spp = 96, 64, 4
    got = _render(rpt, torch_cuda, s, w, h, spp, seed=2)
    want = oracle.render(s.describe(), w, h, spp, seed=2)
    assert_bit_identical(got, want, "large scene with rectangular and distant lights")


def test_context_scratch_is_ordered_across_streams(rpt, oracle):
    """The wavefront form's path buffers and the denoiser's intermediate image belong to the CONTEXT, while rpt_render_device and
    rpt_denoise_device run on whatever stream the caller passes: two launches on two streams, nothing ordering them on the
    host side, must both come out right (the second waits on the device for the first one's event)."""
    import torch
    from rust_pathtracer_amd import scenes
    from test_gpu_parity import assert_bit_identical
    s = scenes.random_spheres_scene(n_spheres=400, n_lights=4)
    w, h, spp = 96, 64, 3
    t = rpt.Tracer(s, device=0, seed=5)
    want = oracle.render(s.describe(), w, h, spp, seed=5)
    streams = [torch.cuda.Stream() for _ in range(2)]
    # the megakernel's dispatch tables (tile costs, order, hand-off words: rpt_set_dispatch) and, in A/B builds, the wavefront form's
    # path buffers
    for form, flags in (("megakernel in one-sample chunks", 0),):
        t.flags = flags
        t.set_dispatch(1, 1000, 1, 4)
        bufs = [rpt.DeviceColorBuffer(w, h) for _ in range(4)]
        torch.cuda.synchronize()
        for i, b in enumerate(bufs):
            with torch.cuda.stream(streams[i % 2]):
                t.render_n(b, spp)
        torch.cuda.synchronize()
        for i, b in enumerate(bufs):
            assert_bit_identical(b.pixels.cpu().numpy(), want, "%s, launch %d on stream %d" % (form, i, i % 2))
    dn_want = oracle.denoise(want, w, h, 4, 2.0)
    outs = []
    for i in range(4):
        with torch.cuda.stream(streams[i % 2]):
            outs.append(bufs[i].denoise(4, 2.0))
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert_bit_identical(o.pixels.cpu().numpy(), dn_want, "denoise %d on stream %d" % (i, i % 2))
    t.close()
    # every launch carries its own scene (camera included: it depends on the frame size) in its kernel arguments: two frame sizes
    # of an SDF scene with media alternating on two streams
    s = scenes.sdf_scene()
    s.media = True
    s.any_hit_uses_max_dist = True
    s.materials[0] = rpt.Material(rgb=(1.0, 1.0, 1.0), roughness=0.05, spec_trans=1.0, ior=1.2,
                                  medium=dict(type="scatter", density=0.8, color=(0.9, 0.9, 0.9), anisotropy=0.3))
    t = rpt.Tracer(s, device=0, seed=3)
    sizes = [(160, 90), (96, 128)]
    wants = [oracle.render(s.describe(), ww, hh, 2, seed=3) for ww, hh in sizes]
    bufs = [rpt.DeviceColorBuffer(*sizes[i % 2]) for i in range(6)]
    torch.cuda.synchronize()
    for i, b in enumerate(bufs):
        with torch.cuda.stream(streams[i % 2 if i < 4 else (i + 1) % 2]):
            t.render_n(b, 2)
    torch.cuda.synchronize()
    for i, b in enumerate(bufs):
        assert_bit_identical(b.pixels.cpu().numpy(), wants[i % 2], "SDF + media launch %d" % i)
    t.close()
