"""Round 5's two pieces of lane scheduling, aimed at directly: lanes of a wave share their pixels' samples (a lane whose pixel is done
renders samples of one that is not; blends stay in sample order), and in SDF scenes an idle lane marches the path ray that waits
behind another lane's shadow ray.  Neither may change a bit: every image here is compared with the oracle's."""
import numpy as np
import pytest

from test_gpu_parity import assert_bit_identical, torch_cuda  # noqa: F401 (the fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,spp,depth,rr", [(21, 13, 33, 12, True), (16, 16, 2, 4, False), (9, 40, 65, 3, False), (64, 8, 17, 30, True)])
def test_small_scene_shared_samples(rpt, oracle, w, h, spp, depth, rr):
    """Pixels of very different cost in one wave (sky beside glass), more samples than lanes stay busy with on their own, ragged
    tiles (lanes without a pixel share nothing), a launch resumed by a second one."""
    A = rpt._abi
    flags = A.RPT_RENDER_RUSSIAN_ROULETTE if rr else 0
    s = rpt.AnalyticalScene()
    s.max_depth = depth
    t = rpt.Tracer(s, device=0, seed=21)
    t.flags = flags
    buf = rpt.ColorBuffer(w, h)
    t.render_n(buf, spp)
    t.render_n(buf, 3)
    want = oracle.render(s.describe(), w, h, spp + 3, seed=21, render_flags=flags)
    assert_bit_identical(buf.image(), want, "shared samples %dx%d x %d depth %d" % (w, h, spp, depth))
    t.close()


@pytest.mark.parametrize("w,h,spp,depth", [(40, 24, 130, 4), (33, 17, 5, 1), (24, 24, 9, 2), (50, 11, 97, 3)])
def test_sdf_scene_shared_samples_and_helped_marches(rpt, oracle, w, h, spp, depth):
    """More samples than one launch of the SDF kernels takes (96: the call is cut in two), paths that end at their first bounce with a
    shadow ray still to march (depth 1), two lights, ragged tiles; with and without any_hit's max_dist (shadow marches stop early)."""
    from rust_pathtracer_amd import scenes
    for use_max in (False, True):
        s = scenes.sdf_scene()
        s.max_depth = depth
        s.any_hit_uses_max_dist = use_max
        t = rpt.Tracer(s, device=0, seed=5)
        buf = rpt.ColorBuffer(w, h)
        t.render_n(buf, spp)
        want = oracle.render(s.describe(), w, h, spp, seed=5)
        assert_bit_identical(buf.image(), want, "sdf sharing %dx%d x %d depth %d use_max %s" % (w, h, spp, depth, use_max))
        t.close()


def test_sdf_general_kernel_helped_marches(rpt, oracle, monkeypatch):
    """The same scene through the kernel that reads the primitives' records at every step (sizes as data): the helpers are its too."""
    from rust_pathtracer_amd import scenes
    monkeypatch.setenv("RPT_NO_SIZED_KERNELS", "1")
    monkeypatch.setenv("RPT_NO_MATERIAL_TABLE", "1")
    rpt.lib().rpt_debug_reload_knobs()
    s = scenes.sdf_scene()
    t = rpt.Tracer(s, device=0, seed=6)
    buf = rpt.ColorBuffer(48, 30)
    t.render_n(buf, 12)
    want = oracle.render(s.describe(), 48, 30, 12, seed=6)
    assert_bit_identical(buf.image(), want, "sdf general kernel")
    t.close()


@pytest.mark.parametrize("w,h,spp", [(30, 20, 25), (16, 16, 3)])
def test_large_scene_shared_samples(rpt, oracle, w, h, spp):
    from rust_pathtracer_amd import scenes
    s = scenes.random_spheres_scene(400, 4)
    s.max_depth = 6
    t = rpt.Tracer(s, device=0, seed=8)
    buf = rpt.ColorBuffer(w, h)
    t.render_n(buf, spp)
    t.render_n(buf, 2)
    want = oracle.render(s.describe(), w, h, spp + 2, seed=8)
    assert_bit_identical(buf.image(), want, "large scene sharing %dx%d x %d" % (w, h, spp))
    t.close()


def test_the_same_frame_twice(rpt, torch_cuda):
    """Which lane renders a sample depends on timing; the frame must not."""
    from rust_pathtracer_amd import scenes
    for make in (rpt.AnalyticalScene, scenes.sdf_scene, lambda: scenes.random_spheres_scene(2000, 8)):
        images = []
        for _ in range(3):
            t = rpt.Tracer(make(), device=0, seed=3)
            buf = rpt.DeviceColorBuffer(320, 200)
            t.render_n(buf, 24)
            torch_cuda.cuda.synchronize()
            images.append(buf.pixels.cpu().numpy().copy())
            t.close()
        assert np.array_equal(images[0].view(np.uint32), images[1].view(np.uint32))
        assert np.array_equal(images[0].view(np.uint32), images[2].view(np.uint32))


@pytest.mark.parametrize("knobs", [
    {"RPT_SHADE_THRESHOLD": "1", "RPT_FINISH_THRESHOLD": "1", "RPT_SDF_MARCH_MIN_LANES": "1"},
    {"RPT_SHADE_THRESHOLD": "64", "RPT_FINISH_THRESHOLD": "64", "RPT_SDF_MARCH_MIN_LANES": "64"},
    {"RPT_SHADE_THRESHOLD": "64", "RPT_FINISH_THRESHOLD": "1", "RPT_SDF_MARCH_MIN_LANES": "33"},
    {"RPT_UNIT_ROUNDS": "100000", "RPT_UNIT_MIN_SPP": "1"},          # every sample a chunk of its own, handed from workgroup to workgroup
    {"RPT_UNIT_ROUNDS": "100000", "RPT_UNIT_MIN_SPP": "3", "RPT_DISPATCH_ORDER": "0"},
])
def test_the_scheduling_knobs_at_their_ends(rpt, oracle, monkeypatch, knobs):
    """Blocked lanes, helpers and hand-offs under every vote the kernels can take: thresholds at 1 and at 64 (a room runs as soon as
    one lane waits / only when all do), chunks of one sample.  None of it may change a pixel."""
    from rust_pathtracer_amd import scenes
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    rpt.lib().rpt_debug_reload_knobs()
    for make, w, h, spp in ((rpt.AnalyticalScene, 37, 21, 11), (scenes.sdf_scene, 29, 18, 7), (lambda: scenes.random_spheres_scene(300, 3), 24, 20, 6)):
        s = make()
        t = rpt.Tracer(s, device=0, seed=12)
        buf = rpt.ColorBuffer(w, h)
        t.render_n(buf, spp)
        want = oracle.render(s.describe(), w, h, spp, seed=12)
        assert_bit_identical(buf.image(), want, "knobs %s, %s" % (knobs, type(s).__name__))
        t.close()
