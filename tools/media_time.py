"""Throughput of scenes with participating media next to the same scenes without (one GPU)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes


def timed(scene, w, h, spp, label):
    t = rpt.Tracer(scene, device=0, seed=1)
    buf = rpt.DeviceColorBuffer(w, h)
    t.render_n(buf, 2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t.render_n(buf, spp); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("%-46s %5dx%-5d x %4d spp: %9.2f ms -> %8.1f Msamples/s" % (label, w, h, spp, ms, w * h * spp / ms / 1e3), flush=True)
    t.close()


timed(rpt.AnalyticalScene(), 1920, 1080, 64, "AnalyticalScene")
timed(scenes.media_scene(), 1920, 1080, 64, "media_scene (fog ball, absorber, glow)")
s = scenes.sdf_scene()
timed(s, 1920, 1080, 32, "SDF scene")
s = scenes.sdf_scene()
s.media = True
s.any_hit_uses_max_dist = True
s.materials[0] = rpt.Material(rgb=(1.0, 1.0, 1.0), roughness=0.05, spec_trans=1.0, ior=1.2,
                              medium=dict(type="scatter", density=0.8, color=(0.9, 0.9, 0.9), anisotropy=0.3))
timed(s, 1920, 1080, 32, "SDF scene, the blob full of fog")
timed(scenes.random_spheres_scene(10000, 16), 2048, 2048, 16, "10k spheres")
timed(scenes.random_spheres_scene(10000, 16, media=True), 2048, 2048, 16, "10k spheres, half the palette with media")
