"""Throughput of the project-defined BASELINE configs 4 and 5 on one GPU (their single-GPU shares)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes

def timed(scene, w, h, spp, label):
    t = rpt.Tracer(scene, device=0, seed=1)
    buf = rpt.DeviceColorBuffer(w, h)
    t.render_n(buf, 1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t.render_n(buf, spp); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("%-52s %5dx%-5d x %4d spp: %9.1f ms -> %8.1f Msamples/s" % (label, w, h, spp, ms, w * h * spp / ms / 1e3))
    t.close()

timed(rpt.AnalyticalScene(), 800, 600, 1, "c1 AnalyticalScene (one render() call)")
timed(rpt.AnalyticalScene(), 1920, 1080, 256, "c2 AnalyticalScene")
timed(scenes.sdf_scene(), 1920, 1080, 64, "c4 SDF sphere-march scene")
timed(scenes.random_spheres_scene(10000, 16), 4096, 4096, 8, "c5 10k spheres + 16 lights (8 of 512 spp)")
if len(sys.argv) > 1 and sys.argv[1] == "full":
    timed(scenes.random_spheres_scene(10000, 16), 4096, 4096, 512, "c5 10k spheres + 16 lights (the full config)")
