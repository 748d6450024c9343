"""Time BASELINE.json configs[3] (SDF scene, 1920x1080x64) or a 10k-sphere frame; used bare under rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
s = scenes.sdf_scene() if which == "c4" else scenes.random_spheres_scene(10000, 16)
w, h, spp = (1920, 1080, 64) if which == "c4" else (2048, 2048, 8)
t = rpt.Tracer(s, device=0, seed=1)
buf = rpt.DeviceColorBuffer(w, h)
t.render_n(buf, spp); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); t.render_n(buf, spp); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print("%s %dx%d x %d spp: %.1f ms -> %.1f Msamples/s  [%s]" % (which, w, h, spp, best * 1e3, w * h * spp / best / 1e6,
      " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("RPT_"))))
