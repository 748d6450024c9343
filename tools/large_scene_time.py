import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
w, h, spp = 1024, 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_lights = int(sys.argv[3]) if len(sys.argv) > 3 else 16
for name, cam in (("horizon view", None), ("looking down (no horizon)", rpt.Pinhole((0.0, 40.0, 10.0), (0.0, 0.0, -50.0), 60.0))):
    s = scenes.random_spheres_scene(n_spheres=n, n_lights=n_lights)
    if cam is not None:
        s.camera = cam
    t = rpt.Tracer(s, device=0, seed=5)
    buf = rpt.DeviceColorBuffer(w, h)
    t.render_n(buf, 1); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t.render_n(buf, spp); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("%d spheres, %d lights, %s, %dx%d x %d spp: %.1f ms -> %.1f Msamples/s" % (n, n_lights, name, w, h, spp, ms, w*h*spp/ms/1e3))
    t.close()
