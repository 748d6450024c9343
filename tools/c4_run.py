import os, sys
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
