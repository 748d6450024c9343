"""One 10 k-sphere frame (2048^2 x 8 spp unless given) for profilers; RPT_LARGE_FORM=wavefront|megakernel picks the form."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
w, h, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 2048, 8)
t = rpt.Tracer(scenes.random_spheres_scene(10000, 16), device=0, seed=1)
buf = rpt.DeviceColorBuffer(w, h)
t.render_n(buf, spp)
torch.cuda.synchronize()
print("frames", buf.frames)
