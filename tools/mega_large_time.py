"""10 k-sphere scene through the MEGAKERNEL (variant timing of the in-kernel grid walk).  usage: [w h spp]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
w, h, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 2048, 32)
t = rpt.Tracer(scenes.random_spheres_scene(10000, 16), device=0, seed=1)
t.flags = rpt._abi.RPT_RENDER_LARGE_MEGAKERNEL
buf = rpt.DeviceColorBuffer(w, h)
t.render_n(buf, spp); torch.cuda.synchronize()
digest = hashlib.sha1(buf.pixels.cpu().numpy().tobytes()).hexdigest()[:12]
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); t.render_n(buf, spp); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print("megakernel %dx%d x %d spp: %.2f ms -> %.1f Msamples/s  image %s" % (w, h, spp, best * 1e3, w * h * spp / best / 1e6, digest))
