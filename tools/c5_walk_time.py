"""A/B of the large-scene kernels on the 10k-sphere scene: walk inside the bounce (default) vs resumable walk."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
w = h = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t = rpt.Tracer(scenes.random_spheres_scene(10000, 16), device=0, seed=1)
bufs = {}
for name, fl in (("walk", rpt._abi.RPT_RENDER_GRID_RESUMABLE_WALK), ("inline", 0)):
    t.flags = fl
    bufs[name] = rpt.DeviceColorBuffer(w, h)
    t.render_n(bufs[name], spp); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); t.render_n(bufs[name], spp); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-7s %dx%d x %d spp: %.1f ms -> %.1f Msamples/s  [%s]" % (name, w, h, spp, best * 1e3, w * h * spp / best / 1e6,
          " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("RPT_"))))
print("bit-identical:", torch.equal(bufs["walk"].pixels.view(torch.int32), bufs["inline"].pixels.view(torch.int32)))
