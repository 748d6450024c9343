"""10 k-sphere scene (BASELINE configs[4]): megakernel vs wavefront form — whole-frame bit equality and time.
usage: wavefront_time.py [width height spp]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
w, h, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2048, 2048, 8)
s = scenes.random_spheres_scene(10000, 16)
t = rpt.Tracer(s, device=0, seed=1)
imgs = {}
for name, flags in (("megakernel", rpt._abi.RPT_RENDER_LARGE_MEGAKERNEL), ("wavefront", rpt._abi.RPT_RENDER_LARGE_WAVEFRONT)):
    t.flags = flags
    buf = rpt.DeviceColorBuffer(w, h)
    t.render_n(buf, spp); torch.cuda.synchronize()          # warm (allocations)
    best = 1e9
    for _ in range(3):
        buf = rpt.DeviceColorBuffer(w, h)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t.render_n(buf, spp)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    imgs[name] = buf.pixels.clone()
    print("%-10s %dx%d x %d spp: %.2f ms -> %.1f Msamples/s" % (name, w, h, spp, best * 1e3, w * h * spp / best / 1e6), flush=True)
same = torch.equal(imgs["megakernel"].view(torch.int32), imgs["wavefront"].view(torch.int32))
print("whole frame bit-identical:", same)
sys.exit(0 if same else 1)
