"""A/B of the SDF kernels on BASELINE configs[3] (1920x1080 x 64 spp): wave march, compacting kernel (paths in LDS), inline march
(and the workgroup march pool in -DRPT_AB_KERNELS builds)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
w, h = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
t = rpt.Tracer(scenes.sdf_scene(), device=0, seed=1)
A = rpt._abi
bufs = {}
forms = [("wave", 0), ("compact", A.RPT_RENDER_SDF_COMPACT), ("inline", A.RPT_RENDER_SDF_INLINE_MARCH)]
if os.environ.get("RPT_AB_POOL"):
    forms.insert(0, ("pool", A.RPT_RENDER_SDF_POOL_MARCH))
for name, fl in forms:
    t.flags = fl
    bufs[name] = rpt.DeviceColorBuffer(w, h)
    t.render_n(bufs[name], spp); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); t.render_n(bufs[name], spp); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-7s %dx%d x %d spp: %.1f ms -> %.1f Msamples/s  [%s]" % (name, w, h, spp, best * 1e3, w * h * spp / best / 1e6,
          " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("RPT_"))), flush=True)
print("bit-identical:", all(torch.equal(bufs["wave"].pixels.view(torch.int32), b.pixels.view(torch.int32)) for b in bufs.values()))
