"""configs[3] (SDF scene 1920x1080) through the compacting SDF kernel only (RPT_SDF_COMPACT_STEPS sweeps).  usage: [spp]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
w, h = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
t = rpt.Tracer(scenes.sdf_scene(), device=0, seed=1)
t.flags = rpt._abi.RPT_RENDER_SDF_COMPACT
buf = rpt.DeviceColorBuffer(w, h)
t.render_n(buf, spp); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); t.render_n(buf, spp); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print("compact %dx%d x %d spp: %.1f ms -> %.1f Msamples/s  [%s]" % (w, h, spp, best * 1e3, w * h * spp / best / 1e6,
      " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("RPT_"))), flush=True)
