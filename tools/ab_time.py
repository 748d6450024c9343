"""One-line timings for A/B runs of library variants (RPT_LIB) and knobs (RPT_* environment): which = c2 | c2s (32 spp) | c4 | c5 (2048^2 x 32,
megakernel) | c5full (4096^2 x 512: configs[4] in one call).   python tools/ab_time.py c2 [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if which == "dn":                                   # the denoiser: 3 iterations at 1080p and 4K, GB/s of its 32 B per pixel per iteration
    import hashlib
    for w, h in ((1920, 1080), (3840, 2160)):
        buf = rpt.DeviceColorBuffer(w, h)
        torch.manual_seed(1)
        buf.pixels.uniform_(0.0, 2.0)
        out = buf.denoise(3, 2.0)
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps + 5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); buf.denoise(3, 2.0, out=out); e1.record(); e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        best = min(ms)
        print("dn   %dx%d x 3 iterations: best %.4f ms -> %7.1f GB/s  image %s  [%s]" % (w, h, best, 3 * 32.0 * w * h / best / 1e6,
              hashlib.sha1(out.pixels.cpu().numpy().tobytes()).hexdigest()[:10],
              " ".join("%s=%s" % (k, os.path.basename(v)) for k, v in sorted(os.environ.items()) if k.startswith("RPT_"))), flush=True)
    sys.exit(0)
A = rpt._abi
cfg = {"c2": (rpt.AnalyticalScene, 1920, 1080, 256, 0), "c2s": (rpt.AnalyticalScene, 1920, 1080, 32, 0), "c4": (scenes.sdf_scene, 1920, 1080, 64, 0),
       "c5": (lambda: scenes.random_spheres_scene(10000, 16), 2048, 2048, 32, 0),
       "c5l": (lambda: scenes.random_spheres_scene(10000, 16), 1024, 1024, 256, 0),
       "c5full": (lambda: scenes.random_spheres_scene(10000, 16), 4096, 4096, 512, 0),   # BASELINE.json configs[4], one call
       "c5x": (lambda: scenes.random_spheres_scene(10000, 16), 4096, 4096, 8, 0)}[which]
t = rpt.Tracer(cfg[0](), device=0, seed=1)
t.flags = cfg[4]
w, h, spp = cfg[1:4]
buf = rpt.DeviceColorBuffer(w, h)
t.render_n(buf, spp); torch.cuda.synchronize()
ms = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t.render_n(buf, spp); e1.record(); e1.synchronize()
    ms.append(e0.elapsed_time(e1))
best = min(ms)
import hashlib
digest = hashlib.sha1(buf.pixels.cpu().numpy().tobytes()).hexdigest()[:10]
print("%-4s %dx%d x %3d spp: best %.3f ms (median %.3f) -> %8.1f Msamples/s  image %s  [%s]" % (which, w, h, spp, best, sorted(ms)[len(ms) // 2], w * h * spp / best / 1e3, digest,
      " ".join("%s=%s" % (k, os.path.basename(v)) for k, v in sorted(os.environ.items()) if k.startswith("RPT_"))), flush=True)
