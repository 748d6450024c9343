"""Msamples/s of the c2 frame as a function of the samples folded into one launch (1 = one reference render() per launch)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
buf = rpt.DeviceColorBuffer(w, h)
for spp in (1, 2, 4, 8, 16, 32, 64, 256):
    n = max(3, 256 // spp)
    for _ in range(3):
        t.render_n(buf, spp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        t.render_n(buf, spp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%dx%d x %3d spp per launch: %8.3f ms -> %7.1f Msamples/s" % (w, h, spp, dt * 1e3, w * h * spp / dt / 1e6))
