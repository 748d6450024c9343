"""One reference render() per launch (spp = 1) on a device-resident frame: per-call wall time; used bare under rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
for w, h in ((800, 600), (1920, 1080), (3840, 2160)):
    t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
    buf = rpt.DeviceColorBuffer(w, h)
    for _ in range(20):
        t.render(buf)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        t.render(buf)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%dx%d, 1 spp per call: %.3f ms per call -> %.1f Msamples/s" % (w, h, dt * 1e3, w * h / dt / 1e6))
    t.close()
