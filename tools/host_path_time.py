"""PCIe-inclusive rate of the literal drop-in call: rpt_render on a HOST ColorBuffer (H2D 33 MB + kernel +
D2H 33 MB per call), for spp = 1 (one reference render()) and spp = 256.  Never the bench `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest  # noqa: E402

rpt = conftest.load_package()
w, h = 1920, 1080
t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
buf = rpt.ColorBuffer(w, h)
t.render(buf)
for spp, reps in ((1, 20), (256, 3)):
    t0 = time.perf_counter()
    for _ in range(reps):
        t.render_n(buf, spp)
    dt = (time.perf_counter() - t0) / reps
    print("host ColorBuffer, %dx%d x %3d spp per call: %8.2f ms/call -> %8.1f Msamples/s (PCIe-inclusive)" % (w, h, spp, dt * 1e3, w * h * spp / dt / 1e6))
dbuf = rpt.DeviceColorBuffer(w, h)
import torch
t.render(dbuf); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    t.render(dbuf)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print("device ColorBuffer, 1 spp per call (one reference render()): %.3f ms/call -> %.1f Msamples/s" % (dt * 1e3, w * h / dt / 1e6))
for (rw, rh) in ((800, 600), (1920, 1080)):
    t.resident_reset()
    t.render_resident(rw, rh); t.resident_to_u8(rw, rh)
    t0 = time.perf_counter()
    for _ in range(50):
        t.render_resident(rw, rh)
        frame = t.resident_to_u8(rw, rh)
    dt = (time.perf_counter() - t0) / 50
    print("resident buffer, %dx%d: render 1 spp + u8 frame download: %.3f ms/redraw -> %.1f Msamples/s, %.0f redraws/s" % (rw, rh, dt * 1e3, rw * rh / dt / 1e6, 1 / dt))
