"""Throughput of the headline kernel against the size of one launch (workgroups per launch / workgroup slots of the chip = rounds):
what the launch tail costs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
spp = 128
for w, h in ((3840, 2160), (1920, 1080), (3840, 270), (1920, 540), (1280, 720), (960, 540), (800, 600), (640, 360)):
    buf = rpt.DeviceColorBuffer(w, h)
    t.render_n(buf, 8); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t.render_n(buf, spp); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    wgs = ((w + 15) // 16) * ((h + 15) // 16)
    print("%4dx%-4d x %d spp: %6d workgroups = %5.2f rounds of 1280: %7.2f ms -> %7.1f Msamples/s" % (w, h, spp, wgs, wgs / 1280.0, best, w * h * spp / best / 1e3), flush=True)
