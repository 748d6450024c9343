"""Cost per sample of coherent views (all sky / all floor / one sphere close-up) vs the stock view."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest  # noqa: E402
import torch  # noqa: E402

rpt = conftest.load_package()
w, h, spp = 1920, 1080, 32
views = {
    "stock": ((0, 0, 3), (0, 0, 0), 80),
    "sky": ((0, 0, 3), (0, 3, 0), 40),
    "floor": ((0, 3, 21), (0, -1, 20), 40),
    "metal_sphere": ((-1.1, 0, 3), (-1.1, 0, 0), 25),
    "coat_sphere": ((1.1, 0, 3), (1.1, 0, 0), 25),
}
for flagname, fl in (("regen", 0), ("nested", 1)):
    for name, (o, c, fov) in views.items():
        sc = rpt.AnalyticalScene()
        sc.camera = rpt.Pinhole(o, c, fov)
        t = rpt.Tracer(sc, device=0, seed=1)
        t.flags = fl
        buf = rpt.DeviceColorBuffer(w, h)
        t.render_n(buf, 2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t.render_n(buf, spp); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        px = buf.pixels
        print("%-7s %-13s %8.2f ms  %9.1f Msamples/s  mean %.4f" % (flagname, name, ms, w * h * spp / ms / 1e3, px[..., :3].mean().item()))
        t.close()
