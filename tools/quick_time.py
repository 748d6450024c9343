"""A/B timing of the render kernels on 1920x1080 (interleaved rounds, one process)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest  # noqa: E402
import torch  # noqa: E402

rpt = conftest.load_package()
w, h = 1920, 1080
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
variants = {"regen": 0, "nested": rpt._abi.RPT_RENDER_NESTED_LOOPS, "fast": rpt._abi.RPT_RENDER_FAST_MATH}
if len(sys.argv) > 2:
    variants = {k: v for k, v in variants.items() if k in sys.argv[2].split(",")}
t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
bufs = {k: rpt.DeviceColorBuffer(w, h) for k in variants}
for k, fl in variants.items():
    t.flags = fl
    t.render_n(bufs[k], 2)
torch.cuda.synchronize()
for rep in range(3):
    for k, fl in variants.items():
        t.flags = fl
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t.render_n(bufs[k], spp); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print("%-7s %dx%d x %d spp: %8.2f ms -> %9.1f Msamples/s" % (k, w, h, spp, ms, w * h * spp / ms / 1e3))
if "regen" in bufs and "nested" in bufs:
    a, b = bufs["regen"].pixels, bufs["nested"].pixels
    same = (a.view(torch.int32) == b.view(torch.int32)).all().item()
    print("regen vs nested bit-identical:", same)
