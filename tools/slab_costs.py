"""Time the stock 1920x1080 view in horizontal slabs (uses the row-tiling entry point)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest  # noqa: E402
import torch  # noqa: E402

rpt = conftest.load_package()
w, h, spp = 1920, 1080, 32
nslab = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = h // nslab
t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
tile = torch.zeros(rows, w, 4, dtype=torch.float32, device="cuda")
tot = 0.0
for r in range(nslab):
    t.render_tile(tile, w, h, 0, 2, rows, r, nslab)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t.render_tile(tile, w, h, 2, spp, rows, r, nslab); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    tot += ms
    print("slab %d rows %4d-%4d: %7.3f ms  %9.1f Msamples/s" % (r, r * rows, (r + 1) * rows - 1, ms, w * rows * spp / ms / 1e3))
full = rpt.DeviceColorBuffer(w, h)
t.render_n(full, 2); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); t.render_n(full, spp); e1.record(); torch.cuda.synchronize()
print("sum of slabs %.3f ms; full frame %.3f ms" % (tot, e0.elapsed_time(e1)))
