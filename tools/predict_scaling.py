"""Predict weak-scaling balance on ONE GPU: for N = 1, 2, 4, 8 render every rank's tile of bench.py's
N-GPU frame (cyclic 2-row blocks, 256 spp) one after the other and compare kernel times.  The slowest
rank bounds the step; the all-gather (not measured here) adds ~1 ms at N = 8."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest, torch
import bench
from rust_pathtracer_amd import tiling
rpt = conftest.load_package()
t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
base = None
for n in (1, 2, 4, 8):
    w, h = bench.frame_size(n)
    times = []
    for r in range(n):
        rows = tiling.tile_row_count(h, 2, r, n)
        tile = torch.zeros(rows, w, 4, dtype=torch.float32, device="cuda")
        t.render_tile(tile, w, h, 0, 8, 2, r, n); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t.render_tile(tile, w, h, 8, 256, 2, r, n); e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    if base is None:
        base = times[0]
    print("N=%d frame %dx%d: per-rank kernel ms min %.2f max %.2f -> predicted efficiency (excl. gather) %.3f" %
          (n, w, h, min(times), max(times), base * (w * h / n) / (1920 * 1080) / max(times)))
