"""One rank's share of BASELINE configs[2] (3840x2160, 8 ranks) against the height of the cyclic row blocks: the per-sample cost of
less coherent wave tiles (a wave's 8 x 8 pixels span 8 / tile_rows blocks that lie `world` blocks apart) against load balance."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import tiling
w, h, world = 3840, 2160, 8
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128          # python tools/tile_rows_time.py [spp] [tile_rows ...]
blocks = [int(a) for a in sys.argv[2:]] or [2, 4, 8, 16, 32]
t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
for tile_rows in blocks:
    worst = 0.0
    line = []
    for rank in range(world):
        rows = tiling.tile_row_count(h, tile_rows, rank, world)
        tile = torch.zeros(max(rows, 1), w, 4, dtype=torch.float32, device="cuda")
        t.render_tile(tile, w, h, 0, 16, tile_rows, rank, world); torch.cuda.synchronize()      # (also teaches the dispatch order)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t.render_tile(tile, w, h, 0, spp, tile_rows, rank, world); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        worst = max(worst, ms)
        line.append("%.2f" % ms)
    print("tile_rows %3d, %d spp: slowest of 8 ranks %.2f ms -> whole frame at most %.0f Msamples/s (%.0f per GPU)   %s  [%s]" % (
          tile_rows, spp, worst, w * h * spp / worst / 1e3, w * h * spp / worst / 1e3 / world, "; ".join(line),
          " ".join("%s=%s" % (k, os.path.basename(v)) for k, v in sorted(os.environ.items()) if k.startswith("RPT_"))), flush=True)
