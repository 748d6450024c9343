"""Does splitting a frame over two streams of ONE GPU fill the launch tails?  The resident frame of a context that drives the same
device twice (virtual ranks, RPT_GATHER=p2p: each rank has its own stream and half of the rows) against one rank, progressive
steps back to back without host synchronisation in between."""
import os, sys, time
os.environ["RPT_GATHER"] = "p2p"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
from rust_pathtracer_amd import tiling
w, h, spp, steps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 256, 10)
which = sys.argv[5] if len(sys.argv) > 5 else "c2"
from rust_pathtracer_amd import scenes
make = {"c2": rpt.AnalyticalScene, "c4": scenes.sdf_scene, "c5": lambda: scenes.random_spheres_scene(10000, 16)}[which]
for n, tile_rows in ((1, 16), (2, 16), (2, 48), (3, 16), (4, 16))[:int(sys.argv[6]) if len(sys.argv) > 6 else 5]:
    t = rpt.Tracer(make(), devices=[0] * n, seed=1)
    r = tiling.TiledRender(t, w, h, tile_rows=tile_rows)
    r.render_n(spp); t.resident_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.render_n(spp)
    t.resident_sync()
    dt = (time.perf_counter() - t0) / steps
    print("%s %d stream(s), %2d-row blocks: %dx%d x %d spp per step: %.3f ms -> %.1f Msamples/s" % (which, n, tile_rows, w, h, spp, dt * 1e3, w * h * spp / dt / 1e6), flush=True)
    t.close()
