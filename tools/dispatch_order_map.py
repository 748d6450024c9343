"""The tile costs a launch leaves behind and the dispatch order made of them (kernels.hip, block_tile): quantiles, a coarse map of
the frame, and how the cost of a tile changes between two launches.   python tools/dispatch_order_map.py [c2|c4|c5] [spp]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
import numpy as np
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = {"c2": (rpt.AnalyticalScene, 1920, 1080, 256), "c4": (scenes.sdf_scene, 1920, 1080, 64),
       "c5": (lambda: scenes.random_spheres_scene(10000, 16), 2048, 2048, 32)}[which]
w, h, spp = cfg[1:4]
if len(sys.argv) > 2: spp = int(sys.argv[2])
t = rpt.Tracer(cfg[0](), device=0, seed=1)
buf = rpt.DeviceColorBuffer(w, h)
lib = rpt.lib()
tx, ty = (w + 15) // 16, (h + 15) // 16
n = tx * ty
raw = np.zeros(n * 10, dtype=np.uint32)
prev = None
for it in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t.render_n(buf, spp); e1.record(); torch.cuda.synchronize()
    nt = C.c_uint32(0)
    rc = lib.rpt_debug_sched_read(t._h, raw.ctypes.data_as(C.POINTER(C.c_uint32)), n, C.byref(nt))
    assert rc == 0 and nt.value == n, (rc, nt.value, n)
    cost = raw[:4 * n].reshape(n, 4).astype(np.float64).sum(axis=1) / 1e5          # ms of wave time per tile
    order = raw[4 * n:5 * n]
    assert sorted(order.tolist()) == list(range(n)), "the order is not a permutation"
    q = np.percentile(cost, [0, 10, 25, 50, 75, 90, 99, 100])
    print("launch %d: %.2f ms; tile cost (ms of wave time, 4 waves): %s  sum/5120 slots = %.2f ms" % (
        it, e0.elapsed_time(e1), " ".join("%.2f" % v for v in q), cost.sum() / 5120.0), flush=True)
    if prev is not None:
        print("   correlation with the previous launch's costs %.4f; mean |change| %.1f %%" % (np.corrcoef(prev, cost)[0, 1], 100 * np.mean(np.abs(cost - prev)) / cost.mean()))
    prev = cost.copy()
    if it in (0, 3):
        m = cost.reshape(ty, tx)
        step_y, step_x = max(1, ty // 17), max(1, tx // 30)
        print("   cost map (rows of tiles, 0 = image bottom... as stored: tile row 0 = top), ms:")
        for y in range(0, ty, step_y):
            print("   " + " ".join("%5.1f" % m[y:y + step_y, x:x + step_x].mean() for x in range(0, tx, step_x)))
        pos = np.empty(n, dtype=np.int64); pos[order] = np.arange(n)
        pm = pos.reshape(ty, tx) / float(n)
        print("   dispatch position of the NEXT launch (0 = first, 1 = last):")
        for y in range(0, ty, step_y):
            print("   " + " ".join("%5.2f" % pm[y:y + step_y, x:x + step_x].mean() for x in range(0, tx, step_x)))
