"""Resident waves over the time of one launch, from the start stamp and the duration every wave leaves behind (RPT_DISPATCH_TIMELINE=1;
kernels.hip, cost_record): how full the chip is at the start, in the middle and in the tail of a launch, per dispatch order.
    RPT_DISPATCH_TIMELINE=1 [RPT_DISPATCH_ORDER=2|1] python tools/dispatch_timeline.py [c2|c4|c5|share] [spp]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
import numpy as np
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes, tiling
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = {"c2": (rpt.AnalyticalScene, 1920, 1080, 256), "c4": (scenes.sdf_scene, 1920, 1080, 64),
       "c5": (lambda: scenes.random_spheres_scene(10000, 16), 2048, 2048, 32), "share": (rpt.AnalyticalScene, 3840, 2160, 512)}[which]
w, h, spp = cfg[1:4]
if len(sys.argv) > 2: spp = int(sys.argv[2])
t = rpt.Tracer(cfg[0](), device=0, seed=1)
lib = rpt.lib()
if which == "share":
    rows = tiling.tile_row_count(h, 2, 3, 8)
    tile = torch.zeros(rows, w, 4, dtype=torch.float32, device="cuda")
    render = lambda: t.render_tile(tile, w, h, 0, spp, 2, 3, 8)
    tx, ty = (w + 15) // 16, (rows + 15) // 16
else:
    buf = rpt.DeviceColorBuffer(w, h)
    render = lambda: t.render_n(buf, spp)
    tx, ty = (w + 15) // 16, (h + 15) // 16
n = tx * ty
raw = np.zeros(n * 10, dtype=np.uint32)
for it in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); render(); e1.record(); torch.cuda.synchronize()
    nt = C.c_uint32(0)
    rc = lib.rpt_debug_sched_read(t._h, raw.ctypes.data_as(C.POINTER(C.c_uint32)), n, C.byref(nt))
    assert rc == 0 and nt.value == n, (rc, nt.value, n)
    dur = raw[:4 * n].astype(np.int64)
    start = raw[6 * n:10 * n].astype(np.int64)
    live = dur > 0
    s0 = start[live].min()
    st = ((start[live] - s0) & 0xFFFFFFFF).astype(np.float64)
    en = st + dur[live]
    T = en.max()
    ticks_per_ms = T / e0.elapsed_time(e1)
    nb = 50
    edges = np.linspace(0, T, nb + 1)
    occ = np.zeros(nb)
    for i in range(nb):                      # wave-time inside each bin / bin length = mean resident waves
        lo, hi = edges[i], edges[i + 1]
        occ[i] = np.clip(np.minimum(en, hi) - np.maximum(st, lo), 0, None).sum() / (hi - lo)
    print("launch %d: %.2f ms (%d waves, clock %.0f ticks/ms); mean resident waves %.0f of 5120; per 2 %% of the launch:" % (
        it, e0.elapsed_time(e1), live.sum(), ticks_per_ms, (en - st).sum() / T))
    print("   " + " ".join("%4d" % v for v in occ))
    # when is the last wave of each decile of dispatch started
    print("   start time of the last wave: %.1f %% of the launch; waves started in the last 10 %%: %d" % (100 * st.max() / T, (st > 0.9 * T).sum()), flush=True)
