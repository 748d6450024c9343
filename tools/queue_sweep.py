"""Msamples/s of the stock frame for a list of samples-per-launch values (argv: width height spp...), one line per spp.
Run once per RPT_CHUNKS_PER_BLOCK / RPT_SHADE_THRESHOLD setting (they are read once per process)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
w, h = int(sys.argv[1]), int(sys.argv[2])
t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
buf = rpt.DeviceColorBuffer(w, h)
out = []
for spp in [int(a) for a in sys.argv[3:]]:
    n = max(3, min(200, 512 // spp))
    for _ in range(3):
        t.render_n(buf, spp)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            t.render_n(buf, spp)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    out.append("%d:%.0f" % (spp, w * h * spp / best / 1e6))
print("%dx%d chunks/block=%s  " % (w, h, os.environ.get("RPT_CHUNKS_PER_BLOCK", "auto")) + "  ".join(out))
