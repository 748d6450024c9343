"""Stock scene, `spp` samples per launch: Msamples/s over many launches and a hash of the image (bit equality across kernels:
run once with RPT_COMPACT_MAX_SPP=0 and once with a large value).  usage: compact_time.py w h spp [launches]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
rpt = conftest.load_package()
w, h, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else max(8, 256 // spp)
t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
buf = rpt.DeviceColorBuffer(w, h)
for _ in range(3):
    t.render_n(buf, spp)
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        t.render_n(buf, spp)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / n)
digest = hashlib.sha1(buf.pixels.cpu().numpy().tobytes()).hexdigest()[:12]
print("%dx%d x %d spp/launch: %.4f ms -> %.0f Msamples/s  image %s  [RPT_COMPACT_MAX_SPP=%s]" % (
    w, h, spp, best * 1e3, w * h * spp / best / 1e6, digest, os.environ.get("RPT_COMPACT_MAX_SPP", "")))
