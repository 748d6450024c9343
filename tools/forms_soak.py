"""Differential soak of the kernel forms against each other on frames larger than the oracle can check in test time: whole
frames must be bit-identical.  Large scenes: megakernel vs wavefront (sphere counts, depths, roulette, resumed accumulation,
ragged sizes); small scenes: megakernel vs compacting kernel at 1-3 samples per launch, progressive."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, torch
import numpy as np
rpt = conftest.load_package()
from rust_pathtracer_amd import scenes
A = rpt._abi
bad = 0


def same(a, b):
    return torch.equal(a.pixels.view(torch.int32), b.pixels.view(torch.int32))


rng = np.random.default_rng(11)
for k in range(10):
    n = int(rng.choice([64, 200, 1000, 4000, 10000]))
    s = scenes.random_spheres_scene(n_spheres=n, n_lights=int(rng.integers(0, 17)), seed=int(rng.integers(1, 2**31)))
    s.max_depth = int(rng.integers(1, 9))
    s.any_hit_uses_max_dist = bool(rng.random() < 0.7)
    w, h = int(rng.integers(300, 1500)), int(rng.integers(200, 1100))
    steps = [int(x) for x in rng.integers(1, 6, size=int(rng.integers(1, 4)))]
    rr = A.RPT_RENDER_RUSSIAN_ROULETTE if rng.random() < 0.4 else 0
    t = rpt.Tracer(s, device=0, seed=k)
    bufs = []
    for form in (A.RPT_RENDER_LARGE_MEGAKERNEL, A.RPT_RENDER_LARGE_WAVEFRONT):
        t.flags = form | rr
        b = rpt.DeviceColorBuffer(w, h)
        for spp in steps:
            t.render_n(b, spp)
        bufs.append(b)
    torch.cuda.synchronize()
    ok = same(*bufs)
    bad += not ok
    print("large  %5d spheres %2d lights depth %d %4dx%-4d steps %-12s rr %d: %s" % (n, len(s.lights), s.max_depth, w, h, steps, bool(rr), "same" if ok else "DIFFERENT"), flush=True)
    t.close()

for k, (w, h) in enumerate([(1920, 1080), (800, 600), (1001, 777), (3840, 2160)]):
    t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=100 + k)
    steps = [1, 2, 1, 3, 1, 1, 2]
    bufs = []
    for form in (0, A.RPT_RENDER_SMALL_COMPACT):
        t.flags = form
        b = rpt.DeviceColorBuffer(w, h)
        for spp in steps:
            # form 0 takes the compacting kernel for 1-2 samples by default: force the megakernel with a 3+ sample split
            if form == 0:
                t.flags = A.RPT_RENDER_NESTED_LOOPS        # the nested-loop kernel: a third, independent schedule
            t.render_n(b, spp)
        bufs.append(b)
    torch.cuda.synchronize()
    ok = same(*bufs)
    bad += not ok
    print("small  %4dx%-4d steps %s nested-loops vs compacting: %s" % (w, h, steps, "same" if ok else "DIFFERENT"), flush=True)
    t.close()
print("soak:", "all same" if bad == 0 else "%d DIFFERENT" % bad)
sys.exit(1 if bad else 0)
