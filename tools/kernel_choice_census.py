"""Which kernel instantiation the parity suite's random small scenes take (tests/test_gpu_parity.py, _random_small_scene): how many of the
fuzzed scenes exercise the material table by class of accepted set (round 6), the 2^n tables, the per-hit kernel.  On the GPU box:
    python tools/kernel_choice_census.py [first seed] [count]"""
import collections, ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest
import numpy as np
import test_gpu_parity as T
rpt = conftest.load_package()
first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 400)
tally, classes = collections.Counter(), collections.Counter()
for seed in range(first, first + count):
    rng = np.random.default_rng(1000 + seed)
    s = T._random_small_scene(rpt, rng)
    t = rpt.Tracer(s, device=0, seed=1)
    buf = rpt.DeviceColorBuffer(32, 32)
    t.render_n(buf, 8)                                      # (more than one sample: the megakernel, not the compacting kernel)
    c = C.c_uint32()
    rpt.lib().rpt_debug_kernel_choice(t._h, C.byref(c))
    n = len(s.spheres) + len(s.planes)
    kind = "by class" if c.value & 8 else ("2^n rows" if c.value & 6 else "per hit")
    tally[(kind, "5-12 primitives" if n >= 5 else "<= 4 primitives")] += 1
    if c.value & 8:
        classes[(c.value >> 8) & 0xFF] += 1
    t.close()
for k in sorted(tally):
    print("%-10s %-16s %4d scenes" % (k[0], k[1], tally[k]))
print("classes of the by-class scenes:", dict(sorted(classes.items())))
