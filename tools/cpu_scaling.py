import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest, oracle_lib, numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f): print(f, open(f).read().strip())
o = oracle_lib.Oracle("liboracle.so")
d = o.scene_analytical()
w, h = 1920, 1080
px = np.zeros((h, w, 4), np.float32)
o.render(d, w, h, 2, pixels=px, threads=o.max_threads())
for thr in (8, 16, 32, 64, 128, 256):
    if thr > 2 * o.max_threads(): break
    t = time.perf_counter(); o.render(d, w, h, 8, seed=1, frames_done=2, pixels=px, threads=thr); dt = time.perf_counter() - t
    print("threads %3d: %.2f Msamples/s" % (thr, w * h * 8 / dt / 1e6))
