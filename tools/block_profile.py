"""Per-block lane utilisation and time shares of the megakernel (csrc/dev_prof.h).

  python tools/block_profile.py build      # here: cross-compile librpt_hip_prof.so (the product's translation units with -DRPT_PROFILE_BLOCKS)
  python tools/block_profile.py [spp] [c2|c4|c5]   # on the GPU box: render that config with it and print the table
"""
import ctypes as C
import importlib.util
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PKG = os.path.join(ROOT, "rust-pathtracer_amd")
PROF_LIB = os.path.join(PKG, "librpt_hip_prof.so")
BLOCKS = ["TRACE", "  closest_hit", "  background", "  finalize", "  finish+camera", "SHADE", "  make_frame", "  nee_sample", "  any_hit",
          "  disney_eval", "  disney_sample", "    lobe diffuse", "    lobe clearcoat", "    lobe spec", "  tail", "PASS", "    grid begin", "    grid cell",
          "    grid cell (shadow)", "      list trip > 1", "      candidate root", "    walk head", "    lights", "ALIVE"]

if len(sys.argv) > 1 and sys.argv[1] == "build":
    spec = importlib.util.spec_from_file_location("_rpt_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    print(b.build(force=True, extra_flags=["-DRPT_PROFILE_BLOCKS"], lib=PROF_LIB, objdir_name="build_prof"))
    sys.exit(0)

os.environ["RPT_LIB"] = PROF_LIB
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402
import torch  # noqa: E402

rpt = conftest.load_package()
from rust_pathtracer_amd import scenes  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
which = sys.argv[2] if len(sys.argv) > 2 else "c2"
w, h = (2048, 2048) if which == "c5" else (1920, 1080)
scene = {"c2": rpt.AnalyticalScene, "c4": scenes.sdf_scene, "c5": lambda: scenes.random_spheres_scene(10000, 16)}[which]()
t = rpt.Tracer(scene, device=0, seed=1)
buf = rpt.DeviceColorBuffer(w, h)
lib = rpt.lib()
lib.rpt_prof_read.restype = C.c_int
out = (C.c_ulonglong * (len(BLOCKS) * 3))()
t.render_n(buf, 2)
torch.cuda.synchronize()
lib.rpt_prof_read(out)                        # discard the warm-up
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); t.render_n(buf, spp); e1.record(); torch.cuda.synchronize()
assert lib.rpt_prof_read(out) == 0
n_samples = w * h * spp
print("profiled build, %s: %dx%d x %d spp in %.2f ms (%.0f Msamples/s with the counters on)" % (which, w, h, spp, e0.elapsed_time(e1), n_samples / e0.elapsed_time(e1) / 1e3))
pass_cycles = out[BLOCKS.index("PASS") * 3 + 2]
print("%-20s %12s %9s %8s %10s %12s" % ("block", "wave execs", "lanes/64", "share", "execs/smp", "lane-exec/smp"))
for i, name in enumerate(BLOCKS):
    ex, ln, cy = out[i * 3], out[i * 3 + 1], out[i * 3 + 2]
    if ex == 0:
        continue
    denom = pass_cycles
    if name == "ALIVE":                       # lanes that still have samples to render, averaged over the waves' time
        print("lanes with samples left, cycle-weighted: %.1f %% (the rest: pixels whose samples are done, waiting for the wave's last)" % (100.0 * ln / cy))
        continue
    print("%-20s %12d %8.1f%% %7.1f%% %10.3f %12.3f" % (name, ex, 100.0 * ln / (64.0 * ex), 100.0 * cy / denom, 64.0 * ex / n_samples, ln / n_samples))
