"""What the material table buys a small scene that does NOT have the reference's table sizes (round 5: render_*_table_kernel, the table's
shape as data): tests/test_gpu_dispatch.py's "three spheres" and "one sphere two planes" scenes, 1920x1080 x 64 spp, with the table
(default) and with RPT_NO_MATERIAL_TABLE=1 (a process each: the knobs are read once).   python tools/table_shape_time.py"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CODE = r"""
import sys, hashlib
sys.path.insert(0, %r)
import conftest, torch
rpt = conftest.load_package()
import test_gpu_dispatch as T
for which in ("three spheres", "one sphere two planes", "three spheres on a floor", "two spheres two planes", "sdf two lights",
              "five spheres on a floor", "six spheres two planes", "six primitives partial patches", "eight spheres four planes"):
    s, _ = T._table_scene(rpt, which)
    t = rpt.Tracer(s, device=0, seed=1)
    buf = rpt.DeviceColorBuffer(1920, 1080)
    t.render_n(buf, 64); torch.cuda.synchronize()
    ms = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); t.render_n(buf, 64); e1.record(); e1.synchronize(); ms.append(e0.elapsed_time(e1))
    print("%%-24s %%8.3f ms  %%8.1f Msamples/s  image %%s" %% (which, min(ms), 1920 * 1080 * 64 / min(ms) / 1e3, hashlib.sha1(buf.pixels.cpu().numpy().tobytes()).hexdigest()[:10]), flush=True)
""" % os.path.join(HERE, "..", "tests")
for no_table in ("0", "1"):
    print("RPT_NO_MATERIAL_TABLE=%s" % no_table, flush=True)
    r = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, RPT_NO_MATERIAL_TABLE=no_table), text=True, capture_output=True)
    print("\n".join(l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l), flush=True)
