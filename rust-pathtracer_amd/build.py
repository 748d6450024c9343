"""Build the HIP extension (librpt_hip.so) for gfx950, in-tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librpt_hip.so")
SOURCES = ["kernels.hip", "kernels_fast.hip", "capi.hip"]
# kernels_fast.hip: the same kernels with relaxed arithmetic (RPT_RENDER_FAST_MATH); every other file is strict
EXTRA_FLAGS = {"kernels_fast.hip": ["-fno-hip-fp32-correctly-rounded-divide-sqrt", "-ffp-contract=fast"]}
HEADERS = ["dev_math.h", "dev_prof.h", "dev_bsdf.h", "dev_scene.h", "dev_scene_large.h", "dev_integrator.h", "launch.h", "host_scene.h",
           os.path.join("..", "..", "include", "rpt.h"), os.path.join("..", "..", "include", "rpt_strict_math.h")]
# -ffp-contract=off: results are compared bit for bit with a CPU restatement, the only
# fused operations are the explicit fma calls of rpt_strict_math.h.
# -mllvm -disable-machine-licm: MachineLICM hoists the materialisation of ~70 literal constants (the
#   f64 polynomial coefficients of rpt_strict_math.h, two VGPRs each) out of the sample loop, where
#   they stay live for the whole kernel: 183 VGPRs (2 waves/SIMD) with it, 115 without.
# -fno-slp-vectorize: SLP packs scalar f32 ops into v_pk_mul/add_f32, which issue at half rate on
#   gfx950 and need paired registers: 115 -> 95 VGPRs and +6 % throughput without it.
# -mllvm -amdgpu-sched-strategy=max-ilp: the machine scheduler interleaves independent chains (the three divides of a
#   normalize, the three pow of the background) instead of minimising register pressure first: +2 % on configs[1]
#   and [3] at the same 96 VGPRs (iterative-ilp / iterative-minreg: no gain).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
         "-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fPIC"]
# The two -mllvm options are tuning only (they change instruction order / register use, never a result); a toolchain that
# does not know them still builds the library without them.
TUNING_FLAGS = ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), lib=LIB, objdir_name="build"):
    """Compile csrc/*.hip -> librpt_hip.so.  hipcc cross-compiles gfx950 without a GPU.
    `extra_flags` / `lib` / `objdir_name`: experiment builds next to the product library (tools/)."""
    if not force and not needs_build():
        return lib
    objdir = os.path.join(HERE, objdir_name)
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:                                   # one object per source (each with its own flags), in parallel
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc()] + FLAGS + EXTRA_FLAGS.get(src, []) + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            plain = [c for c in cmd if c not in TUNING_FLAGS]
            print("build.py: retrying without the -mllvm tuning options: " + " ".join(plain))
            subprocess.run(plain, check=True, cwd=CSRC)
    link = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", lib]
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True, cwd=CSRC)
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
