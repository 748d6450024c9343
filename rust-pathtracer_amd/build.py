"""Build the HIP extension (librpt_hip.so) for gfx950, in-tree.

    python rust-pathtracer_amd/build.py            # the shipped library
    python rust-pathtracer_amd/build.py --ab       # librpt_hip_ab.so: + every kernel form kept for A/B runs (-DRPT_AB_KERNELS); RPT_LIB selects it
"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librpt_hip.so")
# (object, source): kernels.hip is built three times (its first lines say why); everything but kernels_relaxed is strict
OBJECTS = [("kernels", "kernels.hip"), ("kernels_perop", "kernels.hip"), ("kernels_relaxed", "kernels.hip"), ("denoise", "denoise.hip"),
           ("capi", "capi.hip")]
AB_SKIP = {"kernels_perop"}                               # (an A/B build holds every kernel in its one strict object)
EXTRA_FLAGS = {"kernels_relaxed": ["-DRPT_RELAXED_BUILD", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-ffp-contract=fast"],
               # large and SDF scenes' kernels: the range tests of the short divide / sqrt next to every operation (kernels.hip, top)
               "kernels_perop": ["-DRPT_PEROP_BUILD"],
               # the denoiser's taps are independent multiply / add sequences: packed f32 instructions halve their issue slots there
               # (the path kernels lose from SLP: it pins register pairs)
               "denoise": ["-fslp-vectorize"]}
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
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC"]
# The two -mllvm options are tuning only (they change instruction order / register use, never a result); a toolchain that
# does not know them still builds the library without them (probed once, on an empty translation unit).
TUNING_FLAGS = ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]

_tuning_ok = None


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _tuning_flags():
    """TUNING_FLAGS if this hipcc accepts them, else []."""
    global _tuning_ok
    if _tuning_ok is None:
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "empty.hip")
            open(src, "w").write("#include <hip/hip_runtime.h>\n__global__ void k() {}\n")
            r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3"] + TUNING_FLAGS + ["-c", src, "-o", os.path.join(d, "empty.o")],
                               capture_output=True)
            _tuning_ok = r.returncode == 0
            if not _tuning_ok:
                print("build.py: this hipcc rejects the -mllvm tuning options; building without them")
    if os.environ.get("RPT_TUNING_FLAGS") is not None:               # experiments (tools/build_variants.py): replace the tuning options
        return os.environ["RPT_TUNING_FLAGS"].split()
    return TUNING_FLAGS if _tuning_ok else []


def _deps():
    """Every file a change of which means a rebuild: all sources and headers under csrc/ (csrc/ab/ too) and include/."""
    inc = os.path.join(HERE, "..", "include")
    return (glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "ab", "*.h")) +
            glob.glob(os.path.join(inc, "*.h")) + [os.path.abspath(__file__)])


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False, extra_flags=(), lib=LIB, objdir_name="build", ab=False):
    """Compile csrc/*.hip -> librpt_hip.so.  hipcc cross-compiles gfx950 without a GPU.
    `extra_flags` / `lib` / `objdir_name`: experiment builds next to the product library (tools/); `ab`: include the A/B kernels."""
    if not force and not needs_build(lib):
        return lib
    objdir = os.path.join(HERE, objdir_name)
    os.makedirs(objdir, exist_ok=True)
    flags = BASE_FLAGS + _tuning_flags() + (["-DRPT_AB_KERNELS"] if ab else [])
    procs, objs = [], []
    for name, src in OBJECTS:                             # one object each, with its own flags, in parallel
        if (ab or "-DRPT_GUARD_PER_OP" in extra_flags) and name in AB_SKIP:
            continue
        obj = os.path.join(objdir, name + ".o")
        cmd = [_hipcc()] + flags + EXTRA_FLAGS.get(name, []) + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        objs.append(obj)
    failed = [" ".join(cmd) for cmd, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("build.py: compilation failed:\n" + "\n".join(failed))
    os.makedirs(os.path.dirname(os.path.abspath(lib)), exist_ok=True)
    link = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-ldl", "-o", lib]
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True, cwd=CSRC)
    return lib


AB_LIB = os.path.join(HERE, "librpt_hip_ab.so")           # every kernel form ever measured (-DRPT_AB_KERNELS), next to the shipped library

if __name__ == "__main__":
    if "--ab" in sys.argv[1:]:                            # RPT_LIB=rust-pathtracer_amd/librpt_hip_ab.so python -m pytest tests -m gpu
        print(build(force=True, verbose=True, ab=True, lib=AB_LIB, objdir_name="build_ab"))
    else:
        print(build(force=True, verbose=True))
