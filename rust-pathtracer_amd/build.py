"""Build the HIP extension for gfx950, in-tree: librpt_hip.so (the product: exactly include/rpt.h) and librpt_hip_test.so (the SAME
objects linked with the test hooks of include/rpt_test.h and the probe kernels: what the GPU parity tests load).

    python rust-pathtracer_amd/build.py
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
TEST_LIB = os.path.join(HERE, "librpt_hip_test.so")

# One translation unit per kernel class (csrc/kernel_common.h says what each build of them is):
#   strict    k_small tracks the range tests of the short divide / sqrt; k_compact, k_sdf, k_large test next to every operation
#   relaxed   the four render TUs once more with hipcc's fast divide / sqrt and FMA contraction: what RPT_RENDER_FAST_MATH selects
PEROP = ["-DRPT_GUARD_PER_OP"]
RELAXED = ["-DRPT_RELAXED_BUILD", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-ffp-contract=fast"]
# (object, source, extra flags, which library: "both" | "product" | "test")
OBJECTS = [
    ("k_small", "k_small.hip", [], "both"),
    ("k_compact", "k_compact.hip", PEROP, "both"),
    ("k_sdf", "k_sdf.hip", PEROP, "both"),
    ("k_large", "k_large.hip", PEROP, "both"),
    ("k_small_fast", "k_small.hip", RELAXED, "both"),
    ("k_compact_fast", "k_compact.hip", RELAXED, "both"),
    ("k_sdf_fast", "k_sdf.hip", RELAXED, "both"),
    ("k_large_fast", "k_large.hip", RELAXED, "both"),
    ("k_util", "k_util.hip", [], "both"),
    # the denoiser's taps are independent multiply / add sequences: packed f32 instructions halve their issue slots there
    # (the path kernels lose from SLP: it pins register pairs)
    ("denoise", "denoise.hip", ["-fslp-vectorize"], "both"),
    ("capi", "capi.hip", [], "product"),
    ("capi_test", "capi.hip", ["-DRPT_TEST_HOOKS"], "test"),
    ("k_probes", "k_probes.hip", [], "test"),
]
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
# -fvisibility=hidden: the library exports what include/rpt.h declares (capi.hip pushes default visibility around it) and nothing else.
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-fvisibility=hidden"]
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
    """Every file a change of which means a rebuild: all sources and headers under csrc/ and include/."""
    inc = os.path.join(HERE, "..", "include")
    return (glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(inc, "*.h")) +
            [os.path.abspath(__file__)])


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False, extra_flags=(), lib=LIB, objdir_name="build", test_lib=None, only=None, jobs=None):
    """Compile csrc/*.hip -> `lib` (and, when `test_lib` is given, the test build beside it).  hipcc cross-compiles gfx950 without a GPU.
    `extra_flags` / `lib` / `objdir_name`: experiment builds next to the product library (tools/); `only`: recompile just these
    objects (the others are taken from `objdir_name`/ as they are — or, if missing there, from the product's build/)."""
    if not force and not needs_build(lib) and (test_lib is None or not needs_build(test_lib)):
        return lib
    objdir = os.path.join(HERE, objdir_name)
    os.makedirs(objdir, exist_ok=True)
    flags = BASE_FLAGS + _tuning_flags()
    jobs = jobs or max(1, min(8, os.cpu_count() or 1))
    pending, objs = [], {}
    for name, src, extra, where in OBJECTS:
        if where == "test" and test_lib is None:
            continue
        obj = os.path.join(objdir, name + ".o")
        objs[name] = (obj, where)
        if only is not None and name not in only:
            if not os.path.exists(obj):
                shutil.copy(os.path.join(HERE, "build", name + ".o"), obj)
            continue
        pending.append([_hipcc()] + flags + extra + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj])
    running, failed = [], []
    while pending or running:
        while pending and len(running) < jobs:
            cmd = pending.pop(0)
            if verbose:
                print(" ".join(cmd))
            running.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        cmd, p = running.pop(0)
        if p.wait() != 0:
            failed.append(" ".join(cmd))
    if failed:
        raise RuntimeError("build.py: compilation failed:\n" + "\n".join(failed))
    for out, kinds in ((lib, ("both", "product")), (test_lib, ("both", "test"))):
        if out is None:
            continue
        os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
        link = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + [o for o, w in objs.values() if w in kinds] + ["-ldl", "-o", out]
        if verbose:
            print(" ".join(link))
        subprocess.run(link, check=True, cwd=CSRC)
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True, test_lib=TEST_LIB))
