"""Build experiment variants of librpt_hip.so side by side (rust-pathtracer_amd/variants/<name>.so), for A/B timing in ONE
gpurun call (tools/run_variants.sh selects each through RPT_LIB).  Each entry: name -> extra hipcc flags."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "rust-pathtracer_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)

VARIANTS = {
    "base": [],
    "w4": ["-DRPT_SMALL_WAVES_PER_SIMD=4"],
    "w6": ["-DRPT_SMALL_WAVES_PER_SIMD=6"],
    "w8": ["-DRPT_SMALL_WAVES_PER_SIMD=8"],
    # wavefront form of large scenes: occupancy of the two kernels
    "walk6": ["-DRPT_WF_WALK_WAVES_PER_SIMD=6"],
    "walk7": ["-DRPT_WF_WALK_WAVES_PER_SIMD=7"],
    "walk8": ["-DRPT_WF_WALK_WAVES_PER_SIMD=8"],
    "shade3": ["-DRPT_WF_SHADE_WAVES_PER_SIMD=3"],
    "shade5": ["-DRPT_WF_SHADE_WAVES_PER_SIMD=5"],
    "shade6": ["-DRPT_WF_SHADE_WAVES_PER_SIMD=6"],
    # the compacting kernel of one-sample launches
    "compact4": ["-DRPT_COMPACT_WAVES_PER_SIMD=4"],
    "compact5": ["-DRPT_COMPACT_WAVES_PER_SIMD=5"],
    "compact6": ["-DRPT_COMPACT_WAVES_PER_SIMD=6"],
    "compact7": ["-DRPT_COMPACT_WAVES_PER_SIMD=7"],
    "gridbatch2": ["-DRPT_GRID_BATCH=2"],
    "gridbatch4": ["-DRPT_GRID_BATCH=4"],
    "gridbatch4w5": ["-DRPT_GRID_BATCH=4", "-DRPT_LARGE_WAVES_PER_SIMD=5"],
    "gridbatch2any2": ["-DRPT_GRID_BATCH=2", "-DRPT_GRID_BATCH_ANY=2"],
    "gridbatch2any4": ["-DRPT_GRID_BATCH=2", "-DRPT_GRID_BATCH_ANY=4"],
    "gridany2": ["-DRPT_GRID_BATCH_ANY=2"],
    "gridpark": ["-DRPT_GRID_PARK"],
    "parkany": ["-DRPT_GRID_PARK_ANY"],
    "batch3": ["-DRPT_GRID_BATCH=3"],
    "any3": ["-DRPT_GRID_BATCH_ANY=3"],
    "gridpark_w5": ["-DRPT_GRID_PARK", "-DRPT_LARGE_WAVES_PER_SIMD=5"],
    "base_w5": ["-DRPT_LARGE_WAVES_PER_SIMD=5"],
    "base_w7": ["-DRPT_LARGE_WAVES_PER_SIMD=7"],
    "gridbatch1": ["-DRPT_GRID_BATCH=1", "-DRPT_GRID_BATCH_ANY=1"],
    "compact8": ["-DRPT_COMPACT_WAVES_PER_SIMD=8"],
    # BASELINE.json's "scene/material/light tables staged in LDS": the headline kernel reading its tables from LDS instead of SGPRs
    "plain_divides": ["-DRPT_PLAIN_DIVIDES"],
    "plain_sqrt": ["-DRPT_PLAIN_SQRT"],
    "scalar_plain": ["-DRPT_SCALAR_DIVIDES_PLAIN"],
    "large_w5": ["-DRPT_LARGE_WAVES_PER_SIMD=5"],
    "large_w7": ["-DRPT_LARGE_WAVES_PER_SIMD=7"],
    "large_w6": ["-DRPT_LARGE_WAVES_PER_SIMD=6"],
    "slp": ["-fslp-vectorize"],
    "scene_plain": ["-DRPT_SCENE_ARGUMENT_PLAIN"],
    "no_max_ilp": ["-mllvm", "-amdgpu-sched-strategy=max-occupancy"],
    "large_w8": ["-DRPT_LARGE_WAVES_PER_SIMD=8"],
    "large_w5_sp": ["-DRPT_LARGE_WAVES_PER_SIMD=5", "-DRPT_SCALAR_DIVIDES_PLAIN"],
    "sdf_w4": ["-DRPT_SDF_WAVES_PER_SIMD=4"],
    "sdf_w6": ["-DRPT_SDF_WAVES_PER_SIMD=6"],
    "pair_w4": ["-DRPT_LARGE_PAIR_WAVES_PER_SIMD=4"],
    "pair_w6": ["-DRPT_LARGE_PAIR_WAVES_PER_SIMD=6"],
    # round 4
    "denoise_tile32": ["-DRPT_DENOISE_TILE=32"],              # the denoiser's LDS tiles 32 x 32 (1 024 threads) instead of 16 x 16: halos 1.32 instead of 1.69 loads per pixel
    # round 4: code-generation options that cannot change a result (scheduling, register allocation, branch shape)
    "cg_early_ifcvt": ["-mllvm", "-amdgpu-early-ifcvt"],
    "cg_wave_prio": ["-mllvm", "-amdgpu-set-wave-priority"],
    "cg_bias0": ["-mllvm", "-amdgpu-schedule-metric-bias=0"],
    "cg_bias50": ["-mllvm", "-amdgpu-schedule-metric-bias=50"],
    "cg_bias100": ["-mllvm", "-amdgpu-schedule-metric-bias=100"],
    "cg_no_highrp": ["-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule"],
    "cg_no_lowocc": ["-mllvm", "-amdgpu-disable-clustered-low-occupancy-reschedule"],
    "cg_trackers": ["-mllvm", "-amdgpu-use-amdgpu-trackers"],
    "cg_no_liverange": ["-mllvm", "-amdgpu-opt-vgpr-liverange=0"],
    "cg_no_loop_align": ["-mllvm", "-amdgpu-disable-loop-alignment"],
    "cg_preload16": ["-mllvm", "-amdgpu-kernarg-preload-count=16"],
    "cg_no_postsched": ["-mllvm", "-enable-post-misched=0"],
    "cg_prealloc_sgpr": ["-mllvm", "-amdgpu-prealloc-sgpr-spill-vgprs"],
    "cg_dce_in_ra0": ["-mllvm", "-amdgpu-dce-in-ra=0"],
    "cg_O2": ["-O2"],
    "cg_no_unroll": ["-fno-unroll-loops"],
    "cg_inline_all": ["-mllvm", "-amdgpu-early-inline-all"],
    "cg_skip_uniform": ["-mllvm", "-structurizecfg-skip-uniform-regions"],
    "cg_relaxed_uniform": ["-mllvm", "-structurizecfg-relaxed-uniform-regions"],
    "cg_no_prera_opt": ["-mllvm", "-amdgpu-enable-pre-ra-optimizations=0"],
    "cg_clause4": ["-mllvm", "-amdgpu-max-memory-clause=4"],
    "cg_no_cluster": ["-mllvm", "-misched-cluster=0"],
    "cg_exec_pre_ra0": ["-mllvm", "-amdgpu-opt-exec-mask-pre-ra=0"],
    # round 5
    "perop": ["-DRPT_GUARD_PER_OP"],                           # every kernel with the range tests next to the operation (k_small too): the differential partner of the range trackers (tools/range_soak.py)
    "dn_unfused": ["-DRPT_DENOISE_UNFUSED"],                   # the denoiser one pass per iteration (round 4's form) against the fused first three
    "cg_licm_on": [],        # (built with RPT_TUNING_FLAGS="-mllvm -amdgpu-sched-strategy=max-ilp": machine LICM back on)
}

if __name__ == "__main__":
    names = sys.argv[1:] or list(VARIANTS)
    for n in names:
        if "=" in n:                                             # ad hoc: name=-DFLAG,-DOTHER=1
            n, _, fl = n.partition("=")
            VARIANTS[n] = [f for f in fl.split(",") if f]
        flags = VARIANTS[n] if n in VARIANTS else []
        lib = os.path.join(ROOT, "rust-pathtracer_amd", "variants", n + ".so")
        try:
            b.build(force=True, extra_flags=flags, lib=lib, objdir_name="build_" + n)
            print("built", lib)
        except Exception as e:      # noqa: BLE001 - an option this compiler does not know: skip the variant
            print("FAILED", n, str(e).splitlines()[0])
