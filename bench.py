#!/usr/bin/env python3
"""bench.py — Msamples/s of the render hot path on MI355X (driver contract: see the task).

One "step" = one pass of the hot path over one batch: SPP consecutive Tracer::render() calls folded into one
launch sequence over the whole frame, on a frame that is already resident in HBM.

N = 1   BASELINE.json configs[1]: AnalyticalScene 1920x1080, 256 spp per step, f32.
N > 1   BASELINE.json configs[2] EXACTLY: AnalyticalScene 3840x2160, 1024 spp per step, the image row-tiled over the N
        GPUs (cyclic 8-row blocks), one process per GPU; every step ends with the RCCL gather of the tiles to rank 0
        and the scatter into the full image on rank 0's device, inside the timed region.  STRONG scaling: the frame
        is the same for every N; rank 0 also renders it alone after the timed region, so the line carries the
        measured 1-GPU time of the same frame next to the N-GPU time.  Fixed work per GPU (weak scaling) is a
        secondary key.
The multi-GPU machinery is the library's (include/rpt.h: rpt_create_rank, rpt_resident_render,
rpt_resident_gather_device); torch.distributed (gloo) only carries the 128-byte communicator id, the barriers and the
max over ranks of the elapsed time.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]
       (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector
GPU_CLOCK_HZ = 2.4e9           # the same table's engine clock: what FP32_PEAK_TFLOPS is quoted at (the kernels run at 2.1-2.4 GHz: traffic.json, shader_cycles_per_launch)
C2 = (1920, 1080, 256)         # BASELINE.json configs[1]
C3 = (3840, 2160, 1024)        # BASELINE.json configs[2]
# Rows per block of the cyclic row tiling for N > 1.  One rank's share of configs[2] on one GPU (tools/tile_rows_time.py, 8 virtual ranks,
# 1 024 spp): blocks of 1 / 2 / 4 / 8 / 16 rows -> slowest rank 84.7 / 84.7 / 83.6 / 82.5 / 82.0 ms, the eight ranks within 1.9 / 1.8 /
# 1.2 / 2.3 / 7.3 % of each other: with 8 rows a wave's 8 x 8 pixels are one piece of the image again (coherent paths), and the
# ranks still balance.
TILE_ROWS = 8
SECONDARY_LIMIT_S = int(os.environ.get("RPT_BENCH_SECONDARY_LIMIT_S", "150"))        # N > 1: the legs after the headline (weak scaling, the one-GPU frame, configs[4]) may take this long together
PROFILES = os.path.join("profiles", "r6")                    # committed rocprofv3 summaries of this command (tools/collect_profiles.sh)
# What a correctly rounded f32 divide / square root costs the VALU in this library (csrc/dev_math.h): a quotient is v_rcp + 2 fma
# shared by the numerators of one denominator, then mul + 2 fma + v_div_fixup each; a root is v_rsq + 2 mul + 2 fma; plus the range
# test.  Small scenes' megakernel tracks the operands (v_frexp_exp + v_max3 per divide, v_frexp_exp + v_min per root; the two DS
# operations each are not VALU work): 9 for a lone quotient, 6 per quotient of a normalize — 8 is used — and 7 for a root.  The
# other kernels test next to the operation (v_max3, v_frexp_exp, two compares; subtract + compare for a root): 11 / 7.3 — 9 is used —
# and 7.  (Rounds 1-3 priced both at hipcc's expansions, 12 and 15, which the library no longer executes.)
DIV_INSTRUCTIONS, SQRT_INSTRUCTIONS = 9, 7
# VALU wave-instructions per SIMD per quad-cycle of a kernel of nothing but independent v_fma_f32 chains at 5 waves per SIMD, at the
# MEASURED clock (GRBM_GUI_ACTIVE of the same launch): tools/microbench/valu_issue_peak.hip, profiles/r5/valu_issue_peak.txt
# (1 / 2 / 4 / 5 / 8 waves per SIMD: 0.68 / 1.32 / 1.50 / 1.58 / 1.68; v_pk_fma_f32: 0.93 at 8).  Round 4 quoted 1.48 for this — its C
# loop had been SLP-packed into half as many v_pk_fma_f32 and priced at an assumed 2.4 GHz; the two errors nearly cancelled.
VALU_ISSUE_FMA_STREAM = 1.58
DIV_INSTRUCTIONS_TRACKED, SQRT_INSTRUCTIONS_TRACKED = 8, 7


def source_hash():
    """sha256 over the sources a library is built from (csrc/, include/, build.py): what a traffic.json is stamped with beside the
    library's own hash, so that counters survive a rebuild of identical sources and nothing else."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "rust-pathtracer_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.join(pkg, "build.py")])
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read() + b"\0")
    return h.hexdigest()


def counters_for(name, lib_path):
    """The committed PMC figures of profiles/<round>/<name>/traffic.json — IF they were collected from the library that is loaded
    (its sha256, or failing that the sha256 of the sources it is built from).  Returns (dict or None, why)."""
    import hashlib
    path = os.path.join(ROOT, PROFILES, name, "traffic.json")
    rel = os.path.join(PROFILES, name, "traffic.json")
    if not os.path.exists(path):
        return None, "%s does not exist" % rel
    t = json.load(open(path))
    lib_hash = hashlib.sha256(open(lib_path, "rb").read()).hexdigest() if os.path.exists(lib_path) else None
    if t.get("library_sha256") and t["library_sha256"] == lib_hash:
        return t, "%s: rocprofv3 PMC passes of the loaded library (sha256 %s...), committed (%s)" % (rel, lib_hash[:12], t["correction"])
    if t.get("source_sha256") and t["source_sha256"] == source_hash():
        return t, "%s: rocprofv3 PMC passes of a library built from these very sources (source sha256 %s...), committed (%s)" % (rel, t["source_sha256"][:12], t["correction"])
    return None, "%s was collected from another build (library sha256 %s..., loaded %s...): the counters are not this code's" % (
        rel, str(t.get("library_sha256"))[:12], str(lib_hash)[:12])


def weak_frame(n_gpus):
    """Fixed pixels per GPU: configs[1]'s frame scaled by sqrt(N) per axis (N = 4 is configs[2]'s frame)."""
    s = math.sqrt(n_gpus)
    return int(round(1920 * s / 8.0)) * 8, int(round(1080 * s / 8.0)) * 8


def _cpu_quota():
    """CPUs this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return n


def op_counts(desc=None, frame=(240, 136, 4), skip_missed_sphere_tests=False):
    """Algorithmic flops per sample of a scene's view (default: the AnalyticalScene), measured by the op-counting build of
    the oracle on a small frame of the same view.  `skip_missed_sphere_tests`: leave out the operations of ray/sphere tests
    that missed — what a scene with an acceleration structure is priced against (the oracle's loop tests every sphere:
    0.4 Mflop per sample on 10 000 spheres, of which 99.6 % are misses a grid never executes)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oc = oracle_lib.Oracle("liboracle_opcount.so")
    cw, ch, cs = frame
    c, missed = oc.opcount_split(desc if desc is not None else oc.scene_analytical(), cw, ch, cs, seed=1)
    n = cw * ch * cs
    out = {}
    if skip_missed_sphere_tests:
        out["brute_force_flops_per_sample"] = round((c["add"] + c["mul"] + c["div"] + c["sqrt"]) / n, 1)
        c = {k: c[k] - missed[k] for k in c}
    out.update({"flops_per_sample": round((c["add"] + c["mul"] + c["div"] + c["sqrt"]) / n, 1),
                "transcendentals_per_sample": round(c["transc"] / n, 2),
                "divides_per_sample": round(c["div"] / n, 2), "sqrts_per_sample": round(c["sqrt"] / n, 2)})
    return out


def roofline_block(ops, launch_samples, kernel_s, kernel, launches, pixels, tracked=False):
    """FP32-VALU roofline of one config (no dense contraction, HBM is not the limiter): algorithmic flops per
    step = flops per sample (oracle op counts) x samples per step, over the measured duration of the step's launches."""
    tfl = ops["flops_per_sample"] * launch_samples / kernel_s / 1e12
    # the same work with every correctly rounded divide / sqrt counted at the VALU instructions this library issues for it
    # (one operation per instruction, an fma as one): what the VALU actually has to issue for the algorithmic flops
    div_i, sqrt_i = (DIV_INSTRUCTIONS_TRACKED, SQRT_INSTRUCTIONS_TRACKED) if tracked else (DIV_INSTRUCTIONS, SQRT_INSTRUCTIONS)
    expanded = ops["flops_per_sample"] + ops["divides_per_sample"] * (div_i - 1) + ops["sqrts_per_sample"] * (sqrt_i - 1)
    hbm = 32.0 * pixels / kernel_s / 1e9
    return {"bound": "fp32_valu", "achieved": round(tfl, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tfl / FP32_PEAK_TFLOPS, 5), "traffic": None,
            "kernel": kernel, "kernel_ms": round(kernel_s * 1e3, 3), "launches_per_step": launches,
            "algorithmic_flops_per_step": ops["flops_per_sample"] * launch_samples, **ops,
            "ieee_expanded_flops_per_sample": round(expanded, 1),
            "ieee_expanded_frac": round(expanded * launch_samples / kernel_s / 1e12 / FP32_PEAK_TFLOPS, 5),
            "ieee_expanded_note": "flops with a divide counted as %d and a square root as %d operations: the instructions of csrc/dev_math.h's correctly rounded sequences" % (div_i, sqrt_i),
            "hbm_algorithmic_GBs": round(hbm, 3), "hbm_frac": round(hbm / HBM_PEAK_GBS, 6)}


def committed_traffic(name, lib_path):
    """HBM bytes per launch from the committed PMC passes of a profile directory, or None when they are another build's (counters_for)."""
    t, why = counters_for(name, lib_path)
    return (t["hbm_bytes_per_launch"] if t else None), why


def other_configs(rpt, torch, device, small):
    """BASELINE.json configs[3] and configs[4] on one GPU, after the headline's timed region: a few steps each, HIP events on
    the launch stream, with their own op-counted flops per sample -> `roofline_c4` / `roofline_c5` (secondary keys)."""
    from rust_pathtracer_amd import scenes

    def run(scene, w, h, spp, reps):
        tracer = rpt.Tracer(scene, device=device, seed=1)
        buf = rpt.DeviceColorBuffer(w, h, device="cuda:%d" % device)
        tracer.render_n(buf, min(spp, 16))                         # warm-up with the timed launch's own SHAPE: the dispatch order of a launch is
        torch.cuda.synchronize()                                    # learned from the previous one of its shape (capi.hip sched_for: the key holds no
                                                                    # sample count), so a few samples do — configs[4]'s frame is not rendered twice
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            tracer.render_n(buf, spp)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        tracer.close()
        del buf
        return sum(ms) / len(ms) / 1e3

    out = {}
    out.update(headline_variants(rpt, torch, device, small))
    # The headline workload once more with the device listed TWICE in its context (include/rpt.h, rpt_create_multi): two ranks on one
    # GPU, each with its own stream and every other block of 16 rows, the frame resident in the context, steps issued back to back —
    # one rank's launch fills the tail of the other's.  Not the headline `value`: a step is then two concurrent launches.
    import time
    from rust_pathtracer_amd import tiling
    div0 = 8 if small else 1
    w, h, spp, steps = C2[0] // div0, C2[1] // div0, max(1, C2[2] // (16 if small else 1)), 6
    tracer = rpt.Tracer(rpt.AnalyticalScene(), devices=[device, device], seed=1)
    frame = tiling.TiledRender(tracer, w, h, tile_rows=16)
    frame.render_n(spp)
    tracer.resident_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        frame.render_n(spp)
    tracer.resident_sync()
    dt = (time.perf_counter() - t0) / steps
    # ... and as the reference uses it: one render() per redraw (1 spp per step), 200 redraws back to back
    frame.render_n(1)
    tracer.resident_sync()
    t0 = time.perf_counter()
    for _ in range(200):
        frame.render_n(1)
    tracer.resident_sync()
    dt1 = (time.perf_counter() - t0) / 200
    tracer.close()
    # ... and configs[3] (SDF scene) the same way
    w4, h4, spp4 = 1920 // div0, 1080 // div0, max(1, 64 // (16 if small else 1))
    tracer = rpt.Tracer(scenes.sdf_scene(), devices=[device, device], seed=1)
    frame = tiling.TiledRender(tracer, w4, h4, tile_rows=16)
    frame.render_n(spp4)
    tracer.resident_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        frame.render_n(spp4)
    tracer.resident_sync()
    dt4 = (time.perf_counter() - t0) / steps
    tracer.close()
    out["resident_two_streams_c4"] = {"value": round(w4 * h4 * spp4 / dt4 / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt4 * 1e3, 3), "steps": steps,
                                      "workload": "SDF sphere-march scene %dx%d x %d spp per step, the GPU listed twice" % (w4, h4, spp4)}
    out["resident_two_streams_1spp"] = {"value": round(w * h / dt1 / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt1 * 1e3, 4), "steps": 200,
                                        "workload": "the same context, one sample per pixel per step (the reference's render() per redraw)"}
    out["resident_two_streams"] = {"value": round(w * h * spp / dt / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
                                   "workload": "AnalyticalScene %dx%d x %d spp per step, frame resident in a context that lists the GPU twice "
                                               "(two streams, cyclic 16-row blocks), host clock over %d steps" % (w, h, spp, steps)}
    div = 8 if small else 1
    sdf = scenes.sdf_scene()
    w, h, spp = 1920 // div, 1080 // div, 64 // (4 if small else 1)
    t = run(sdf, w, h, spp, 3)
    blk = roofline_block(op_counts(sdf.describe(), (240, 136, 2)), w * h * spp, t, "render_sdf_march2_sized_table_kernel<3u>", 1, w * h)
    blk.update({"workload": "SDF sphere-march scene %dx%d x %d spp per step (BASELINE.json configs[3])" % (w, h, spp),
                "value": round(w * h * spp / t / 1e6, 2), "value_unit": "Msamples/s"})
    if not small:
        blk["traffic"], blk["traffic_source"] = committed_traffic("c4", rpt._lib.LIB_PATH)
    out["roofline_c4"] = blk
    big = scenes.random_spheres_scene(10000, 16)
    ops = op_counts(big.describe(), (128, 128, 1), skip_missed_sphere_tests=True)
    w = h = 4096 // div
    full = 512 // (16 if small else 1)
    t_full = run(big, w, h, full, 1)
    blk = roofline_block(ops, w * h * full, t_full, "render_large_regen_kernel", 1, w * h)
    blk.update({"workload": "10k spheres + 16 lights %dx%d x %d spp in one call (BASELINE.json configs[4], the whole frame on one GPU)" % (w, h, full),
                "value": round(w * h * full / t_full / 1e6, 2), "value_unit": "Msamples/s",
                "note": "flops per sample exclude ray/sphere tests that missed (brute_force_flops_per_sample is the oracle's loop); "
                        "the grid walk's own arithmetic is not algorithmic work and is not counted"})
    t8 = run(big, w, h, 8, 2)
    blk["progressive_8spp"] = {"kernel": "render_large_regen_kernel",
                               "kernel_ms": round(t8 * 1e3, 3), "value": round(w * h * 8 / t8 / 1e6, 2),
                               "frac": round(ops["flops_per_sample"] * w * h * 8 / t8 / 1e12 / FP32_PEAK_TFLOPS, 5)}
    if not small:
        # (the committed counters: of this very launch, and of the same kernel on the 2048 x 2048 x 32-spp frame: 2 x 16 B per pixel + the spills)
        blk["traffic"], blk["traffic_source"] = committed_traffic("c5_full", rpt._lib.LIB_PATH)
        blk["progressive_8spp"]["traffic_2048x2048x32"] = committed_traffic("c5", rpt._lib.LIB_PATH)[0]
    out["roofline_c5"] = blk
    # the denoiser (include/rpt.h, project-defined): an HBM pass, 32 B per pixel per iteration
    for name, (dw, dh) in (("roofline_denoise_1080p", (1920 // div, 1080 // div)), ("roofline_denoise_4k", (3840 // div, 2160 // div))):
        buf = rpt.DeviceColorBuffer(dw, dh, device="cuda:%d" % device)
        buf.pixels.uniform_(0.0, 2.0)
        iters = 3
        res = buf.denoise(iters, 2.0)
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            buf.denoise(iters, 2.0, out=res)
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        del res
        t = min(ms) / 1e3
        # The fused kernel reads the frame once and writes it once whatever the iteration count (<= 3): its ALGORITHMIC traffic is 32 B
        # per pixel, not the 3 x 32 B of three separate passes (which rounds 4-5 priced it against: a 0.50 that was never HBM's).
        algo = 32.0 * dw * dh
        gbs = algo / t / 1e9
        c, tr_src = counters_for("denoise_1080p" if name.endswith("1080p") else "denoise_4k", rpt._lib.LIB_PATH) if not small else (None, None)
        tr = c["hbm_bytes_per_launch"] if c else None
        out[name] = {"bound": "latency (barriers and LDS reads between the fused stages)", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                     "traffic": tr, **({"traffic_source": tr_src} if tr_src else {}),
                     **({"counter_GBs": round(tr / t / 1e9, 1), "counter_frac": round(tr / t / 1e9 / HBM_PEAK_GBS, 4)} if tr else {}),
                     **({"wait_inst_lds_share_of_wave_time": c["wait_inst_lds_share"]} if c and "wait_inst_lds_share" in c else {}),
                     "kernel": "denoise_fused_kernel<3> (the three iterations in one pass through LDS: one read and one write of the frame)",
                     "kernel_ms": round(t * 1e3, 4),
                     "algorithmic_bytes_per_step": algo,
                     "three_pass_equivalent_GBs": round(iters * algo / t / 1e9, 1),
                     "note": "NOT HBM-bound: `achieved` is the fused kernel's algorithmic 32 B per pixel over its time, `counter_GBs` what the PMC "
                             "passes saw it move; its waves spend 24 / 30 / 46 %% of their time issuing / stalled / in s_waitcnt behind the barriers and "
                             "LDS reads between the stages, at 99 %% lane utilisation and 7 waves per SIMD; waiting for the LDS unit to ISSUE is 3 %% "
                             "(SQ_WAIT_INST_LDS: wait_inst_lds_share_of_wave_time) (profiles/%s/denoise/summary.txt); three_pass_equivalent_GBs is what "
                             "rounds 4-5 printed as `achieved`" % PROFILES.split(os.sep)[-1],
                     "workload": "a-trous denoiser, %d iterations on a %dx%d RGBA f32 buffer" % (iters, dw, dh)}
        del buf
    try:
        out["c3_rank_tiles"] = rank_tiles_leg(rpt, torch, device, small)
    except Exception as e:      # noqa: BLE001 - a secondary leg must not cost the line
        out["c3_rank_tiles"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # the secondary legs' values as plain numbers under ONE key (a driver record that keeps top-level scalars and small dicts keeps these)
    out["secondary"] = {"unit": "Msamples/s",
                        "c3_sdf": out["roofline_c4"]["value"], "c3_sdf_frac": out["roofline_c4"]["frac"],
                        "c4_large": out["roofline_c5"]["value"], "c4_large_frac": out["roofline_c5"]["frac"],
                        "general": out.get("general_kernels", {}).get("value"), "relaxed": out.get("relaxed", {}).get("value"),
                        "six_primitives": out.get("six_primitives", {}).get("value"),
                        "six_primitives_material_per_hit": out.get("six_primitives", {}).get("value_material_per_hit"),
                        "relaxed_rmse": out.get("relaxed", {}).get("rmse_vs_strict"),
                        "c2_projected_8gpu": out["c3_rank_tiles"].get("projected_value"),
                        "c2_projected_scaling": out["c3_rank_tiles"].get("projected_scaling_vs_whole_frame"),
                        "c2_tiles_max_over_mean": out["c3_rank_tiles"].get("max_over_mean")}
    return out


def rank_tiles_leg(rpt, torch, device, small, world=8):
    """BASELINE.json configs[2] on the ONE GPU there is: each of the 8 ranks' tiles (cyclic TILE_ROWS-row blocks, exactly the launch
    rank r of `--gpus 8` makes: same rows, same global pixel keys) rendered ALONE and timed with HIP events on the launch stream, and
    the whole frame on the same GPU beside them.  The scaling figures derived from them are a PROJECTION: no gather, no second GPU —
    what they do measure is the load balance of the tiling (max / mean), the thing SURVEY.md section 7 says limits scaling."""
    from rust_pathtracer_amd import tiling
    w, h, spp = (C3[0] // 8, C3[1] // 8, max(1, C3[2] // 16)) if small else C3
    tracer = rpt.Tracer(rpt.AnalyticalScene(), device=device, seed=1)
    dev = "cuda:%d" % device

    def timed(launch, reps):
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        return min(ms)

    tile_ms = []
    for r in range(world):
        rows = tiling.tile_row_count(h, TILE_ROWS, r, world)
        tile = torch.zeros(max(rows, 1), w, 4, dtype=torch.float32, device=dev)
        launch = lambda n=spp: tracer.render_tile(tile, w, h, 0, n, TILE_ROWS, r, world)       # noqa: E731
        launch(max(1, spp // 8))                      # this rank's dispatch order (the tables are keyed by rank: capi.hip sched_for)
        torch.cuda.synchronize()
        tile_ms.append(timed(launch, 2))
        del tile
    buf = rpt.DeviceColorBuffer(w, h, device=dev)
    tracer.render_n(buf, max(1, spp // 8))
    torch.cuda.synchronize()
    t1 = timed(lambda: tracer.render_n(buf, spp), 2)
    tracer.close()
    del buf
    mx, mean = max(tile_ms), sum(tile_ms) / len(tile_ms)
    return {"workload": "AnalyticalScene %dx%d x %d spp (BASELINE.json configs[2]): each of the %d ranks' tiles (cyclic %d-row blocks) rendered alone "
                        "on this one GPU, best of 2 launches each, HIP events" % (w, h, spp, world, TILE_ROWS),
            "tile_ms": [round(t, 3) for t in tile_ms], "max_ms": round(mx, 3), "mean_ms": round(mean, 3), "max_over_mean": round(mx / mean, 4),
            "whole_frame_one_gpu_ms": round(t1, 3),
            "slowest_rank_Msamples_per_s": round(w * (h // world) * spp / mx / 1e3, 1),
            "projected_scaling": round(sum(tile_ms) / mx, 3),
            "projected_scaling_vs_whole_frame": round(t1 / mx, 3),
            "projected_value": round(w * h * spp / mx / 1e3, 1), "unit": "Msamples/s",
            "note": "projected, unmeasured on hardware: sum of the tile times / the slowest tile (and the one-GPU whole-frame time / the slowest "
                    "tile); excludes the RCCL gather (16.6 MB per rank per step, enqueued beside the next step's render) and assumes 8 GPUs "
                    "as fast as this one"}


def timed_steps(torch, tracer, buf, spp, steps):
    """Mean device time of `steps` launches of `spp` samples (HIP events on the launch stream), in seconds."""
    tracer.render_n(buf, spp)
    torch.cuda.synchronize()
    ms = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tracer.render_n(buf, spp)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return sum(ms) / len(ms) / 1e3


def general_kernels_leg(small):
    """`bench.py --general-kernels-leg` (a process of its own: the library reads its knobs once): configs[1] with
    RPT_NO_SIZED_KERNELS=1 RPT_NO_MATERIAL_TABLE=1, i.e. through the kernel any small scene OTHER than the reference's takes."""
    import torch
    import __graft_entry__ as entry
    rpt = entry._load_package()
    w, h, spp = (C2[0] // 8, C2[1] // 8, max(1, C2[2] // 16)) if small else C2
    tracer = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
    buf = rpt.DeviceColorBuffer(w, h, device="cuda:0")
    t = timed_steps(torch, tracer, buf, spp, 5)
    tracer.close()
    from rust_pathtracer_amd import scenes
    tracer = rpt.Tracer(scenes.six_primitive_scene(), device=0, seed=1)
    t6 = timed_steps(torch, tracer, buf, spp, 5)
    tracer.close()
    print("GENERAL " + json.dumps({"value": round(w * h * spp / t / 1e6, 2), "unit": "Msamples/s", "kernel_ms": round(t * 1e3, 3),
                                   "kernel": "render_small_regen_kernel",
                                   "six_primitives_value": round(w * h * spp / t6 / 1e6, 2),
                                   "workload": "AnalyticalScene %dx%d x %d spp per step with RPT_NO_SIZED_KERNELS=1 RPT_NO_MATERIAL_TABLE=1" % (w, h, spp)}))


def headline_variants(rpt, torch, device, small):
    """What the headline does NOT say (N = 1, after the timed region):
    general_kernels  the headline kernel is instantiated for exactly the reference scene's table sizes (2 spheres / 1 plane / 1
                     light) and reads a hit's material from a 32-row table that exists for <= 3 primitives; every other small
                     scene takes render_small_regen_kernel.  The same frame through THAT kernel.
    relaxed          SURVEY.md 8c tier T1: configs[1] under RPT_RENDER_FAST_MATH (hipcc's fast divide / sqrt, FMA contraction) —
                     Msamples/s, and against the strict frame after the same 256 spp: whole-frame RMSE and the number of pixels with
                     |delta| > 1e-4 in some channel (an ulp flips a branch now and then: a sample changes by O(1), a pixel by O(1/spp)).
                     What bit-exact IEEE divides and roots cost, and what they buy."""
    import subprocess
    out = {}
    w, h, spp = (C2[0] // 8, C2[1] // 8, max(1, C2[2] // 16)) if small else C2
    try:
        # the child renders on ITS device 0: the entry of the parent's visible-device list at index `device` (the parent's list may
        # not start at 0), and a profiler wrapped around the parent stays with the parent
        visible = [v for v in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if v.strip() != ""]
        env = dict(os.environ, RPT_NO_SIZED_KERNELS="1", RPT_NO_MATERIAL_TABLE="1",
                   HIP_VISIBLE_DEVICES=visible[device] if device < len(visible) else str(device))
        for k in list(env):
            if k in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES") or k.startswith(("ROCPROF", "ROCPROFILER_", "ROCTRACER_")):
                del env[k]
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--general-kernels-leg"] + (["--small"] if small else []),
                           capture_output=True, text=True, timeout=300, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("GENERAL ")]
        out["general_kernels"] = json.loads(line[-1][8:]) if line else {"error": (r.stderr or r.stdout)[-400:]}
        if line:
            out["general_kernels"]["note"] = ("what any small scene other than the reference's gets: table sizes as data, the material built "
                                              "per hit; the headline value is the sized + material-table instantiation of the same kernel")
    except Exception as e:      # noqa: BLE001 - a secondary leg must not cost the line
        out["general_kernels"] = {"error": "%s: %s" % (type(e).__name__, e)}
    try:
        # A small scene of MORE than four primitives (scenes.six_primitive_scene: five spheres on the checker floor): until round 6 every
        # such scene took the kernel that builds the material per hit; now its accepted sets are sorted into classes of equal material on
        # the host and the megakernel reads 64 LDS rows by class (csrc/launch.h, MatClassMap).
        from rust_pathtracer_amd import scenes
        six = rpt.Tracer(scenes.six_primitive_scene(), device=device, seed=1)
        t6 = timed_steps(torch, six, rpt.DeviceColorBuffer(w, h, device="cuda:%d" % device), spp, 5)
        six.close()
        out["six_primitives"] = {"value": round(w * h * spp / t6 / 1e6, 2), "unit": "Msamples/s", "kernel_ms": round(t6 * 1e3, 3),
                                 "kernel": "render_small_regen_maptable_kernel",
                                 "value_material_per_hit": out.get("general_kernels", {}).get("six_primitives_value"),
                                 "workload": "five spheres with whole materials on the checker floor, %dx%d x %d spp per step: the material table by class "
                                             "of accepted set (12 classes); value_material_per_hit: the same scene with RPT_NO_MATERIAL_TABLE=1" % (w, h, spp)}
    except Exception as e:      # noqa: BLE001
        out["six_primitives"] = {"error": "%s: %s" % (type(e).__name__, e)}
    try:
        A = rpt._abi
        strict = rpt.Tracer(rpt.AnalyticalScene(), device=device, seed=1)
        fast = rpt.Tracer(rpt.AnalyticalScene(), device=device, seed=1)
        fast.flags = A.RPT_RENDER_FAST_MATH
        fs = rpt.DeviceColorBuffer(w, h, device="cuda:%d" % device)
        ff = rpt.DeviceColorBuffer(w, h, device="cuda:%d" % device)
        strict.render_n(fs, spp)
        fast.render_n(ff, spp)
        torch.cuda.synchronize()
        d = (ff.pixels[..., :3].double() - fs.pixels[..., :3].double())
        finite = torch.isfinite(d).all(dim=-1)
        dd = torch.where(torch.isfinite(d), d, torch.zeros_like(d))
        rmse = float(torch.sqrt((dd * dd).mean()).item())
        outliers = int(((dd.abs() > 1e-4).any(dim=-1) | ~finite).sum().item())
        same = int((ff.pixels.view(torch.int32) == fs.pixels.view(torch.int32)).all(dim=-1).sum().item())
        tbuf = rpt.DeviceColorBuffer(w, h, device="cuda:%d" % device)
        t_fast = timed_steps(torch, fast, tbuf, spp, 5)
        t_strict = timed_steps(torch, strict, tbuf, spp, 5)
        out["relaxed"] = {"value": round(w * h * spp / t_fast / 1e6, 2), "unit": "Msamples/s", "kernel_ms": round(t_fast * 1e3, 3),
                          "strict_value_same_run": round(w * h * spp / t_strict / 1e6, 2),
                          "kernel": "render_small_regen_sized_table_kernel_fast", "flags": "RPT_RENDER_FAST_MATH",
                          "rmse_vs_strict": rmse, "pixels_over_1e-4": outliers, "pixels_bit_identical": same, "pixels": w * h,
                          "max_abs_delta": float(dd.abs().max().item()),
                          "workload": "AnalyticalScene %dx%d, both frames after %d spp from an empty buffer, seed 1" % (w, h, spp),
                          "note": "NOT the headline arithmetic: v_rcp / v_rsq based divide and sqrt (~2.5 ulp) and FMA contraction, in the same "
                                  "instantiation as the headline (sized tables, material table: since round 5); strict_value_same_run / value "
                                  "is the price of the correctly rounded operations"}
        strict.close(); fast.close()
    except Exception as e:      # noqa: BLE001
        out["relaxed"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def cpu_baseline(width, height, budget_s=12.0):
    """Time the CPU oracle (a port of the reference's rayon path: OpenMP over scanlines) on
    the host cores, on a bounded sample of the same workload: the full 1920x1080 frame at a
    few spp (Msamples/s does not depend on spp on the CPU).  The build timed is the one with the
    PLATFORM libm (liboracle_libm.so: glibc sinf/cosf/powf/log2f, what the reference's Rust f32 methods
    call on Linux) — 1.6x faster than the bit-reproducible strict-math build the parity tests use, so it
    is the fairer stand-in for the reference binary.  The thread count is the best of
    {quota, 2 x quota} CPUs (cgroup-aware: oversubscribing a quota makes the baseline slower,
    which would flatter the GPU)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    o = oracle_lib.Oracle("liboracle_libm.so")
    desc = o.scene_analytical()
    quota = _cpu_quota()
    px = np.zeros((height, width, 4), dtype=np.float32)
    o.render(desc, width, height, 1, seed=1, pixels=px, threads=quota)        # warms the thread pool and the pages
    best_rate, threads = 0.0, quota
    for cand in sorted({quota, min(2 * quota, max(quota, o.max_threads()))}):
        t0 = time.perf_counter()
        o.render(desc, width, height, 8, seed=1, frames_done=1, pixels=px, threads=cand)
        rate = width * height * 8 / (time.perf_counter() - t0)
        if rate > best_rate:
            best_rate, threads = rate, cand
    spp = max(1, min(1024, int(budget_s * best_rate / (width * height))))
    t0 = time.perf_counter()
    o.render(desc, width, height, spp, seed=1, frames_done=9, pixels=px, threads=threads)
    t1 = time.perf_counter()
    msps = width * height * spp / (t1 - t0) / 1e6
    return {
        "value": round(msps, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
        "per_thread": round(msps / threads, 3), "threads_of_visible": "%d/%d" % (threads, os.cpu_count() or 0),
        "sample": "%dx%d x %d spp, same scene/seed (%.1f s of CPU work); OpenMP scanline loop, g++ -O3 -march=x86-64-v3, glibc libm; "
                  "%d threads on a %d-CPU quota (%d logical CPUs visible)"
                  % (width, height, spp, t1 - t0, threads, quota, os.cpu_count() or 0),
    }


def f64_reference(rpt, torch, device, threads):
    """BASELINE.json's "radiance within a stated float tolerance of the reference's CPU path ... per-pixel L2 error < 1e-4 after 256
    spp", MEASURED against something that is not the same f32 arithmetic: the oracle's statements instantiated over double
    (oracle/rpt_oracle.hpp RPT_ORACLE_F64: same draws, same operation order, glibc's double libm, f64 running mean).  The GPU's strict
    f32 frame of configs[1] after 256 spp against that frame on the same rows; beside it the two f32 CPU oracles (strict libm — bit-identical
    to the GPU — and glibc libm: the freedom the real Rust binary's platform libm has).  An f32 rounding now and then flips a branch
    (r1 < cdf, d2 > radius2): that sample changes by O(1), its pixel by O(1/spp) — which is what `pixels_over_1e-4` counts; the RMSE
    is the figure to hold against 1e-4.  Bounded: every third row of the frame (a fixed set: ~15 s of CPU on 16 threads)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    w, h, spp = C2
    o64 = oracle_lib.Oracle("liboracle_f64.so")
    olm = oracle_lib.Oracle("liboracle_libm.so")
    ost = oracle_lib.Oracle("liboracle.so")
    desc = o64.scene_analytical()
    tracer = rpt.Tracer(rpt.AnalyticalScene(), device=device, seed=1)
    buf = rpt.DeviceColorBuffer(w, h, device="cuda:%d" % device)
    tracer.render_n(buf, spp)
    torch.cuda.synchronize()
    gpu = buf.pixels.cpu().numpy()
    tracer.close()
    # every third row of the frame (360 rows: a fixed, evenly spread third of it — the figures do not depend on how fast this host is;
    # rows are independent, and bands through the spheres' contact region weigh 3 x the frame's mean: a subset must not pick its rows)
    rows = np.arange(1, h, 3, dtype=np.uint32)
    t0 = time.perf_counter()
    frames = {}
    for name, o in (("f64", o64), ("glibc", olm), ("strict", ost)):
        px = np.zeros((h, w, 4), dtype=np.float32)
        o.render_rows(desc, w, h, spp, rows, seed=1, pixels=px, threads=threads)
        frames[name] = px[rows]
    cpu_s = time.perf_counter() - t0

    def against(x, y):
        d = x[..., :3].astype(np.float64) - y[..., :3].astype(np.float64)
        finite = np.isfinite(d).all(axis=-1)
        d = np.where(np.isfinite(d), d, 0.0)
        l2 = np.sqrt((d * d).sum(axis=-1))                      # per-pixel L2 over the colour channels
        return {"rmse": float(np.sqrt((d * d).mean())), "median_abs": float(np.median(np.abs(d))), "max_abs": float(np.abs(d).max()),
                "per_pixel_l2_mean": float(l2.mean()), "pixels_l2_over_1e-4": int((l2 > 1e-4).sum()), "pixels_nonfinite": int((~finite).sum())}

    g = gpu[rows]
    return {"workload": "AnalyticalScene %dx%d after %d spp from an empty buffer, seed 1 (configs[1]); every third row (%d of %d rows: "
                        "%.0f s of CPU on %d threads for the three CPU frames)" % (w, h, spp, len(rows), h, cpu_s, threads),
            "whole_frame_on_cpu": "strict f32 oracle (= the GPU frame, bit for bit) against f64, all 1080 rows, computed once in the dev container "
                                  "(profiles/r6/f64_whole_frame.txt): rmse 9.79e-05, 0.444 % of the pixels beyond 1e-4 in L2, max 0.0216",
            "pixels": int(len(rows) * w),
            "gpu_f32_vs_f64": against(g, frames["f64"]),
            "gpu_f32_bit_identical_to_strict_oracle": bool((g.view(np.uint32) == frames["strict"].view(np.uint32)).all()),
            "glibc_f32_vs_f64": against(frames["glibc"], frames["f64"]),
            "strict_f32_vs_glibc_f32": against(frames["strict"], frames["glibc"]),
            "note": "f64 = the oracle's statements over double (same draws and operation order): the frame every f32 frame is a rounding of. "
                    "rmse is the figure to hold against BASELINE.json's 1e-4; pixels_l2_over_1e-4 counts pixels in which an f32 rounding "
                    "flipped a branch of some sample (a sample then changes by O(1), its pixel by O(1/spp)); strict_f32_vs_glibc_f32 is "
                    "what the real binary's platform libm may differ by"}


def config1_line(rpt, torch, device, threads):
    """BASELINE.json configs[0] exactly ("c1"): AnalyticalScene 800 x 600, ONE sample per pixel per call — one
    reference render() — on the CPU port (glibc libm, median of 9 calls after a warm one) and on the GPU (a device-resident
    buffer, 400 calls back to back, host clock)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    w, h = 800, 600
    o = oracle_lib.Oracle("liboracle_libm.so")
    desc = o.scene_analytical()
    px = np.zeros((h, w, 4), dtype=np.float32)
    o.render(desc, w, h, 1, seed=1, pixels=px, threads=threads)
    times = []
    for k in range(9):
        t0 = time.perf_counter()
        o.render(desc, w, h, 1, seed=1, frames_done=1 + k, pixels=px, threads=threads)
        times.append(time.perf_counter() - t0)
    cpu_s = sorted(times)[len(times) // 2]
    tracer = rpt.Tracer(rpt.AnalyticalScene(), device=device, seed=1)
    buf = rpt.DeviceColorBuffer(w, h, device="cuda:%d" % device)
    for _ in range(8):
        tracer.render(buf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        tracer.render(buf)
    torch.cuda.synchronize()
    gpu_s = (time.perf_counter() - t0) / 400
    tracer.close()
    return {"workload": "AnalyticalScene 800x600 x 1 spp per call (BASELINE.json configs[0]: one reference render())",
            "cpu_ms_per_call": round(cpu_s * 1e3, 3), "cpu_value": round(w * h / cpu_s / 1e6, 2), "cpu_cores": threads, "cpu_kind": "port",
            "gpu_ms_per_call": round(gpu_s * 1e3, 4), "gpu_value": round(w * h / gpu_s / 1e6, 1), "unit": "Msamples/s",
            "gpu_kernel": "render_small_compact_dense_sized_table_kernel", "gpu_over_cpu": round(cpu_s / gpu_s, 1)}


class TorchGatherRender:
    """Insurance only (see main): the interface of tiling.TiledRender over a plain per-rank Tracer, with the tiles
    gathered to rank 0 by torch.distributed's RCCL backend and scattered by the library's untile kernel."""

    def __init__(self, tracer, tiling, width, height, tile_rows, rank, world, local_rank):
        import torch
        import torch.distributed as dist
        self.tracer, self.tiling, self.dist = tracer, tiling, dist
        self.width, self.height, self.tile_rows, self.rank, self.world = width, height, tile_rows, rank, world
        dev = torch.device("cuda", local_rank)
        if not hasattr(TorchGatherRender, "group"):
            TorchGatherRender.group = dist.new_group(backend="nccl")
        self.tile = torch.zeros(tiling.padded_rows(height, tile_rows, world), width, 4, dtype=torch.float32, device=dev)
        self.parts = [torch.empty_like(self.tile) for _ in range(world)] if rank == 0 else None
        self.frames, self.image, self._ms = 0, None, 0.0

    def render_n(self, spp):
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.tracer.render_tile(self.tile, self.width, self.height, self.frames, spp, self.tile_rows, self.rank, self.world)
        e1.record()
        self._ev = (e0, e1)
        self.frames += spp

    def kernel_ms(self):
        self._ev[1].synchronize()
        return self._ev[0].elapsed_time(self._ev[1])

    def gather_begin(self):
        import torch
        self.dist.gather(self.tile, self.parts, dst=0, group=TorchGatherRender.group)
        if self.rank == 0:
            self.image = self.tiling.untile(torch.stack(self.parts), self.width, self.height, self.tile_rows, self.world, tracer=self.tracer)

    def gather_end(self):
        import torch
        torch.cuda.synchronize()
        return self.image

    def gather(self):
        self.gather_begin()
        return self.gather_end()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (profiling runs)")
    ap.add_argument("--headline-only", action="store_true", help="skip the secondary configs[3] / configs[4] legs (profiling runs)")
    ap.add_argument("--smoke-shared-gpu", action="store_true",
                    help="TEST ONLY: every rank drives cuda:0 and the tiles are gathered with peer copies (RCCL rejects "
                         "duplicate devices), to exercise the N>1 control flow on a 1-GPU box; the numbers mean nothing")
    ap.add_argument("--force-multi", action="store_true", help="TEST ONLY: take the N > 1 path with however many ranks there are (one, without a launcher)")
    ap.add_argument("--small", action="store_true", help="TEST ONLY: 1/8-size frames and 1/16 of the samples (control-flow rehearsals)")
    ap.add_argument("--general-kernels-leg", action="store_true", help="INTERNAL: the `general_kernels` secondary leg (a process of its own: see headline_variants)")
    args = ap.parse_args()
    if args.general_kernels_leg:
        return general_kernels_leg(args.small)

    # (the host driver of this pool supports dmabuf IPC only: without this RCCL's peer mappings fail with `hipIpcGetMemHandle: invalid
    #  argument`; the launcher's environment normally carries it already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import __graft_entry__ as entry
    rpt = entry._load_package()
    from rust_pathtracer_amd import tiling

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # TEST ONLY: the N > 1 code path (library communicator, tiles, gather beside the render, secondary legs) with a world of ONE rank on
    # one GPU — what a test box can run of it for real (tests/test_gpu_multi.py); the numbers mean nothing
    multi = world > 1 or args.force_multi
    if args.force_multi and world == 1:
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("MASTER_PORT", "29531")
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    if args.smoke_shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (the data path's RCCL lives in the library).  Gloo announces its connections on stdout from
        # C++: send fd 1 to stderr while it does, so that stdout carries the one JSON line and nothing else.
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)

    def shrink(cfg):
        return (cfg[0] // 8, cfg[1] // 8, max(1, cfg[2] // 16)) if args.small else cfg

    def host_fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    scene = rpt.AnalyticalScene()
    extra = {}                          # secondary results of the JSON line (written under emit_lock once the watchdog runs)
    emitted = []
    emit_lock = threading.Lock()        # the line is printed ONCE, by whoever gets here first — the end of main or the watchdog —, and whole

    def emit():
        """Rank 0 prints the ONE JSON line.  Holds emit_lock from the look at `emitted` to the end of the print, so that the other
        caller waits for the line instead of leaving the process under it."""
        if rank != 0:
            return
        with emit_lock:
            if emitted:
                return
            emit_locked()
            sys.stdout.flush()
            emitted.append(True)

    def emit_locked():
        samples = width * height * spp * args.steps
        value = samples / elapsed / 1e6
        avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
        ops = op_counts()
        # FP32-VALU roofline of the megakernel: this rank's pixels x spp per step over the step's measured launch time
        launch_samples = local_pixels * spp
        algo_bytes = 32.0 * local_pixels          # 16 B read + 16 B write of the running mean per pixel per launch sequence
        hbm = algo_bytes / avg_kernel_s / 1e9
        launches = 1                              # one launch whatever spp is (kernel_common.h: a launch is tiles x chunks of samples)
        kernel = "render_small_regen_sized_table_kernel" if spp > 1 else "render_small_compact_sized_table_kernel"     # (capi.hip: launch_render, KernelChoice)
        roofline = roofline_block(ops, launch_samples, avg_kernel_s, kernel, launches, local_pixels, tracked=spp > 1)
        roofline["note"] = ("algorithmic flops (add/mul/div/sqrt = 1 each, counted by the oracle's op-counting build); a correctly "
                            "rounded f32 divide or sqrt costs 8-13 VALU instructions in this library (ieee_expanded_*); kernel_ms = HIP events on the launch stream")
        if not multi and not args.small:
            t, why = counters_for("c2_bench", rpt._lib.LIB_PATH)
            roofline["traffic_source"] = why + ("" if t is None else "; bench.py cannot collect counters itself")
            if t is not None:
                roofline["traffic"] = t["hbm_bytes_per_launch"]
            if t is not None and "valu_insts_per_launch" in t:
                # How busy the vector ALUs are, from the same committed PMC passes: VALU wave-instructions over the SIMDs' quad-cycles
                # of the profiled launch itself (GRBM_GUI_ACTIVE / 8: no clock is assumed).  2.0 per quad-cycle is the nominal issue
                # peak the 157 TFLOP/s figure assumes; a stream of nothing but independent v_fma_f32 reaches VALU_ISSUE_FMA_STREAM.
                cyc = t.get("shader_cycles_per_launch") or avg_kernel_s * GPU_CLOCK_HZ
                simd_quads = 1024.0 * cyc / 4.0
                rate = t["valu_insts_per_launch"] / simd_quads
                lanes = t.get("valu_lane_utilisation", 0.0)
                roofline["valu_issue"] = {
                    "insts_per_launch": t["valu_insts_per_launch"], "lane_utilisation": round(lanes, 3),
                    "insts_per_simd_quad_cycle": round(rate, 3),
                    "frac_of_dual_issue_peak": round(rate / 2.0, 3),
                    "frac_of_lane_issue_capacity": round(rate / 2.0 * lanes, 3),
                    "fma_stream_at_5_waves": VALU_ISSUE_FMA_STREAM,
                    "clock_GHz": round(cyc / (avg_kernel_s * 1e9), 3) if t.get("shader_cycles_per_launch") else None,
                    "note": "VALU wave-instructions per SIMD per 4 cycles, cycles counted (GRBM_GUI_ACTIVE / 8) in the same launch; 2.0 is the "
                            "nominal issue peak; frac_of_dual_issue_peak is the issue figure, x lane_utilisation = the share of the ALU lanes' "
                            "issue capacity doing path work; a stream of nothing but independent v_fma_f32 issues fma_stream_at_5_waves "
                            "at this occupancy (profiles/r5/valu_issue_peak.txt)"}
        out = {
            "metric": "Msamples/s (pixels x spp) on AnalyticalScene 1920x1080 f32; 1/2/4/8-GPU scaling",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "none" if not multi else "strong",     # one GPU: nothing scales; N > 1: configs[2]'s frame is the same for every N
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("AnalyticalScene %dx%d x %d spp per step, f32, seed 1 (BASELINE.json configs[1])" % (width, height, spp)) if not multi else
                                   ("AnalyticalScene %dx%d x %d spp per step, f32, seed 1 (BASELINE.json configs[2]): cyclic %d-row tiles over %d GPUs, "
                                    "RCCL gather to rank 0 + scatter per step inside the timed region" % (width, height, spp, TILE_ROWS, world)),
                       "spp_per_step": spp, "width": width, "height": height, "parallelism": "rows%d" % world,
                       **({"gather": gather_mode} if multi else {})},
            "roofline": roofline,
            "roofline_hbm": {"bound": "hbm", "achieved": round(hbm, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm / HBM_PEAK_GBS, 6),
                             "algorithmic_bytes_per_step": algo_bytes,
                             "note": "32/S bytes per pixel-sample: the honest signature of an ALU-bound path, not the binding roofline"},
        }
        out.update(dict(extra))
        if not multi and not args.headline_only:
            out.update(other_configs(rpt, torch, local_rank, args.small))
        if not multi and not args.no_cpu_baseline and not args.small:
            cpu = cpu_baseline(width, height)
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
            out["config1"] = config1_line(rpt, torch, local_rank, cpu["cores"])
            try:
                out["f64_reference"] = f64_reference(rpt, torch, local_rank, cpu["cores"])
            except Exception as e:      # noqa: BLE001 - a secondary leg must not cost the line
                out["f64_reference"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out))

    def secondary_legs():
        if os.environ.get("RPT_BENCH_TEST_STALL") == str(rank):         # tests: this rank never reaches the legs' first collective
            time.sleep(3600)
        # secondary: fixed work per GPU (weak scaling), a few steps
        ww, wh = weak_frame(world)
        ww, wh, wspp = shrink((ww, wh, C2[2]))
        wjob = job_of(ww, wh)
        wsteps = max(1, min(args.steps, 5))
        if wjob:
            wjob.render_n(wspp)
            wjob.gather()
        host_fence()
        tw = time.perf_counter()
        for _ in range(wsteps):
            if wjob:
                wjob.render_n(wspp)
                wjob.gather_begin()
        if wjob:
            wjob.gather_end()
        host_fence()
        tw = time.perf_counter() - tw
        t = torch.tensor([tw], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        extra["weak_scaling"] = {"workload": "AnalyticalScene %dx%d x %d spp per step (fixed pixels and samples per GPU)" % (ww, wh, wspp),
                                 "steps": wsteps, "ms_per_step": round(float(t.item()) / wsteps * 1e3, 3),
                                 "value": round(ww * wh * wspp * wsteps / float(t.item()) / 1e6, 2), "unit": "Msamples/s"}
        # the same configs[2] frame on ONE GPU (rank 0 alone), for the strong-scaling ratio
        if rank == 0:
            solo = rpt.Tracer(scene, device=local_rank, seed=1)
            sbuf = rpt.DeviceColorBuffer(width, height, device="cuda:%d" % local_rank)
            solo.render_n(sbuf, max(1, spp // 8))           # warm
            torch.cuda.synchronize()
            ts = time.perf_counter()
            solo.render_n(sbuf, spp)
            torch.cuda.synchronize()
            ts = time.perf_counter() - ts
            solo.close()
            del sbuf
            extra["strong_scaling"] = {"t1_ms": round(ts * 1e3, 3), "tN_ms": round(elapsed / args.steps * 1e3, 3),
                                       "speedup": round(ts / (elapsed / args.steps), 3), "n_gpus": world,
                                       "note": "same configs[2] frame rendered by rank 0 alone after the timed region (1 step)"}
        # secondary: BASELINE.json configs[4] as written — 10 k spheres + 16 lights, 4096 x 4096 x 512 spp, row-tiled over the N GPUs
        # (cyclic TILE_ROWS-row blocks), gathered to rank 0: one step, host clock, max over ranks.  The same contexts and communicator: the
        # scene is swapped (Tracer.scene() + upload_scene()).
        if not isinstance(job, TorchGatherRender):
            from rust_pathtracer_amd import scenes
            big = scenes.random_spheres_scene(10000, 16)
            if tracer:
                tracer._scene = big
                tracer.upload_scene()
            bw, bh, bspp = shrink((4096, 4096, 512))
            bjob = job_of(bw, bh)
            if bjob:
                bjob.render_n(max(1, bspp // 32))      # warm: device tables, the dispatch order, the communicator's buffers for this size
                bjob.gather()
            host_fence()
            tb = time.perf_counter()
            if bjob:
                bjob.render_n(bspp)
                bjob.gather_begin()
                bjob.gather_end()
            host_fence()
            tb = time.perf_counter() - tb
            t = torch.tensor([tb], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            extra["configs4"] = {"workload": "10k spheres + 16 lights %dx%d x %d spp in one step, cyclic %d-row tiles over %d GPUs, gather to rank 0 inside the "
                                             "timed region (BASELINE.json configs[4])" % (bw, bh, bspp, TILE_ROWS, world),
                                 "steps": 1, "ms_per_step": round(float(t.item()) * 1e3, 3),
                                 "value": round(bw * bh * bspp / float(t.item()) / 1e6, 2), "unit": "Msamples/s",
                                 "note": "render_large_regen_kernel; the one-GPU figure of the same frame is roofline_c5.value of the N = 1 line"}
            if tracer:
                tracer._scene = scene
                tracer.upload_scene()

    if not multi:
        # ---- one GPU: configs[1].  The frame is a torch tensor, the launches go to torch's current stream, and
        # torch.cuda.Event pairs on that stream time each step's kernel.
        width, height, spp = shrink(C2)
        tracer = rpt.Tracer(scene, device=local_rank, seed=1)
        buf = rpt.DeviceColorBuffer(width, height, device="cuda:%d" % local_rank)
        for _ in range(args.warmup):
            tracer.render_n(buf, spp)
        host_fence()
        evs = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                     # torch's current stream == the stream the kernel is launched on
            tracer.render_n(buf, spp)
            e1.record()
            evs.append((e0, e1))
        host_fence()
        elapsed = time.perf_counter() - t0
        kernel_ms = [a.elapsed_time(b) for a, b in evs]
        local_pixels = width * height
    else:
        # ---- N GPUs: configs[2], strong scaling, gather to rank 0 inside the timed region
        width, height, spp = shrink(C3)
        gather_mode = "library RCCL send/recv to rank 0"
        if args.smoke_shared_gpu:
            os.environ["RPT_GATHER"] = "p2p"
            tracer = None
            if rank == 0:                   # one process drives all "ranks" of cuda:0; the others only keep the barriers company
                tracer = rpt.Tracer(scene, devices=[0] * world, seed=1)
        else:
            # The library's own communicator (rpt_create_rank).  If it cannot be set up on this node, every rank falls
            # back TOGETHER to per-rank tiles + a torch.distributed (RCCL) gather, and the JSON line says so.
            why = ""
            t_init = t_first = None
            try:
                tc = time.perf_counter()
                tracer = tiling.rank_tracer(scene, local_rank, seed=1)                # rpt_comm_unique_id on rank 0, the id over gloo, ncclCommInitRank
                t_init = time.perf_counter() - tc
                tc = time.perf_counter()
                first = tiling.TiledRender(tracer, 64, 64, tile_rows=TILE_ROWS)       # the communicator's first exchange, on a small frame
                first.render_n(1)
                first.gather()
                t_first = time.perf_counter() - tc
                del first
            except Exception as e:          # noqa: BLE001 - any failure of the collective set-up takes the fallback
                tracer, why = None, "%s: %s" % (type(e).__name__, e)
            setup = {"rank": rank, "device": local_rank, "comm_init_s": None if t_init is None else round(t_init, 3),
                     "first_exchange_s": None if t_first is None else round(t_first, 3), "error": why or None}
            ok = torch.tensor([1 if tracer else 0])
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if tracer:
                    tracer.close()
                tracer = rpt.Tracer(scene, device=local_rank, seed=1)
                gather_mode = "torch.distributed RCCL gather (fallback: the library communicator failed to initialise%s)" % (
                    ": " + why if why else " on another rank")
            # Every rank says how its set-up went — on stderr at once (a run that dies later still leaves it in the log's tail) and, through
            # gloo, in rank 0's JSON line: a driver-run scaling file is then diagnosable from what it kept.
            setup["gather"] = gather_mode
            print("[bench rank %d/%d] %s" % (rank, world, json.dumps(setup)), file=sys.stderr, flush=True)
            all_setup = [None] * world
            dist.all_gather_object(all_setup, setup)
            extra["rank_setup"] = all_setup
        if gather_mode.startswith("torch"):
            job_of = lambda w, h: TorchGatherRender(tracer, tiling, w, h, TILE_ROWS, rank, world, local_rank)     # noqa: E731
        else:
            job_of = lambda w, h: tiling.TiledRender(tracer, w, h, tile_rows=TILE_ROWS) if tracer else None      # noqa: E731
        job = job_of(width, height)

        def step():
            if job:
                job.render_n(spp)
                job.gather_begin()          # enqueued behind the render on the library's streams: no host wait

        def drain():
            if job:
                job.gather_end()

        if job:                             # set the communicator's connections up outside the timed region, whatever --warmup is
            job.render_n(1)
            job.gather()
        for _ in range(args.warmup):
            step()
        drain()
        host_fence()
        kernel_ms = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            if isinstance(job, TorchGatherRender):
                kernel_ms.append(job.kernel_ms())
            elif tracer:
                kernel_ms.append(tracer.resident_kernel_ms())     # HIP events around this step's launches, on their stream
        drain()
        host_fence()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        local_pixels = tiling.tile_row_count(height, TILE_ROWS, rank, world) * width
        # The headline is measured.  What follows are secondary legs on the same communicator: they must never cost the line.
        # A leg that raises is recorded and ends the legs on this rank; ranks that then wait for it in a collective — or a leg that
        # hangs — are cut off by a timer that emits the line with what there is and leaves.
        def cut_off():
            # The legs hang (or a rank never reached them): the line goes out with what there is, and the process ends with a
            # status that says so — a hung rank must not look like success to the launcher.
            with emit_lock:
                if not emitted:
                    extra["secondary_legs"] = "cut off after %d s" % SECONDARY_LIMIT_S
            emit()
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(3)

        watchdog = threading.Timer(SECONDARY_LIMIT_S, cut_off)
        watchdog.daemon = True
        watchdog.start()
        try:
            secondary_legs()
        except Exception as e:              # noqa: BLE001
            with emit_lock:
                extra["secondary_legs"] = "stopped by %s: %s" % (type(e).__name__, e)
        dist.barrier()
        watchdog.cancel()

    emit()
    if multi:
        dist.barrier()
        if tracer:
            tracer.close()
        dist.destroy_process_group()
    else:
        tracer.close()


if __name__ == "__main__":
    main()
