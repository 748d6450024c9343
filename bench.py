#!/usr/bin/env python3
"""bench.py — Msamples/s of the render hot path on MI355X (driver contract: see the task).

One "step" = one pass of the hot path over one batch: `SPP` (256) consecutive
Tracer::render() calls folded into one launch over the whole frame, on a frame that is
already resident in HBM; for N > 1 every step's per-rank row tiles are all-gathered over RCCL
(the image is row-tiled across the GPUs; ranks exchange nothing while rendering) and scattered into
the full image on every rank.  The gather of step k runs on RCCL's stream while step k+1 renders
(TiledRender.gather_begin / gather_end); all K gathers complete inside the timed region.

N = 1 workload = BASELINE.json configs[1]: AnalyticalScene 1920x1080, 256 spp, f32.
N > 1 is WEAK scaling: the same view at round(1920*sqrt(N)) x round(1080*sqrt(N)) pixels
(N = 4 is 3840x2160, configs[2]'s frame), i.e. a fixed number of pixels and samples per GPU.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]
       (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPP = 256
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector


def frame_size(n_gpus):
    s = math.sqrt(n_gpus)
    w = int(round(1920 * s / 8.0)) * 8
    h = int(round(1080 * s / 8.0)) * 8
    return w, h


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


def cpu_baseline(width, height, budget_s=12.0):
    """Time the CPU oracle (a port of the reference's rayon path: OpenMP over scanlines) on
    the host cores, on a bounded sample of the same workload: the full 1920x1080 frame at a
    few spp (Msamples/s does not depend on spp on the CPU).  The build timed is the one with the
    PLATFORM libm (liboracle_libm.so: glibc sinf/cosf/powf/log2f, what the reference's Rust f32 methods
    call on Linux) — 1.6x faster than the bit-reproducible strict-math build the parity tests use, so it
    is the fairer stand-in for the reference binary.  The thread count is the best of
    {quota, 2 x quota} CPUs (cgroup-aware: oversubscribing a quota makes the baseline slower,
    which would flatter the GPU).  Also counts flops per sample."""
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
    # algorithmic flops per sample, measured by the op-counting build of the oracle
    oc = oracle_lib.Oracle("liboracle_opcount.so")
    cw, ch, cs = 240, 136, 4
    c = oc.opcount(oc.scene_analytical(), cw, ch, cs, seed=1)
    n = cw * ch * cs
    flops = (c["add"] + c["mul"] + c["div"] + c["sqrt"]) / n
    return {
        "value": round(msps, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": "%dx%d x %d spp, same scene/seed (%.1f s of CPU work); OpenMP scanline loop, g++ -O3 -march=x86-64-v3, glibc libm; "
                  "%d threads on a %d-CPU quota (%d logical CPUs visible)"
                  % (width, height, spp, t1 - t0, threads, quota, os.cpu_count() or 0),
    }, {"flops_per_sample": round(flops, 1), "transcendentals_per_sample": round(c["transc"] / n, 2),
        "divides_per_sample": round(c["div"] / n, 2), "sqrts_per_sample": round(c["sqrt"] / n, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (profiling runs)")
    ap.add_argument("--smoke-shared-gpu", action="store_true",
                    help="TEST ONLY: all ranks use cuda:0 and the gloo backend (RCCL rejects duplicate devices), to exercise "
                         "the N>1 code path on a 1-GPU box; the numbers mean nothing")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as entry
    rpt = entry._load_package()
    from rust_pathtracer_amd import tiling

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    dist = None
    if args.smoke_shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.smoke_shared_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        warm = torch.zeros(1, device="cuda")
        dist.all_reduce(warm)                   # create the communicator now, not inside the first timed gather (--warmup 0)
        torch.cuda.synchronize()

    width, height = frame_size(world)
    tracer = rpt.Tracer(rpt.AnalyticalScene(), device=local_rank, seed=1)
    job = tiling.TiledRender(tracer, width, height, tile_rows=2)

    def step():
        job.render_n(SPP)
        return job.gather() if world > 1 else None

    def finish_gather(pending):
        return job.gather_end(pending) if pending is not None else None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms = []
    evs = []
    t0 = time.perf_counter()
    pending = None
    for _ in range(args.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()                         # torch's current stream == the stream the kernel is launched on
        job.render_n(SPP)
        e1.record()
        evs.append((e0, e1))
        if world > 1:
            finish_gather(pending)          # the previous step's gather ran while this step rendered
            pending = job.gather_begin()
    finish_gather(pending)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        samples = width * height * SPP * args.steps
        value = samples / elapsed / 1e6
        avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
        local_pixels = job.rows * width
        # HBM roofline of the megakernel: the only framebuffer traffic is one 16 B read + one
        # 16 B write of the RGBA-f32 running mean per pixel per launch (32/S B
        # per pixel-sample x pixels*S samples per launch).
        algo_bytes = 32.0 * local_pixels
        achieved = algo_bytes / avg_kernel_s / 1e9
        out = {
            "metric": "Msamples/s (pixels x spp) on AnalyticalScene 1920x1080 f32",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "AnalyticalScene %dx%d x %d spp per step, f32, seed 1%s" %
                                   (width, height, SPP, "" if world == 1 else ", cyclic 2-row tiles over %d GPUs + RCCL all-gather per step" % world),
                       "spp_per_step": SPP, "width": width, "height": height, "parallelism": "rows%d" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                         "kernel": "render_small_regen_kernel", "kernel_ms": round(avg_kernel_s * 1e3, 3),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "the path is FP32-VALU bound (no dense contraction, 32/S bytes per sample): see roofline_valu"},
        }
        # HBM bytes per launch measured with rocprofv3 PMC passes of this same command (committed
        # under profiles/; bench.py cannot collect counters itself)
        tj = os.path.join(ROOT, "profiles", "r1", "v6_max_ilp", "traffic.json")
        if world == 1 and os.path.exists(tj):
            t = json.load(open(tj))
            out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = "profiles/r1/v6_max_ilp/traffic.json (%s)" % t["correction"]
        if world == 1 and not args.no_cpu_baseline:
            cpu, ops = cpu_baseline(width, height)
            out["cpu_baseline"] = cpu
            tfl = ops["flops_per_sample"] * local_pixels * SPP / avg_kernel_s / 1e12
            # the same work with every correctly rounded divide / sqrt counted at the 12 / 15 VALU instructions
            # (~2 flops each where they are fmas) gfx950 needs for it: what the VALU actually has to issue
            expanded = ops["flops_per_sample"] + ops["divides_per_sample"] * (2 * 12 - 1) + ops["sqrts_per_sample"] * (2 * 15 - 1)
            out["roofline_valu"] = {"bound": "fp32_valu", "achieved": round(tfl, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(tfl / FP32_PEAK_TFLOPS, 5), **ops,
                                    "ieee_expanded_flops_per_sample": round(expanded, 1),
                                    "ieee_expanded_frac": round(expanded * local_pixels * SPP / avg_kernel_s / 1e12 / FP32_PEAK_TFLOPS, 5),
                                    "note": "algorithmic flops (add/mul/div/sqrt = 1 each, counted by the oracle's op-counting build); "
                                            "a correctly rounded f32 divide or sqrt costs 12-15 VALU instructions on gfx950"}
            out["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    tracer.close()


if __name__ == "__main__":
    main()
