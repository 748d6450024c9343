"""HBM traffic of the denoiser's passes from rocprofv3 PMC passes of `tools/ab_time.py dn` (FETCH_SIZE and WRITE_SIZE in separate runs:
tools/collect_profiles.sh): per frame size the bytes one 3-iteration call moves against its algorithmic 3 x 32 B per pixel, and
where the rest must have come from (a 1080p RGBA f32 buffer and its ping-pong partner are 66 MB: they fit the 256 MB Infinity
Cache, so iteration i + 1 can read what iteration i wrote without touching HBM).
    python tools/denoise_traffic.py gpurun_out/prof_<variant> [launches per call: 1]  ->  <dir>/traffic_1080p.json, traffic_4k.json + a table"""
import collections
import csv
import json
import os
import sys

d = sys.argv[1]
sizes = {1920 * 1080: "1080p", 3840 * 2160: "4k"}
per = {k: collections.defaultdict(list) for k in sizes.values()}
kernels = set()
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # kernel launches per 3-iteration call: 1 since round 5 (the fused kernel), 3 before
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(os.path.join(d, "pmc_%s.csv" % name))):
        if "denoise" not in r["Kernel_Name"]:
            continue
        # one thread per pixel (whole 32 x 32 tiles of the fused kernel, 16 x 16 of the one-pass kernels): grid size ~ pixels
        px = min(sizes, key=lambda s: abs(s - int(r["Grid_Size"])))
        kernels.add(r["Kernel_Name"].split("(")[0])
        per[sizes[px]][name].append(float(r["Counter_Value"]))
# the share of the fused kernel's wave time spent waiting for LDS instructions (SQ_WAIT_INST_LDS of pass sq4 over SQ_WAVE_CYCLES of
# pass sq2, the same dispatches): what bounds it, since it moves a fraction of HBM's rate
lds_share = {}
try:
    acc = {k: collections.defaultdict(float) for k in sizes.values()}
    for fn, name in (("pmc_sq4.csv", "SQ_WAIT_INST_LDS"), ("pmc_sq2.csv", "SQ_WAVE_CYCLES")):
        for r in csv.DictReader(open(os.path.join(d, fn))):
            if "denoise" in r["Kernel_Name"] and r["Counter_Name"] == name:
                acc[sizes[min(sizes, key=lambda s: abs(s - int(r["Grid_Size"])))]][name] += float(r["Counter_Value"])
    for k, v in acc.items():
        if v.get("SQ_WAVE_CYCLES"):
            lds_share[k] = v.get("SQ_WAIT_INST_LDS", 0.0) / v["SQ_WAVE_CYCLES"]
except OSError:
    pass
for label, px in (("1080p", 1920 * 1080), ("4k", 3840 * 2160)):
    f, w = per[label]["FETCH_SIZE"], per[label]["WRITE_SIZE"]
    if not f or not w:
        continue
    calls = len(f) // passes
    fetch = sum(f) / calls * 1024.0 * 2.0                            # KiB -> B, gfx950 counts 16-B-per-lane reads at half (MI355X_MICROARCH.md)
    write = sum(w) / calls * 1024.0
    algo = 32.0 * px                                                 # the fused kernel: one read and one write of the frame (three separate passes: 3 x)
    out = {"kernels": sorted(kernels), "calls_averaged": calls, "hbm_bytes_per_launch": fetch + write,
           "hbm_read_bytes_per_call": fetch, "hbm_write_bytes_per_call": write, "algorithmic_bytes_per_call": algo,
           "correction": "FETCH_SIZE x2 (gfx950 half-count of 16 B/lane reads), WRITE_SIZE x1; summed over the call's launches",
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of tools/ab_time.py dn"}
    # which code the counters belong to (bench.py prints them only for the library they were collected from)
    import hashlib
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    rec = os.path.join(d, "loaded_lib.txt")           # what the profiled process loaded (collect_profiles.sh, _lib.py)
    sys.path.insert(0, root)
    import bench
    if os.path.exists(rec):
        out["library_sha256"] = open(rec).read().split()[0]
        out["source_sha256"] = bench.source_hash()
    if label in lds_share:
        out["wait_inst_lds_share"] = round(lds_share[label], 4)
    json.dump(out, open(os.path.join(d, "traffic_%s.json" % label), "w"), indent=1)
    print("%-6s %d calls: read %.1f MB + written %.1f MB = %.1f MB per call against %.1f MB algorithmic (%.2f x): %s" % (
        label, calls, fetch / 1e6, write / 1e6, (fetch + write) / 1e6, algo / 1e6, (fetch + write) / algo,
        "LDS wait share of wave time %.1f %%" % (100.0 * lds_share[label]) if label in lds_share else "no SQ passes"))
