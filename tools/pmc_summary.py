"""Summarise a tools/collect_profiles.sh output directory for the kernels matching a substring: the last dispatch's
counters, the derived figures DESIGN.md quotes (lane utilisation, waves per SIMD, HBM bytes per launch with the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md, instruction mix), and traffic.json."""
import collections
import csv
import glob
import json
import os
import sys

d, pat = sys.argv[1], sys.argv[2]
vals, kernels = {}, collections.Counter()
for f in sorted(glob.glob(os.path.join(d, "pmc_*.csv"))):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kernels[r["Kernel_Name"].split("(")[0]] += 1
    for k, v in per.items():
        vals[k] = sum(v) if os.environ.get("PMC_AGG") == "sum" else v[-1]      # PMC_AGG=sum: all dispatches (multi-launch forms)
print("kernels matched:", ", ".join(sorted(kernels)))
for k in sorted(vals):
    print("%-28s %.5g" % (k, vals[k]))
g = vals.get
if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
    print("VALU lane utilisation        %.1f %%" % (100.0 * g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))))
if g("SQ_WAVE_CYCLES") and g("GRBM_GUI_ACTIVE"):
    print("waves resident per SIMD      %.2f" % (4.0 * g("SQ_WAVE_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0)))
if g("SQ_WAVE_CYCLES"):
    for name, key in (("issuing", "SQ_ACTIVE_INST_ANY"), ("issue-stalled", "SQ_WAIT_INST_ANY"), ("waiting (s_waitcnt)", "SQ_WAIT_ANY")):
        if g(key):
            print("wave time %-18s %.1f %%" % (name, 100.0 * g(key) / g("SQ_WAVE_CYCLES")))
if g("SQ_INSTS_VALU"):
    tot = g("SQ_INSTS_VALU")
    known = 0.0
    for key in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT",
                "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"):
        if g(key) is not None:
            known += g(key)
            print("  %-26s %5.1f %% of VALU instructions" % (key[14:], 100.0 * g(key) / tot))
    print("  %-26s %5.1f %% (moves, selects, compares, ...)" % ("other", 100.0 * (tot - known) / tot))
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    out = {"kernels": sorted(kernels), "FETCH_SIZE_KiB": g("FETCH_SIZE"), "WRITE_SIZE_KiB": g("WRITE_SIZE"),
           "hbm_bytes_per_launch": g("FETCH_SIZE") * 1024 * 2 + g("WRITE_SIZE") * 1024,
           "correction": "FETCH_SIZE x2 (gfx950 half-count of 16 B/lane reads; uncalibrated for this kernel's 128-B row segments), WRITE_SIZE x1",
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/collect_profiles.sh)"}
    if g("SQ_INSTS_VALU"):
        out["valu_insts_per_launch"] = g("SQ_INSTS_VALU")
    if g("SQ_ACTIVE_INST_VALU"):
        out["valu_active_quad_cycles_per_launch"] = g("SQ_ACTIVE_INST_VALU")
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        out["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
    if g("GRBM_GUI_ACTIVE"):
        out["shader_cycles_per_launch"] = g("GRBM_GUI_ACTIVE") / 8.0            # the counter sums the 8 XCDs
    # Which code the counters belong to: bench.py prints them only for the library (or the sources) they were collected from.
    import hashlib
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    # The hash of the library the PROFILED PROCESS loaded (it leaves "<sha256> <path>" in loaded_lib.txt: collect_profiles.sh,
    # _lib.py); without that record nothing is stamped and bench.py will not print these counters as the loaded library's.
    rec = os.path.join(d, "loaded_lib.txt")
    if os.path.exists(rec):
        out["library_sha256"], out["library_path"] = open(rec).read().split(None, 1)
        out["library_path"] = os.path.relpath(out["library_path"].strip(), os.path.abspath(root))
    else:
        print("NOT STAMPED: %s is missing (the profiled program did not load the library through rust-pathtracer_amd/_lib.py)" % rec)
    sys.path.insert(0, root)
    try:
        import bench
        if "library_sha256" in out:
            out["source_sha256"] = bench.source_hash()
    except Exception as e:      # noqa: BLE001
        out["source_sha256_error"] = str(e)
    json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
    print("HBM bytes per launch         %.4g (2 x FETCH + WRITE)" % out["hbm_bytes_per_launch"])
ks = os.path.join(d, "kernel_stats.csv")
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        if pat in r["Name"]:
            print("kernel_stats: %s calls=%s avg=%.3f ms" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e6))
