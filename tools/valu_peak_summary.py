"""rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES of tools/microbench/valu_issue_peak ->
VALU wave-instructions per SIMD per quad-cycle at the MEASURED clock, per stream and waves per SIMD.   python tools/valu_peak_summary.py counters.csv trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(sys.argv[2]))}
by = {}
for r in rows:
    by.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"])})[r["Counter_Name"]] = float(r["Counter_Value"])
best = {}
for d, c in by.items():
    t = trace[d]
    ns = int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
    stream = c["name"].split("<")[1].split(">")[0] if "<" in c["name"] else "?"
    key = (stream, c["grid"] // 256 // 256)
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0                                   # summed over the 8 XCDs
    rate = c["SQ_INSTS_VALU"] / 1024.0 / cyc * 4.0
    if key not in best or ns < best[key][0]:
        best[key] = (ns, cyc / ns, rate, c["SQ_INSTS_VALU"])
names = {"0": "v_fma_f32", "1": "v_pk_fma_f32", "2": "fma/mul/add/cndmask"}
print("stream               waves/SIMD   time us   clock GHz   VALU wave-instructions per SIMD per quad-cycle (SQ_INSTS_VALU / 1024 / cycles x 4)")
for (stream, w), (ns, ghz, rate, n) in sorted(best.items()):
    print("%-20s %6d %11.1f %10.2f %10.2f" % (names.get(stream, stream), w, ns / 1e3, ghz, rate))
