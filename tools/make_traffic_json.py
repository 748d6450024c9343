#!/usr/bin/env python3
"""profiles/<round>/<variant>/bench_n1_pmc_{FETCH,WRITE}_SIZE.csv -> traffic.json (HBM bytes per launch
of the render kernel), corrected as MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced read, so it is doubled;
WRITE_SIZE is exact for 16 B/lane stores.  The two counters come from separate --pmc passes."""
import csv
import json
import os
import sys

d = sys.argv[1]


def last(fname, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(d, fname)))
            if r["Counter_Name"] == counter and "render_small" in r["Kernel_Name"]]
    return vals[-1]


fetch_kib = last("bench_n1_pmc_FETCH_SIZE.csv", "FETCH_SIZE")
write_kib = last("bench_n1_pmc_WRITE_SIZE.csv", "WRITE_SIZE")
out = {
    "kernel": "render_small_regen_kernel", "workload": "AnalyticalScene 1920x1080 x 256 spp per launch",
    "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
    "hbm_bytes_per_launch": fetch_kib * 1024 * 2 + write_kib * 1024,
    "correction": "FETCH_SIZE x2 (gfx950 half-count of 16 B/lane reads; uncalibrated for this kernel's 128-B row segments), WRITE_SIZE x1",
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
}
json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print(out)
