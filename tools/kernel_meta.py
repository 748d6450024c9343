"""Register / scratch / LDS figures of every kernel in a built object (from the code object's metadata notes):
    python tools/kernel_meta.py [rust-pathtracer_amd/build/kernels.o] [substring]
What DESIGN.md quotes for spills and occupancy comes from here, i.e. from the shipped binary."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "rust-pathtracer_amd", "build", "kernels.o")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "co")
    fat = os.path.join(d, "fatbin")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], check=True, capture_output=True)
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", out], check=True, capture_output=True, text=True).stdout
rows = []
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if pat in name:
        rows.append((name, g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print("%-58s %5s %6s %5s %6s %8s %6s" % ("kernel", "vgpr", "vspill", "sgpr", "sspill", "scratchB", "ldsB"))
for r in sorted(rows):
    print("%-58s %5s %6s %5s %6s %8s %6s" % r)
