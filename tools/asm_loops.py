"""Loop structure of one kernel in a device assembly listing (hipcc --offload-device-only -S): every backward branch with its
instruction count and how many scratch (spill), sqrt, LDS and ballot-count instructions the loop body holds — where the
spills sit relative to the hot loops.   python tools/asm_loops.py kernels.s <kernel substring> [min instrs]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 100
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % pat, l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
is_ins = lambda l: l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")
print(lines[start].split(":")[0], "instructions:", sum(map(is_ins, body)), "scratch:", sum("scratch_" in l for l in body))
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
seen = set()
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        seg = body[a:i + 1]
        n = sum(map(is_ins, seg))
        if n >= lo and (a, n // 8) not in seen:
            seen.add((a, n // 8))
            print("  %-12s %6d instrs  scratch %3d  sqrt %3d  ds %3d  bcnt %2d  rcp %3d" % (m.group(1), n, sum("scratch_" in x for x in seg),
                  sum("v_sqrt_f32" in x for x in seg), sum("\tds_" in x for x in seg), sum("s_bcnt1" in x for x in seg), sum("v_rcp_f32" in x for x in seg)))
