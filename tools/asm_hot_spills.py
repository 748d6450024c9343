"""Which spill instructions of small scenes' megakernel sit in blocks that run (device assembly: hipcc --offload-device-only -S).  A block
that holds v_div_scale / v_div_fmas belongs to the second computation of a flagged sample (namespace rptplain: hipcc's divide) and
practically never runs; every other block is the kernel proper.   python tools/asm_hot_spills.py kernels.s [kernel substring]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else "render_small_regen_kernelN"
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % pat, l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", []]
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), []]
    elif l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
        cur[1].append(l.strip())
blocks.append(cur)
hot = cold = 0
for name, ins in blocks:
    sc = [i for i in ins if i.startswith("scratch_")]
    if not sc:
        continue
    plain = any("v_div_scale" in i or "v_div_fmas" in i for i in ins)
    if plain:
        cold += len(sc)
    else:
        hot += len(sc)
        print("%-12s %4d instructions: %s" % (name, len(ins), ", ".join(s.split()[0].replace("scratch_", "") + " @" + s.split("offset:")[-1].split()[0] if "offset:" in s else s.split()[0] for s in sc)))
print("scratch instructions in running blocks: %d, in the second computation's blocks: %d" % (hot, cold))
