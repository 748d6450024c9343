"""rust/gpu_tracer.rs cannot be compiled here (no rustc), so its FFI surface is verified mechanically instead:
every `#[repr(C)]` struct is parsed and laid out by the repr(C) rules and compared — field order, field types,
offsets, size — with what gcc says about the struct of the same name in include/rpt.h; every function of the
`extern "C"` block is compared — argument count, argument types, return type — with the prototype in rpt.h.
CPU only."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "rust", "gpu_tracer.rs")
HEADER = os.path.join(ROOT, "include", "rpt.h")

RENAMES = {("RptLight", "light_type"): "type"}             # `type` is a Rust keyword
PRIMS = {"u8": (1, 1), "u32": (4, 4), "i32": (4, 4), "c_int": (4, 4), "u64": (8, 8), "f32": (4, 4), "c_char": (1, 1)}
C_PRIMS = {"uint8_t": "u8", "uint32_t": "u32", "int": "c_int", "uint64_t": "u64", "float": "f32", "char": "c_char", "void": "c_void", "size_t": "usize"}


def camel(c_name):
    return "".join(p.capitalize() for p in c_name.split("_"))


def strip_comments(src, rust=False):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


# ---- Rust side -------------------------------------------------------------------------------------------
def rust_type(t):
    t = t.strip()
    m = re.fullmatch(r"\[(.+);\s*(\d+)\]", t)
    if m:
        return ("array", rust_type(m.group(1)), int(m.group(2)))
    m = re.fullmatch(r"\*(const|mut)\s+(.+)", t)
    if m:
        return ("ptr", m.group(1), rust_type(m.group(2)))
    return ("name", t)


def split_top(s, sep=","):
    out, depth, cur = [], 0, ""
    for ch in s:
        depth += ch in "[(<"
        depth -= ch in "])>"
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def parse_rust(src):
    src = strip_comments(src)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+)\s*\{([^}]*)\}", src):
        fields = []
        for f in split_top(m.group(2)):
            fm = re.fullmatch(r"\s*(?:pub\s+)?(\w+)\s*:\s*(.+?)\s*", f, flags=re.S)
            assert fm, "cannot parse Rust field %r of %s" % (f, m.group(1))
            fields.append((fm.group(1), rust_type(fm.group(2))))
        structs[m.group(1)] = fields
    fns = {}
    ext = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S)
    assert ext, 'no extern "C" block'
    for m in re.finditer(r"fn (\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", ext.group(1), flags=re.S):
        args = []
        for a in split_top(m.group(2)):
            am = re.fullmatch(r"\s*(\w+)\s*:\s*(.+?)\s*", a, flags=re.S)
            assert am, "cannot parse Rust argument %r of %s" % (a, m.group(1))
            args.append(rust_type(am.group(2)))
        fns[m.group(1)] = (rust_type(m.group(3)) if m.group(3) else ("name", "c_void"), args)
    return structs, fns


def rust_layout(structs, t):
    """(size, align) of a Rust type under repr(C)."""
    if t[0] == "ptr":
        return 8, 8
    if t[0] == "array":
        s, a = rust_layout(structs, t[1])
        return s * t[2], a
    name = t[1]
    if name in PRIMS:
        return PRIMS[name]
    return rust_struct_layout(structs, name)[:2]


def rust_struct_layout(structs, name):
    off, align, offsets = 0, 1, {}
    for fname, ftype in structs[name]:
        s, a = rust_layout(structs, ftype)
        off = (off + a - 1) // a * a
        offsets[fname] = off
        off += s
        align = max(align, a)
    return (off + align - 1) // align * align, align, offsets


# ---- C side ----------------------------------------------------------------------------------------------
def c_type(t, macros):
    t = " ".join(t.split())
    depth = t.count("*")
    base = t.replace("*", " ").split()
    const = "const" in base
    base = [b for b in base if b not in ("const", "struct")]
    assert len(base) == 1, "cannot parse C type %r" % t
    name = base[0]
    core = ("name", C_PRIMS[name]) if name in C_PRIMS else ("name", camel(name))
    for level in range(depth):
        core = ("ptr", "const" if (const and level == 0) else "mut", core)
    return core


def parse_c(src):
    macros = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(\w+)\s+(\d+)u?\s*$", src, flags=re.M)}
    src = strip_comments(src)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    structs = {}
    for m in re.finditer(r"typedef struct (\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            first, *more = [d.strip() for d in decl.split(",")]
            dm = re.fullmatch(r"(.+?)\s*(\w+)\s*(?:\[(\w+)\])?", first)
            assert dm, "cannot parse C declaration %r in %s" % (decl, m.group(1))
            base = dm.group(1)
            for d in [dm.group(2) + ("[%s]" % dm.group(3) if dm.group(3) else "")] + more:
                nm = re.fullmatch(r"(\w+)\s*(?:\[(\w+)\])?", d)
                t = c_type(base, macros)
                if nm.group(2):
                    n = int(nm.group(2)) if nm.group(2).isdigit() else macros[nm.group(2)]
                    t = ("array", t, n)
                fields.append((nm.group(1), t))
        structs[m.group(3)] = fields
    fns = {}
    flat = re.sub(r"\{[^{}]*\}", "{}", src)                          # drop struct / enum bodies (extern "C" { survives as text)
    for stmt in flat.split(";"):
        m = re.search(r"((?:const\s+)?\w+\s*\**)\s*(rpt_\w+)\s*\(([^)]*)\)\s*$", stmt, flags=re.S)
        if not m:
            continue
        args = []
        arglist = " ".join(m.group(3).split())
        if arglist and arglist != "void":
            for a in arglist.split(","):
                am = re.fullmatch(r"\s*(.+?)\s*(\w+)\s*", a)
                assert am, "cannot parse C argument %r of %s" % (a, m.group(2))
                args.append(c_type(am.group(1), macros))
        fns[m.group(2)] = (c_type(m.group(1), macros), args)
    return structs, fns


def gcc_layout(c_structs, names, tmp_path):
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "rpt.h"', "int main(void) {"]
    for n in names:
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (n, n))
        for f, _ in c_structs[n]:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (n, f, n, f))
    lines += ["return 0; }"]
    prog = os.path.join(str(tmp_path), "rust_layout.c")
    open(prog, "w").write("\n".join(lines))
    exe = os.path.join(str(tmp_path), "rust_layout")
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), prog, "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    return {k: int(v) for k, v in (line.rsplit(" ", 1) for line in out.splitlines())}


def check_binding(rust_src, tmp_path):
    """-> list of mismatches between the Rust source text and include/rpt.h (empty = consistent)."""
    problems = []
    r_structs, r_fns = parse_rust(rust_src)
    c_structs, c_fns = parse_c(open(HEADER).read())
    by_rust_name = {camel(n): n for n in c_structs}
    mirrored = [by_rust_name[r] for r in r_structs if r in by_rust_name]
    truth = gcc_layout(c_structs, mirrored, tmp_path)
    for rname, rfields in r_structs.items():
        if rname == "RptCtx":
            continue                                                   # opaque handle
        if rname not in by_rust_name:
            problems.append("%s: no struct of that name in rpt.h" % rname)
            continue
        cname = by_rust_name[rname]
        cfields = c_structs[cname]
        rn = [RENAMES.get((rname, f), f) for f, _ in rfields]
        cn = [f for f, _ in cfields]
        if rn != cn:
            problems.append("%s: fields %s != rpt.h %s" % (rname, rn, cn))
            continue
        for (rf, rt), (cf, ct) in zip(rfields, cfields):
            if rt != ct:
                problems.append("%s.%s: type %s != rpt.h %s" % (rname, rf, rt, ct))
        size, _, offsets = rust_struct_layout(r_structs, rname)
        if size != truth[cname]:
            problems.append("%s: size %d != sizeof(%s) = %d" % (rname, size, cname, truth[cname]))
        for (rf, _), (cf, _) in zip(rfields, cfields):
            if offsets[rf] != truth["%s.%s" % (cname, cf)]:
                problems.append("%s.%s: offset %d != offsetof = %d" % (rname, rf, offsets[rf], truth["%s.%s" % (cname, cf)]))
    for fname, (rret, rargs) in r_fns.items():
        if fname not in c_fns:
            problems.append("%s: not declared in rpt.h" % fname)
            continue
        cret, cargs = c_fns[fname]
        if rret != cret:
            problems.append("%s: return type %s != rpt.h %s" % (fname, rret, cret))
        if len(rargs) != len(cargs):
            problems.append("%s: %d arguments != rpt.h %d" % (fname, len(rargs), len(cargs)))
            continue
        for i, (ra, ca) in enumerate(zip(rargs, cargs)):
            if ra != ca:
                problems.append("%s: argument %d is %s != rpt.h %s" % (fname, i, ra, ca))
    for must in ("rpt_create", "rpt_create_multi", "rpt_destroy", "rpt_upload_scene", "rpt_render", "rpt_last_error", "rpt_sizeof_scene_desc",
                 "rpt_resident_render", "rpt_resident_download_u8", "rpt_convert_to_u8", "rpt_denoise", "rpt_set_dispatch", "rpt_host_pin"):
        if must not in r_fns:
            problems.append("%s: not bound" % must)
    if not re.search(r"unsafe\s+impl\s+Send\s+for\s+GpuTracer", strip_comments(rust_src, rust=True)):
        problems.append("GpuTracer is not Send (the reference's Tracer is: scene.rs:5)")
    # every struct reachable from the scene descriptor must be mirrored
    for cname in ("rpt_scene_desc", "rpt_material", "rpt_sphere", "rpt_plane", "rpt_light", "rpt_camera", "rpt_background", "rpt_sdf", "rpt_sdf_prim"):
        if camel(cname) not in r_structs:
            problems.append("%s: no Rust mirror" % cname)
    # constants: every `pub const RPT_*` must have the header's value (gcc evaluates the enumerators), and every render /
    # scene flag and status code of the header must be there
    r_consts = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"pub const (RPT_[A-Z0-9_]+): [iu]32 = (-?(?:0x[0-9A-Fa-f]+|\d+));", rust_src)}
    hdr_names = sorted(set(re.findall(r"\b(RPT_(?:RENDER|SCENE|ERR|LIGHT|BG|MEDIUM)_[A-Z0-9_]+|RPT_OK|RPT_MAT_ALL|RPT_MAT_MEDIUM)\s*=", open(HEADER).read())))
    prog = os.path.join(str(tmp_path), "consts.c")
    with open(prog, "w") as f:
        f.write('#include <stdio.h>\n#include "rpt.h"\nint main(void) {\n')
        for n in hdr_names:
            f.write('    printf("%s %%lld\\n", (long long)%s);\n' % (n, n))
        f.write("    return 0;\n}\n")
    exe = os.path.join(str(tmp_path), "consts")
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), prog, "-o", exe], check=True)
    c_consts = {ln.split()[0]: int(ln.split()[1]) for ln in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()}
    for n in hdr_names:
        if n not in r_consts:
            problems.append("%s: constant not in the Rust binding" % n)
        elif r_consts[n] != c_consts[n]:
            problems.append("%s: %d != rpt.h %d" % (n, r_consts[n], c_consts[n]))
    for n in r_consts:
        if n != "RPT_ABI_VERSION" and n not in c_consts:
            problems.append("%s: no such constant in rpt.h" % n)
    m = re.search(r"pub const RPT_ABI_VERSION: u32 = (\d+);", rust_src)
    c_ver = re.search(r"#define RPT_ABI_VERSION (\d+)u", open(HEADER).read())
    if not m or not c_ver or m.group(1) != c_ver.group(1):
        problems.append("RPT_ABI_VERSION differs")
    return problems


def test_rust_binding_matches_header(tmp_path):
    problems = check_binding(open(RUST).read(), tmp_path)
    assert not problems, "\n".join(problems)


def test_checker_catches_a_short_scene_desc(tmp_path):
    """The round-1 bug: RptSceneDesc without its trailing `sdf` member (rpt_upload_scene then reads past the Rust
    object and rpt_scene_analytical writes past it).  The checker must flag it."""
    src = open(RUST).read()
    assert "    pub sdf: RptSdf,\n" in src
    broken = src.replace("    pub sdf: RptSdf,\n", "")
    problems = check_binding(broken, tmp_path)
    assert any("RptSceneDesc" in p for p in problems), problems


def test_checker_catches_a_wrong_signature_and_field_order(tmp_path):
    src = open(RUST).read()
    a = src.replace("frames_done: u64, spp: u32, seed: u64, flags: u32) -> c_int;", "frames_done: u32, spp: u32, seed: u64, flags: u32) -> c_int;", 1)
    assert a != src and any("rpt_render" in p for p in check_binding(a, tmp_path))
    b = src.replace("pub radius: f32, pub material: u32 }", "pub material: u32, pub radius: f32 }", 1)
    assert b != src and any("RptSphere" in p for p in check_binding(b, tmp_path))


def test_scene_desc_size_export_matches_c(rpt):
    import ctypes as C
    assert rpt.lib().rpt_sizeof_scene_desc() == C.sizeof(rpt._abi.rpt_scene_desc)


def test_checker_catches_a_wrong_flag_value(tmp_path):
    src = open(RUST).read()
    assert "pub const RPT_RENDER_RUSSIAN_ROULETTE: u32 = 0x20;" in src
    broken = src.replace("pub const RPT_RENDER_RUSSIAN_ROULETTE: u32 = 0x20;", "pub const RPT_RENDER_RUSSIAN_ROULETTE: u32 = 0x80;")
    assert any("RPT_RENDER_RUSSIAN_ROULETTE" in p for p in check_binding(broken, tmp_path))
