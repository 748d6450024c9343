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
    hdr_names = sorted(set(re.findall(r"\b(RPT_(?:RENDER|SCENE|ERR|LIGHT|BG|MEDIUM|MAT|PROC)_[A-Z0-9_]+|RPT_OK)\s*=", open(HEADER).read())))
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


# ---- the scene adapter: SceneDescBuilder, the From / patch helpers, AnalyticalScene's describe(), AutoTracer ------
ADAPTER = os.path.join(ROOT, "rust", "analytical_gpu.rs")


def _rust_fn_body(src, signature_start):
    """The text between the braces of the first fn whose signature starts with `signature_start`."""
    i = src.index(signature_start)
    j = src.index("{", i)
    depth, k = 0, j
    while True:
        depth += src[k] == "{"
        depth -= src[k] == "}"
        if depth == 0:
            return src[j + 1:k]
        k += 1


def test_the_rust_builder_describes_the_stock_scene_like_the_library(rpt):
    """rust/analytical_gpu.rs builds AnalyticalScene's descriptor from the Rust side's own values (lights through Scene::light_at,
    closest_hit's literals restated) with SceneDescBuilder.  Through the Python mirror of that builder and of that describe()
    (tests/scene_builder.py, statement for statement): every byte of the descriptor and of its four tables equals
    the library's rpt_scene_analytical."""
    import ctypes as C
    import scene_builder as sb

    def dump(d):
        out = [d.abi_version, d.flags, bytes(d.camera), bytes(d.background), d.eps, d.max_depth, d.n_spheres, d.n_planes, d.n_lights, d.n_materials, d.sdf.n_prims]
        out += [bytes(d.spheres[i]) for i in range(d.n_spheres)] + [bytes(d.planes[i]) for i in range(d.n_planes)]
        out += [bytes(d.lights[i]) for i in range(d.n_lights)] + [bytes(d.materials[i]) for i in range(d.n_materials)]
        return out

    lib_desc = rpt._abi.rpt_scene_desc()
    assert rpt.lib().rpt_scene_analytical(C.byref(lib_desc)) == 0
    built = sb.analytical_describe(sb.RefAnalyticalScene())
    assert built.with_desc(dump) == dump(lib_desc)
    # ... and it is a scene the library accepts as it is (argument checks run without a GPU)
    assert built.with_desc(lambda d: rpt.lib().rpt_upload_scene(None, C.byref(d))) == rpt._abi.RPT_ERR_INVALID_ARG      # (NULL context: the descriptor was not the problem)
    # a full patch zeroes nothing; the helper that canonicalises a patch keeps exactly the masked fields
    m = sb.RptMaterial.full(sb.RefMaterial()).zero_unmasked().c
    assert (m.ior, m.roughness, tuple(m.rgb)) == (C.c_float(1.45).value, 0.5, (1.5, 1.5, 1.5))
    m = sb.RptMaterial.patch(sb.RefMaterial(), rpt._abi.RPT_MAT_ROUGHNESS).zero_unmasked().c
    assert (m.ior, m.roughness, tuple(m.rgb)) == (0.0, 0.5, (0.0, 0.0, 0.0))


def test_the_python_mirror_is_the_rust_adapter():
    """The mirror can only stand for the Rust files if it IS them: SceneDescBuilder's and RptMaterial's public methods are the same
    set on both sides, and describe() of rust/analytical_gpu.rs and scene_builder.analytical_describe make the same builder calls
    with the same literals in the same order."""
    import inspect
    import scene_builder as sb
    rust = strip_comments(open(RUST).read())
    impl = rust[rust.index("impl SceneDescBuilder {"):rust.index("pub trait GpuScene")]
    rust_methods = set(re.findall(r"pub fn (\w+)", impl))
    py_methods = {n for n, _ in inspect.getmembers(sb.SceneDescBuilder, inspect.isfunction) if not n.startswith("_")}
    assert rust_methods - {"new", "flags"} == py_methods - {"set_flags"}, (rust_methods, py_methods)      # (`new` is __init__; `flags` is an attribute's name in Python)
    impl = rust[rust.index("impl RptMaterial {"):rust.index("pub struct SceneDescBuilder")]
    assert set(re.findall(r"pub fn (\w+)", impl)) == {n for n, _ in inspect.getmembers(sb.RptMaterial, inspect.isfunction) if not n.startswith("_")}
    # the two describe() bodies: builder calls and numeric literals, in order
    body = _rust_fn_body(strip_comments(open(ADAPTER).read()), "fn describe(&self)")
    py = inspect.getsource(sb.analytical_describe)
    py = py[py.index('"""', py.index('"""') + 3) + 3:]
    calls = lambda s: re.findall(r"\bb\s*\.\s*(\w+)\s*\(", s)                               # noqa: E731
    r_calls = [c for c in calls(body) if c != "materials"]
    p_calls = [c for c in calls(py) if c != "materials"]
    assert r_calls == p_calls, (r_calls, p_calls)
    nums = lambda s: [float(x) for x in re.findall(r"(?<![\w.])-?\d+\.\d+(?:e-?\d+)?", s)]   # noqa: E731
    assert nums(body.replace("F3::new_x(1.0)", "F3::new(1.0, 1.0, 1.0)")) == nums(py), (nums(body), nums(py))
    masks = lambda s: re.findall(r"RPT_MAT_\w+", s)                                          # noqa: E731
    assert masks(body) == masks(py)


def test_the_rust_surface_the_integration_guide_promises():
    """Items INTEGRATION.md names, with the signatures that make them a drop-in: a Result-returning constructor that hands the scene
    back, scene() typed like tracer.rs:629, render() that cannot panic on a library error, AutoTracer with the CPU fallback on
    RPT_ERR_NO_DEVICE, the conversions from the reference's own types."""
    src = strip_comments(open(RUST).read())
    for needle in ("pub struct RptError { pub status: i32, pub message: String }",
                   "pub fn try_new(scene: Box<dyn Scene>, describe: Describer) -> Result<Self, (RptError, Box<dyn Scene>)>",
                   "pub fn scene(&mut self) -> &mut Box<dyn Scene>",
                   "pub fn try_render(&mut self, buffer: &mut ColorBuffer) -> Result<(), RptError>",
                   "pub fn render(&mut self, buffer: &mut ColorBuffer)",
                   "pub enum AutoTracer", "AutoTracer::Cpu(Tracer::new(scene))", "pub fn backend(&self) -> &'static str",
                   "impl From<&AnalyticalLight> for RptLight", "impl From<&Light> for RptLight",
                   "pub fn patch(m: &Material, mask: u32) -> Self", "pub fn with_desc<R>(&self, f: impl FnOnce(&RptSceneDesc) -> R) -> R",
                   "pub type Describer = fn(&mut dyn Scene) -> Option<SceneDescBuilder>;",
                   "pub fn describer_of<T: GpuScene + 'static>(scene: &mut dyn Scene) -> Option<SceneDescBuilder>",
                   "pub fn no_device(&self) -> bool { self.status == RPT_ERR_NO_DEVICE }"):
        assert needle in src, needle
    # render() must not panic on a library error: no assert!/panic!/unwrap/expect inside it, nor inside AutoTracer::render
    gpu_impl = src[src.index("impl GpuTracer {"):src.index("impl Drop for GpuTracer")]
    render_body = _rust_fn_body(gpu_impl, "pub fn render(&mut self, buffer: &mut ColorBuffer)")
    auto_body = _rust_fn_body(src[src.index("impl AutoTracer {"):], "pub fn render(&mut self, buffer: &mut ColorBuffer)")
    for body in (render_body, auto_body, _rust_fn_body(gpu_impl, "pub fn try_render_n")):
        assert not re.search(r"\b(assert!|panic!|unwrap\(|expect\()", body), body
    adapter = strip_comments(open(ADAPTER).read())
    assert "impl GpuScene for AnalyticalScene" in adapter and "rpt_scene_analytical" not in adapter and "lights_of(self)" in adapter
    # every fn the extern block declares is used, or marked as deliberately unused
    ext = src[src.index('extern "C" {'):src.index("}", src.index('extern "C" {'))]
    for m in re.finditer(r"(#\[allow\(dead_code\)\]\s*)?fn (rpt_\w+)\(", ext):
        assert m.group(1) or src.count(m.group(2) + "(") >= 2, "%s is declared and never called" % m.group(2)


REFERENCE = "/root/reference"


def test_the_integration_patch_applies_to_the_reference_tree(tmp_path):
    """rust/integration.patch is everything a maintainer changes in the reference beside dropping the two source files in:
    `pub mod gpu_tracer;` (rust-pathtracer/src/lib.rs:12-22), the link stanza (build.rs + Cargo.toml), `mod analytical_gpu;` and
    the ONE changed statement of renderer/src/main.rs:41-42.  `git apply --check` against a copy of /root/reference (never the
    tree itself), then the patched tree is looked at the way rustc's module resolution would: every `mod` names a file that
    exists, the crate that forbids `unsafe` gained none, what main.rs imports is `pub` in gpu_tracer.rs, and `pt.render(&mut buffer)`
    / `buffer.convert_to_u8(frame)` (main.rs:118-122) are untouched."""
    import pytest
    import shutil
    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference tree is not on this machine (GPU box)")
    tree = tmp_path / "reference"
    shutil.copytree(REFERENCE, tree, ignore=shutil.ignore_patterns("images", ".git"))
    patch = os.path.join(ROOT, "rust", "integration.patch")
    before = (tree / "renderer" / "src" / "main.rs").read_text()
    for args in (["--check"], []):
        r = subprocess.run(["git", "apply", "-p1"] + args + [patch], cwd=tree, capture_output=True, text=True)
        assert r.returncode == 0, "git apply %s: %s" % (" ".join(args), r.stderr)
    shutil.copy(os.path.join(ROOT, "rust", "gpu_tracer.rs"), tree / "rust-pathtracer" / "src" / "gpu_tracer.rs")
    shutil.copy(os.path.join(ROOT, "rust", "analytical_gpu.rs"), tree / "renderer" / "src" / "analytical_gpu.rs")
    # module resolution: `mod x;` in a crate root names src/x.rs or src/x/mod.rs
    for crate, root_file in (("rust-pathtracer", "lib.rs"), ("renderer", "main.rs")):
        src = strip_comments((tree / crate / "src" / root_file).read_text())
        mods = re.findall(r"^\s*(?:pub\s+)?mod\s+(\w+)\s*;", src, flags=re.M)
        assert ("gpu_tracer" in mods) if crate == "rust-pathtracer" else ("analytical_gpu" in mods)
        for m in mods:
            assert (tree / crate / "src" / (m + ".rs")).exists() or (tree / crate / "src" / m / "mod.rs").exists(), (crate, m)
    main = (tree / "renderer" / "src" / "main.rs").read_text()
    # renderer is #![forbid(unsafe_code)] (main.rs:2): the file added to it must hold none
    assert "#![forbid(unsafe_code)]" in main
    assert not re.search(r"\bunsafe\b", strip_comments((tree / "renderer" / "src" / "analytical_gpu.rs").read_text()))
    # what main.rs now imports exists and is public; what analytical_gpu.rs implements is a public trait of the library crate
    lib = strip_comments((tree / "rust-pathtracer" / "src" / "gpu_tracer.rs").read_text())
    assert re.search(r"pub enum AutoTracer\b", lib) and re.search(r"pub fn describer_of\s*<", lib) and re.search(r"pub trait GpuScene\s*:\s*Scene", lib)
    assert "use rust_pathtracer::gpu_tracer::{AutoTracer, describer_of};" in main
    assert "AutoTracer::new(scene, describer_of::<AnalyticalScene>)" in main and "Tracer::new(scene)" not in main.replace("AutoTracer::new(scene", "")
    # the redraw handler is the reference's own, line for line: only additions above it and one statement changed
    changed = [l for l in before.splitlines() if l not in main.splitlines()]
    assert changed == ["    let mut pt = Tracer::new(scene);"], changed
    assert "pt.render(&mut buffer);" in main and "buffer.convert_to_u8(frame);" in main
    # the link stanza: a build script the manifest names, asking for the library include/rpt.h is the header of
    cargo = (tree / "rust-pathtracer" / "Cargo.toml").read_text()
    build = (tree / "rust-pathtracer" / "build.rs").read_text()
    assert 'build = "build.rs"' in cargo and "cargo:rustc-link-lib=dylib=rpt_hip" in build and "RPT_LIB_DIR" in build
    assert os.path.exists(os.path.join(ROOT, "rust-pathtracer_amd", "librpt_hip.so")), "the library the stanza links (built by __graft_entry__.build())"
    # AutoTracer keeps the three entry points the reference's Tracer has (tracer.rs:13, 22, 629) with their signatures
    auto = lib[lib.index("impl AutoTracer"):]
    assert re.search(r"pub fn new\(scene: Box<dyn Scene>, describe: Describer\) -> Self", auto)
    assert re.search(r"pub fn render\(&mut self, buffer: &mut ColorBuffer\)", auto)
    assert re.search(r"pub fn scene\(&mut self\) -> &mut Box<dyn Scene>", auto)
