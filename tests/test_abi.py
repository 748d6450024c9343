"""The C-ABI library loads without a GPU, exports every symbol include/rpt.h declares, and the
ctypes mirror of the structs matches the C layout.  No compute calls (CPU only)."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header="rpt.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rpt_[a-z0-9_]+)\s*\(", src)))


def _exported(lib_name):
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "rust-pathtracer_amd", lib_name)], check=True, capture_output=True, text=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if " T " in line)


def test_library_exports_every_declared_symbol_and_nothing_else(rpt):
    """The product library's dynamic symbol table is EXACTLY include/rpt.h (it is built with -fvisibility=hidden): no test hook, no
    launch wrapper, no device stub.  The test build adds exactly include/rpt_test.h."""
    names = _declared_functions()
    hooks = _declared_functions("rpt_test.h")
    assert len(names) >= 14 and len(hooks) >= 5 and not set(names) & set(hooks)
    assert _exported("librpt_hip.so") == names
    assert _exported("librpt_hip_test.so") == sorted(names + hooks)
    assert sorted(rpt._abi.SYMBOLS) == names, "ctypes mirror and header disagree"
    assert sorted(rpt._abi.TEST_SYMBOLS) == hooks, "ctypes mirror and test header disagree"
    assert rpt.lib().rpt_abi_version() == rpt._abi.RPT_ABI_VERSION
    product = C.CDLL(os.path.join(ROOT, "rust-pathtracer_amd", "librpt_hip.so"))
    product.rpt_build_has_test_hooks.restype = C.c_uint32
    assert product.rpt_build_has_test_hooks() == 0 and rpt.lib().rpt_build_has_test_hooks() == 1


def test_struct_layout_matches_c(rpt, tmp_path):
    prog = tmp_path / "layout.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "rpt.h"
#define S(t) printf(#t " %zu\n", sizeof(t))
#define O(t, f) printf(#t "." #f " %zu\n", offsetof(t, f))
int main(void) {
  S(rpt_material); S(rpt_sphere); S(rpt_plane); S(rpt_light); S(rpt_camera); S(rpt_background); S(rpt_scene_desc);
  O(rpt_material, rgb); O(rpt_material, ior); O(rpt_material, proc_params); O(rpt_material, medium_type); O(rpt_material, medium_color); O(rpt_material, medium_anisotropy);
  O(rpt_light, radius); O(rpt_light, area); O(rpt_plane, min_denom);
  O(rpt_scene_desc, camera); O(rpt_scene_desc, background); O(rpt_scene_desc, eps); O(rpt_scene_desc, spheres);
  O(rpt_scene_desc, planes); O(rpt_scene_desc, lights); O(rpt_scene_desc, n_materials); O(rpt_scene_desc, materials);
  return 0; }''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = dict(line.rsplit(" ", 1) for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    A = rpt._abi
    for name in ("rpt_material", "rpt_sphere", "rpt_plane", "rpt_light", "rpt_camera", "rpt_background", "rpt_scene_desc"):
        assert C.sizeof(getattr(A, name)) == int(out[name]), name
    for key, val in out.items():
        if "." in key:
            t, f = key.split(".")
            assert getattr(getattr(A, t), f).offset == int(val), key


def test_no_gpu_means_loud_failure_not_fallback(rpt):
    """Without a device the product refuses to run (this container has no GPU); with one, the
    call succeeds.  Either way nothing is computed on the CPU."""
    import torch
    h = C.c_void_p()
    rc = rpt.lib().rpt_create(C.byref(h), 0)
    if torch.cuda.is_available():
        assert rc == 0
        rpt.lib().rpt_destroy(h)
    else:
        assert rc == rpt._abi.RPT_ERR_NO_DEVICE
        assert b"no CPU fallback" in rpt.lib().rpt_last_error(None)
        try:
            rpt.Tracer(rpt.AnalyticalScene())
            raise AssertionError("Tracer() must raise without a GPU")
        except rpt.RptError as e:
            assert e.status == rpt._abi.RPT_ERR_NO_DEVICE


def test_scene_builders_agree(rpt, oracle):
    """rpt_scene_analytical (product), AnalyticalScene().describe() (host mirror) and the oracle's own
    transcription of renderer/src/analytical.rs are byte-identical."""
    def dump(d):
        out = [d.abi_version, d.flags, bytes(d.camera), bytes(d.background), d.eps, d.max_depth,
               d.n_spheres, d.n_planes, d.n_lights, d.n_materials]
        out += [bytes(d.spheres[i]) for i in range(d.n_spheres)] + [bytes(d.planes[i]) for i in range(d.n_planes)]
        out += [bytes(d.lights[i]) for i in range(d.n_lights)] + [bytes(d.materials[i]) for i in range(d.n_materials)]
        return out
    p = rpt._abi.rpt_scene_desc()
    assert rpt.lib().rpt_scene_analytical(C.byref(p)) == 0
    s = rpt.AnalyticalScene()
    assert dump(p) == dump(oracle.scene_analytical()) == dump(s.describe())
    assert (p.n_spheres, p.n_planes, p.n_lights, p.max_depth) == (2, 1, 1, 4)


def test_argument_validation_without_gpu(rpt):
    lib = rpt.lib()
    assert lib.rpt_create(None, 0) == rpt._abi.RPT_ERR_INVALID_ARG
    assert lib.rpt_render(None, None, 1, 1, 0, 1, 1, 0) == rpt._abi.RPT_ERR_INVALID_ARG
    assert lib.rpt_upload_scene(None, None) == rpt._abi.RPT_ERR_INVALID_ARG
    assert lib.rpt_scene_analytical(None) == rpt._abi.RPT_ERR_INVALID_ARG
    lib.rpt_destroy(None)                                     # harmless


def test_multi_gpu_entry_points_validate_without_gpu(rpt):
    """The multi-GPU constructors and helpers reject bad arguments before touching a device (and, on this GPU-less box,
    fail loudly with RPT_ERR_NO_DEVICE for good ones: no CPU fallback)."""
    import torch
    lib, A = rpt.lib(), rpt._abi
    h = C.c_void_p()
    ids = (C.c_int * 2)(0, 1)
    assert lib.rpt_create_multi(None, ids, 2) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_create_multi(C.byref(h), None, 2) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_create_multi(C.byref(h), ids, 0) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_create_multi(C.byref(h), ids, 65) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_comm_unique_id(None) == A.RPT_ERR_INVALID_ARG
    uid = A.rpt_unique_id()
    assert lib.rpt_create_rank(C.byref(h), 0, 2, 2, C.byref(uid)) == A.RPT_ERR_INVALID_ARG      # rank >= world
    assert lib.rpt_create_rank(C.byref(h), 0, 0, 0, C.byref(uid)) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_create_rank(C.byref(h), 0, 0, 1, None) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_set_tile_rows(None, 2) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_set_dispatch(None, 1, 12, 64, 0) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_host_pin(None, 16) == A.RPT_ERR_INVALID_ARG and lib.rpt_host_unpin(None) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_build_has_test_hooks() in (0, 1)
    assert lib.rpt_debug_sched_read(None, None, 0, None) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_world(None, None, None, None) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_resident_gather_device(None, None) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_resident_sync(None) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_resident_upload(None, None, 1, 1, 0) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_resident_kernel_ms(None, None) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_probe_fn(None, 0, None, None, 0, None, None) == A.RPT_ERR_INVALID_ARG
    assert lib.rpt_tile_rows_padded(1080, 2, 8) == 136 and lib.rpt_tile_rows_padded(2160, 2, 8) == 270
    assert lib.rpt_tile_rows_padded(7, 2, 4) == 2 and lib.rpt_tile_rows_padded(10, 0, 4) == 0
    if not torch.cuda.is_available():
        assert lib.rpt_create_multi(C.byref(h), ids, 2) == A.RPT_ERR_NO_DEVICE
        assert b"no CPU fallback" in lib.rpt_last_error(None)
        try:
            rpt.Tracer(rpt.AnalyticalScene(), devices=[0, 1])
            raise AssertionError("a multi-device Tracer must raise without a GPU")
        except rpt.RptError as e:
            assert e.status == A.RPT_ERR_NO_DEVICE


def test_tile_row_maps(rpt):
    """Cyclic row-block tiling: every row belongs to exactly one rank, in order within a rank."""
    from rust_pathtracer_amd import tiling
    for height, tile_rows, world in ((1080, 2, 8), (1080, 16, 8), (54, 4, 3), (7, 2, 4), (2160, 8, 8), (5, 8, 2), (9, 1, 9)):
        owner = {}
        for r in range(world):
            rows = tiling.tile_global_rows(height, tile_rows, r, world)
            assert rows == sorted(rows)
            assert len(rows) == tiling.tile_row_count(height, tile_rows, r, world)
            for g in rows:
                assert g not in owner and (g // tile_rows) % world == r
                owner[g] = r
        assert sorted(owner) == list(range(height))
        counts = [tiling.tile_row_count(height, tile_rows, r, world) for r in range(world)]
        assert max(counts) - min(counts) <= tile_rows


def test_rccl_that_cannot_be_loaded_is_an_error_not_a_crash(rpt):
    """rpt_comm_unique_id / rpt_create_rank with an RCCL library that cannot be loaded return RPT_ERR_RCCL with dlopen's
    message (round 2 read dlerror() twice and built a std::string from NULL).  RPT_RCCL_LIB is read once per process: child."""
    import sys
    code = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
import conftest
rpt = conftest.load_package()
uid = rpt._abi.rpt_unique_id()
for _ in range(2):                                 # the second call takes the cached failure
    rc = rpt.lib().rpt_comm_unique_id(C.byref(uid))
    msg = rpt.lib().rpt_last_error(None)
    assert rc == rpt._abi.RPT_ERR_RCCL, rc
    assert b"cannot load RCCL" in msg and b"no-such-rccl" in msg, msg
h = C.c_void_p()
assert rpt.lib().rpt_create_rank(C.byref(h), 0, 0, 1, C.byref(uid)) == rpt._abi.RPT_ERR_RCCL
print("ok")
""" % os.path.join(ROOT, "tests")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, RPT_RCCL_LIB="/no-such-rccl/librccl.so.1"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
