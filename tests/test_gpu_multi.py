"""Multi-GPU behind the C ABI, on the one GPU a test box has (needs an MI355X).

  * rpt_create_multi with n = 1: the same code path an 8-device context takes (tiles, ncclCommInitAll, grouped
    ncclSend / ncclRecv to rank 0, scatter kernel) with a one-rank communicator; every entry point must give the
    single-device image bit for bit.
  * rpt_create_rank with world = 1: the one-process-per-GPU construction (unique id, ncclCommInitRank).
  * virtual ranks: RPT_GATHER=p2p accepts the same device id several times and gathers with peer copies instead of
    RCCL (which needs distinct devices): n = 2, 3, 8 ranks of the real tiling, copies, gather offsets and scatter
    through the real entry points.
Bit-identical to the oracle throughout (the image must not depend on the number of ranks)."""
import ctypes as C
import os

import numpy as np
import pytest

import conftest
from test_gpu_parity import assert_bit_identical

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def _exercise(rpt, oracle, t, w, h, torch, what):
    """render (host buffer), resident render in two steps, download f32 / u8, device gather, resume from a host buffer."""
    want1 = oracle.render(oracle.scene_analytical(), w, h, 2, seed=1)
    want2 = oracle.render(oracle.scene_analytical(), w, h, 5, seed=1)
    buf = rpt.ColorBuffer(w, h)
    t.render_n(buf, 2)
    assert buf.frames == 2
    assert_bit_identical(buf.image(), want1, what + ": rpt_render, host buffer")
    t.render_n(buf, 3)                                                # resume through the host buffer
    assert_bit_identical(buf.image(), want2, what + ": rpt_render resumed")

    t.resident_reset()
    t.render_resident(w, h, 2)
    got = t.resident_to_host(w, h)
    assert got.frames == 2
    assert_bit_identical(got.image(), want1, what + ": resident step 1")
    t.render_resident(w, h, 3)
    assert t.resident_frames() == 5
    img = torch.empty(h, w, 4, dtype=torch.float32, device="cuda:0")
    t.resident_gather(img)
    t.resident_sync()
    assert_bit_identical(img.cpu().numpy(), want2, what + ": rpt_resident_gather_device")
    assert t.resident_kernel_ms() > 0.0
    u8 = t.resident_to_u8(w, h)
    want_u8 = np.zeros(w * h * 4, dtype=np.uint8)
    oracle.lib.oracle_convert_to_u8(want2.ctypes.data, want_u8.ctypes.data, w, h)
    assert np.array_equal(u8, want_u8), what + ": rpt_resident_download_u8"

    t.resident_reset()                                                # resume a cloned host ColorBuffer in the resident buffer
    half = rpt.ColorBuffer(w, h)
    half.pixels[:] = want1.reshape(-1)
    half.frames = 2
    t.resident_upload(half)
    t.render_resident(w, h, 3)
    assert_bit_identical(t.resident_to_host(w, h).image(), want2, what + ": rpt_resident_upload + render")


@pytest.mark.parametrize("w,h", [(72, 54), (40, 27)])
def test_multi_context_with_one_device_uses_rccl_and_matches(rpt, oracle, torch_cuda, w, h):
    os.environ.pop("RPT_GATHER", None)
    t = rpt.Tracer(rpt.AnalyticalScene(), devices=[0], seed=1)
    assert t.world() == (0, 1, 1)
    _exercise(rpt, oracle, t, w, h, torch_cuda, "rpt_create_multi n=1 (RCCL)")
    t.close()


def test_rank_context_world_one(rpt, oracle, torch_cuda):
    uid = rpt.comm_unique_id()
    assert len(uid) == 128
    t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1, rank=0, world=1, unique_id=uid)
    assert t.world() == (0, 1, 1)
    _exercise(rpt, oracle, t, 72, 54, torch_cuda, "rpt_create_rank world=1 (RCCL)")
    t.close()


@pytest.mark.parametrize("n,tile_rows,w,h", [(2, 2, 72, 54), (3, 4, 72, 54), (8, 2, 72, 54), (4, 16, 40, 27), (8, 1, 24, 5), (2, 5, 40, 27)])
def test_virtual_ranks_through_the_multi_context(rpt, oracle, torch_cuda, n, tile_rows, w, h):
    """n ranks on one device (peer-copy gather): the real tiling arithmetic, strided host copies, gather offsets and
    scatter kernel of an n-GPU context.  Sizes include ranks that own no row and a short last block."""
    os.environ["RPT_GATHER"] = "p2p"
    try:
        t = rpt.Tracer(rpt.AnalyticalScene(), devices=[0] * n, seed=1)
    finally:
        os.environ.pop("RPT_GATHER", None)
    assert t.world() == (0, n, n)
    t.set_tile_rows(tile_rows)
    _exercise(rpt, oracle, t, w, h, torch_cuda, "virtual ranks n=%d tile_rows=%d" % (n, tile_rows))
    t.close()


@pytest.mark.parametrize("pin", ["1", "0"])
def test_multi_device_render_fans_out(rpt, oracle, torch_cuda, pin):
    """rpt_render on an n-device context is the reference's parallel render() (tracer.rs:29-32): every device must have BEGUN
    its rows before the first one has finished.  Virtual ranks on one GPU: the event of rank k's begin (recorded on its stream
    before its upload) must precede the event of rank 0's end (recorded behind its kernel) — with one loop doing upload ->
    kernel -> download per device on a pageable buffer (round 2's code) rank 1 begins only after rank 0's download, i.e. after
    rank 0's end.  Run in a child process per setting: RPT_PIN_HOST is read once per process (1: the caller's buffer is
    page-locked for the call; 0: pageable copies, the two-pass order alone must do it)."""
    import subprocess
    import sys
    code = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import conftest, oracle_lib
rpt = conftest.load_package()
os.environ["RPT_GATHER"] = "p2p"
n, w, h, spp = 4, 512, 384, 64
t = rpt.Tracer(rpt.AnalyticalScene(), devices=[0] * n, seed=1)
buf = rpt.ColorBuffer(w, h)
t.render_n(buf, 1)                                 # warm: module load, staging allocations
buf = rpt.ColorBuffer(w, h)
t.render_n(buf, spp)
ms = C.c_float()
gaps = []
for b in range(1, n):
    rpt._lib.check(rpt.lib().rpt_debug_render_overlap_ms(t._h, 0, b, C.byref(ms)), t._h)
    gaps.append(ms.value)
rpt._lib.check(rpt.lib().rpt_debug_render_overlap_ms(t._h, 0, 0, C.byref(ms)), t._h)
own = ms.value
o = oracle_lib.Oracle("liboracle.so")
want = o.render(o.scene_analytical(), w, h, spp, seed=1)
same = (buf.image().view(np.uint32) == want.view(np.uint32)).all()
import json
print("RESULT" + json.dumps([own, gaps, bool(same)]))
t.close()
""" % (os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = dict(os.environ, RPT_PIN_HOST=pin)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1]
    import json
    own, gaps, same = json.loads(line[len("RESULT"):])
    assert same, "fan-out changed the image"
    assert own > 0.0
    # every other rank began before rank 0 ended (a positive gap), by a good part of rank 0's own span
    assert all(g > 0.0 for g in gaps), "devices ran one after the other: begin(rank k) - end(rank 0) = %r ms (rank 0 took %.3f ms)" % (gaps, own)


def test_a_device_listed_twice_renders_on_two_streams(rpt, oracle, torch_cuda):
    """rpt_create_multi with a repeated device id (no environment knob): two ranks on one GPU, each with its own stream and its
    share of the rows, gathered with device copies instead of RCCL.  Progressive steps issued back to back without host
    synchronisation (the point of it: one rank's launch fills the tail of the other's) give the oracle's frame bit for bit."""
    os.environ.pop("RPT_GATHER", None)
    w, h = 200, 120
    t = rpt.Tracer(rpt.AnalyticalScene(), devices=[0, 0], seed=4)
    assert t.world() == (0, 2, 2)
    for spp in (3, 1, 2):
        t.render_resident(w, h, spp)
    got = t.resident_to_host(w, h).image()
    assert_bit_identical(got, oracle.render(oracle.scene_analytical(), w, h, 6, seed=4), "device listed twice")
    t.close()


def test_multi_context_large_scene_and_progressive_gathers(rpt, oracle, torch_cuda):
    """A scene with device tables (one copy per device of the context) on 3 virtual ranks, gathered after each of two
    steps: the second gather reuses the staging buffers of the first."""
    from rust_pathtracer_amd import scenes
    s = scenes.random_spheres_scene(n_spheres=300, n_lights=6)
    w, h = 64, 40
    os.environ["RPT_GATHER"] = "p2p"
    try:
        t = rpt.Tracer(s, devices=[0, 0, 0], seed=3)
    finally:
        os.environ.pop("RPT_GATHER", None)
    desc = s.describe()
    t.render_resident(w, h, 1)
    assert_bit_identical(t.resident_to_host(w, h).image(), oracle.render(desc, w, h, 1, seed=3), "large scene, step 1")
    t.render_resident(w, h, 2)
    assert_bit_identical(t.resident_to_host(w, h).image(), oracle.render(desc, w, h, 3, seed=3), "large scene, step 2")
    t.close()


@pytest.mark.parametrize("tile_rows", [2, 8])
def test_full_size_config3_whole_frame_through_eight_ranks(rpt, oracle, torch_cuda, tile_rows):
    """BASELINE.json configs[2] WHOLE: AnalyticalScene 3840x2160 x 1024 spp through the 8-rank code path — rpt_create_multi with
    device 0 listed eight times (cyclic row blocks: 2 rows, the library's default, and the 8 rows bench.py asks for), every rank's
    strided launch, the gather to rank 0 and the scatter at full size.  The gathered image must be the single-context frame bit for
    bit, and one complete row out of EVERY rank's tile the oracle's (global rows follow tracer.rs:29-37: the pixel's index in the
    whole frame keys its stream, whichever rank renders it)."""
    from rust_pathtracer_amd import tiling
    torch = torch_cuda
    w, h, spp, world = 3840, 2160, 1024, 8
    os.environ.pop("RPT_GATHER", None)
    t = rpt.Tracer(rpt.AnalyticalScene(), devices=[0] * world, seed=1)
    assert t.world() == (0, world, world)
    job = tiling.TiledRender(t, w, h, tile_rows=tile_rows)
    job.render_n(spp)
    image = job.gather()
    assert t.resident_frames() == spp
    single = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
    buf = rpt.DeviceColorBuffer(w, h)
    single.render_n(buf, spp)
    torch.cuda.synchronize()
    assert torch.equal(image.view(torch.int32), buf.pixels.view(torch.int32)), "the 8-rank frame is not the one-context frame"
    got = image.cpu().numpy()
    assert np.all(got[..., 3] == 1.0)
    # one row per rank, spread over sky, spheres and floor: block b of `tile_rows` rows belongs to rank b % 8
    picked, owned = {}, 0
    for rank in range(world):
        rows = tiling.tile_global_rows(h, tile_rows, rank, world)
        assert abs(len(rows) - h // world) <= tile_rows          # (270 blocks of 8 rows over 8 ranks: 272 or 264 rows each)
        owned += len(rows)
        g = rows[(37 * (rank + 1)) % len(rows)]
        assert (g // tile_rows) % world == rank
        picked[rank] = g
    assert owned == h
    for rank, g in picked.items():
        px = np.zeros((h, w, 4), dtype=np.float32)
        oracle.render(oracle.scene_analytical(), w, h, spp, seed=1, pixels=px, rows=(g, g + 1))
        assert_bit_identical(got[g], px[g], "configs[2] whole frame, rank %d's global row %d (blocks of %d rows)" % (rank, g, tile_rows))
    single.close()
    t.close()


def test_bench_fallback_gather_gives_the_same_image(rpt, oracle, torch_cuda):
    """bench.py's insurance path (per-rank tile + torch.distributed RCCL gather + the library's scatter), on a one-rank
    group: the image of the plain render, bit for bit."""
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import bench
    from rust_pathtracer_amd import tiling
    import socket
    with socket.socket() as sk:                                       # any free port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        w, h = 70, 37
        t = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
        job = bench.TorchGatherRender(t, tiling, w, h, 2, 0, 1, 0)
        job.render_n(2)
        job.render_n(3)
        img = job.gather()
        assert job.kernel_ms() > 0.0
        assert_bit_identical(img.cpu().numpy(), oracle.render(oracle.scene_analytical(), w, h, 5, seed=1), "fallback gather")
        t.close()
    finally:
        dist.destroy_process_group()


def test_a_gather_in_flight_shows_the_frame_it_was_asked_for(rpt, oracle, torch_cuda):
    """The gather runs beside the renders that follow it (a snapshot of every tile, sent on a second stream): asked for after
    step 1 and waited for after step 2 has been enqueued, it must deliver step 1's frame; the next one step 2's."""
    torch = torch_cuda
    from rust_pathtracer_amd import tiling
    s = rpt.AnalyticalScene()
    w, h = 160, 96
    os.environ["RPT_GATHER"] = "p2p"
    try:
        t = rpt.Tracer(s, devices=[0, 0, 0], seed=2)
    finally:
        os.environ.pop("RPT_GATHER", None)
    job = tiling.TiledRender(t, w, h, tile_rows=2)
    images = []
    for step in range(3):
        job.render_n(2)
        job.gather_begin()
        job.render_n(1)                                             # enqueued behind the snapshot, beside the exchange
        images.append(job.gather_end().cpu().numpy().copy())
    desc = s.describe()
    for step, img in enumerate(images):
        assert_bit_identical(img, oracle.render(desc, w, h, 3 * step + 2, seed=2), "gather %d" % step)
    assert_bit_identical(t.resident_to_host(w, h).image(), oracle.render(desc, w, h, 9, seed=2), "download after the last step")
    t.close()


def test_bench_runs_the_multi_gpu_control_flow_on_one_gpu(rpt, torch_cuda):
    """bench.py --gpus 2 as the driver launches it, with both ranks on this box's one GPU (--smoke-shared-gpu: virtual ranks, peer
    copies) and small frames: the JSON line must carry configs[2]'s strong-scaling step, the fixed-work-per-GPU leg, the one-GPU
    time of the same frame and the configs[4] leg."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--small",
                        "--smoke-shared-gpu"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["width"] == 3840 // 8 and d["roofline"]["launches_per_step"] == 1
    for key in ("weak_scaling", "strong_scaling", "configs4"):
        assert key in d and (d[key].get("value", 1) > 0), key
    assert "10k spheres" in d["configs4"]["workload"]


def test_bench_keeps_its_line_when_a_secondary_leg_hangs(rpt, torch_cuda):
    """The legs bench.py runs after the headline (weak scaling, the one-GPU frame, configs[4]) share the communicator with it; a rank
    that stalls there must cost those legs, not the line: a timer emits it — whole, under the lock the main thread prints under — with
    what there is, and the job ends with a NON-ZERO status: a hung rank must not look like success to the launcher."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["RPT_BENCH_SECONDARY_LIMIT_S"] = "8"
    env["RPT_BENCH_TEST_STALL"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--small",
                        "--smoke-shared-gpu"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode != 0, "a cut-off run must not report success"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "ONE JSON line:\n" + r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["secondary_legs"].startswith("cut off")


def test_bench_multi_gpu_path_with_the_library_communicator_and_one_rank(rpt, torch_cuda):
    """bench.py's N > 1 path for real — tiling.rank_tracer (unique id over the job's gloo group, ncclCommInitRank), the first exchange,
    configs[2]'s steps with the gather beside the next render, the secondary legs, the configs[4] leg — with the one rank a one-GPU box
    can give it (--force-multi): everything but a second peer.  The line must name the library's gather, not the fallback."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = "29533"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--small", "--force-multi"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "ONE JSON line"
    d = json.loads(lines[0])
    assert d["value"] > 0 and d["config"]["gather"].startswith("library RCCL"), d["config"]
    for key in ("weak_scaling", "strong_scaling", "configs4"):
        assert key in d and d[key].get("value", d[key].get("speedup", 1)) > 0, (key, d.get("secondary_legs"))
    assert "secondary_legs" not in d
