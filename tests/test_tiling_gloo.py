"""The N > 1 path on CPU, world size 2 over gloo.  There is no GPU here, so the kernel is replaced by the CPU oracle
rendering exactly the rows a rank owns; what is under test is everything around it that the multi-process job relies on:
the library's tile arithmetic (which rows a rank owns, where they sit in its compact tile, the padded tile size), the
distribution of rank 0's communicator id over the job's process group, the gather of equal-size tiles to rank 0 and the
scatter into the top-down image, and progressive accumulation across steps (frames carried between render calls)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, w, h, tile_rows, steps, out_dir):
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401
    import oracle_lib
    from rust_pathtracer_amd import tiling
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the id hand-off of tiling.rank_tracer, with a stand-in for the library's rpt_comm_unique_id (no RCCL without a GPU)
        box = [bytes(range(128)) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        assert box[0] == bytes(range(128))

        o = oracle_lib.Oracle()
        desc = o.scene_analytical()
        rows = tiling.tile_global_rows(h, tile_rows, rank, world)
        padded = tiling.padded_rows(h, tile_rows, world)
        assert len(rows) <= padded
        tile = np.zeros((padded, w, 4), dtype=np.float32)
        full_scratch = np.zeros((h, w, 4), dtype=np.float32)
        frames = 0
        for k, spp in enumerate(steps):                      # progressive: each step continues the rank's running means
            for lr, g in enumerate(rows):
                full_scratch[g] = tile[lr]
                o.render(desc, w, h, spp, seed=1, frames_done=frames, pixels=full_scratch, rows=(g, g + 1), threads=1)
                tile[lr] = full_scratch[g]
            frames += spp
            gathered = tiling.gather_tiles(torch.from_numpy(tile), world, dst=0)
            if rank == 0:
                img = tiling.untile(gathered, w, h, tile_rows, world)
                np.save(os.path.join(out_dir, "step%d.npy" % k), img.numpy())
            else:
                assert gathered is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tile_rows,h", [(2, 27), (5, 27), (2, 8), (16, 9)])
def test_two_rank_tiled_render_equals_single_process(tmp_path, oracle, tile_rows, h):
    w, world, steps = 40, 2, (2, 1)
    port = 29500 + (os.getpid() % 2000) + tile_rows + h
    mp.spawn(_worker, args=(world, port, w, h, tile_rows, steps, str(tmp_path)), nprocs=world, join=True)
    total = 0
    for k, spp in enumerate(steps):
        total += spp
        want = oracle.render(oracle.scene_analytical(), w, h, total, seed=1)
        got = np.load(os.path.join(str(tmp_path), "step%d.npy" % k))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "step %d" % k


def _failing_worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    if rank == 1:
        os.environ["RPT_RCCL_LIB"] = "/no-such-rccl/librccl.so.1"     # this rank cannot even load RCCL
    import conftest  # noqa: F401
    import rust_pathtracer_amd as rpt
    from rust_pathtracer_amd import tiling
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        why = ""
        try:
            tiling.rank_tracer(rpt.AnalyticalScene(), 0, seed=1)
        except RuntimeError as e:
            why = str(e)
        # bench.py's next step: the ranks agree on the fallback (must not hang: every rank is here)
        ok = torch.tensor([0 if why else 1])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        open(os.path.join(out_dir, "rank%d.txt" % rank), "w").write("%d|%s" % (int(ok.item()), why))
    finally:
        dist.destroy_process_group()


def test_rank_tracer_fails_on_every_rank_together(tmp_path):
    """tiling.rank_tracer when the library's communicator cannot be set up (here: no GPU on either rank, and rank 1 cannot load
    RCCL at all): every rank raises, none is left waiting in a broadcast or inside ncclCommInitRank, and the job's next
    collective (bench.py's agreement on the fallback) completes."""
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (on a GPU box rank_tracer succeeds)")
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_failing_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, why = open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read().split("|", 1)
        assert ok == "0" and "cannot be set up" in why, (r, ok, why)


def _protocol_worker(rank, world, port, out_dir, failing):
    sys.path.insert(0, HERE)
    import conftest  # noqa: F401
    import rust_pathtracer_amd as rpt
    from rust_pathtracer_amd import tiling
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def probe():
        if rank in failing:
            raise OSError("this rank's device is gone")

    try:
        kind, why = "", ""
        try:
            tiling.rank_tracer(rpt.AnalyticalScene(), 0, seed=1, probe=probe)
        except Exception as e:              # noqa: BLE001 - which exception, on which rank, is what the test looks at
            kind, why = type(e).__name__, str(e)
        ok = torch.tensor([0 if why else 1])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)            # the job's next collective (bench.py's agreement on the fallback)
        open(os.path.join(out_dir, "rank%d.txt" % rank), "w").write("%d|%s|%s" % (int(ok.item()), kind, why))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("failing", [(2,), (0,), (1, 3), ()])
def test_rank_tracer_agreement_protocol_with_four_ranks(tmp_path, failing):
    """tiling.rank_tracer with world = 4 and mixed outcomes of the per-rank pre-check (a stand-in for "RCCL loads and the device
    opens"): whichever ranks fail — a middle one, rank 0 (which would have made the unique id), two at once — EVERY rank raises the same
    error, none is left in the broadcast or inside ncclCommInitRank, and the job's next collective completes.  With no failing rank
    the protocol goes on to the library's communicator, which on a box without a GPU fails on every rank alike (no device)."""
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (on a GPU box four ranks would need four devices)")
    world = 4
    port = 33500 + (os.getpid() % 2000) + 7 * len(failing) + sum(failing)
    mp.spawn(_protocol_worker, args=(world, port, str(tmp_path), tuple(failing)), nprocs=world, join=True)
    got = [open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read().split("|", 2) for r in range(world)]
    assert all(g[0] == "0" for g in got), got                # the next collective ran on every rank, and says "fallback"
    assert len({g[1] for g in got}) == 1, got                # the same exception type everywhere
    if failing:
        for g in got:
            assert g[1] == "RuntimeError" and "cannot be set up" in g[2], g
            assert any("rank %d:" % f in g[2] for f in failing) or "another rank" in g[2], g
    else:
        assert all(g[2] for g in got), got                   # no device here: every rank failed, together, after the agreement


def test_tile_copy_plan_covers_exactly_the_ranks_rows(rpt):
    """rpt_tile_copy_plan (what rpt_render / rpt_resident_upload follow for their one strided copy per device) against
    the row-by-row definition rpt_tile_global_row, for every rank of many image / block / world sizes."""
    import ctypes as C
    lib = rpt.lib()
    for height in (1, 2, 7, 9, 16, 27, 54, 600, 1080, 2160):
        for tile_rows in (1, 2, 3, 5, 8, 16, 4000):
            for world in (1, 2, 3, 4, 8):
                seen = {}
                for rank in range(world):
                    plan = rpt._abi.rpt_tile_plan()
                    assert lib.rpt_tile_copy_plan(height, tile_rows, rank, world, C.byref(plan)) == 0
                    pairs = []                                   # (host row, tile row) the plan copies
                    for b in range(plan.full_blocks):
                        for r in range(plan.block_rows):
                            pairs.append((plan.host_row0 + b * plan.host_row_stride + r, b * plan.block_rows + r))
                    for r in range(plan.ragged_rows):
                        pairs.append((plan.ragged_host_row0 + r, plan.ragged_tile_row0 + r))
                    n = lib.rpt_tile_row_count(height, tile_rows, rank, world)
                    want = [(lib.rpt_tile_global_row(i, tile_rows if world > 1 else height, rank, world), i) for i in range(n)]
                    assert sorted(pairs, key=lambda p: p[1]) == want, (height, tile_rows, world, rank)
                    assert n <= lib.rpt_tile_rows_padded(height, tile_rows, world)
                    for g, _ in pairs:
                        assert g not in seen and 0 <= g < height
                        seen[g] = rank
                assert len(seen) == height
    bad = rpt._abi.rpt_tile_plan()
    assert lib.rpt_tile_copy_plan(10, 0, 0, 1, C.byref(bad)) == rpt._abi.RPT_ERR_INVALID_ARG
    assert lib.rpt_tile_copy_plan(10, 2, 3, 3, C.byref(bad)) == rpt._abi.RPT_ERR_INVALID_ARG
