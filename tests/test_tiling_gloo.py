"""The N > 1 path on CPU: two gloo ranks deal the image rows cyclically, each fills its compact tile
(here with the CPU oracle standing in for the GPU kernel — this test is about the tiling, the
all-gather and the untile step, not about rendering), all-gather, untile, and every rank must hold
the single-process image bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, w, h, tile_rows, spp, out_dir):
    sys.path.insert(0, HERE)
    import conftest
    import oracle_lib
    from rust_pathtracer_amd import tiling
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = oracle_lib.Oracle()
        desc = o.scene_analytical()
        rows = tiling.tile_global_rows(h, tile_rows, rank, world)
        padded = tiling.padded_rows(h, tile_rows, world)
        tile = np.zeros((padded, w, 4), dtype=np.float32)
        full_scratch = np.zeros((h, w, 4), dtype=np.float32)
        for lr, g in enumerate(rows):                       # this rank renders only the rows it owns
            o.render(desc, w, h, spp, seed=1, pixels=full_scratch, rows=(g, g + 1), threads=1)
            tile[lr] = full_scratch[g]
        t = torch.from_numpy(tile)
        img = tiling.all_gather_tiles(t, world)
        img = tiling.untile(img, w, h, tile_rows, world)
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), img.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tile_rows", [2, 5])
def test_two_rank_tiled_render_equals_single_process(tmp_path, oracle, tile_rows):
    w, h, spp, world = 40, 27, 2, 2
    port = 29500 + (os.getpid() % 2000) + tile_rows
    mp.spawn(_worker, args=(world, port, w, h, tile_rows, spp, str(tmp_path)), nprocs=world, join=True)
    want = oracle.render(oracle.scene_analytical(), w, h, spp, seed=1)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "rank %d" % r


def _pipelined_worker(rank, world, port, w, h, tile_rows, out_dir):
    """bench.py's N > 1 loop: the gather of step k is begun, step k+1 updates the tile in place, then the gather is
    ended — it must deliver step k's image (the tile is snapshotted), and the next gather step k+1's."""
    sys.path.insert(0, HERE)
    import conftest
    import oracle_lib
    from rust_pathtracer_amd import tiling
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = oracle_lib.Oracle()
        desc = o.scene_analytical()
        job = tiling.TiledRender(None, w, h, tile_rows=tile_rows, device=torch.device("cpu"))
        rows = tiling.tile_global_rows(h, tile_rows, rank, world)
        full = np.zeros((h, w, 4), dtype=np.float32)

        def render_step(frames_done, spp):                  # the oracle stands in for the kernel, rows of this rank only
            for lr, g in enumerate(rows):
                o.render(desc, w, h, spp, seed=1, frames_done=frames_done, pixels=full, rows=(g, g + 1), threads=1)
                job.tile[lr] = torch.from_numpy(full[g])

        render_step(0, 2)
        pending = job.gather_begin()
        render_step(2, 1)                                   # updates job.tile in place while the gather is in flight
        img1 = job.gather_end(pending).clone()
        img2 = job.gather()
        np.save(os.path.join(out_dir, "p%d_1.npy" % rank), img1.numpy())
        np.save(os.path.join(out_dir, "p%d_2.npy" % rank), img2.numpy())
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_delivers_the_step_it_was_begun_for(tmp_path, oracle):
    w, h, world, tile_rows = 40, 27, 2, 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_pipelined_worker, args=(world, port, w, h, tile_rows, str(tmp_path)), nprocs=world, join=True)
    want1 = oracle.render(oracle.scene_analytical(), w, h, 2, seed=1)
    want2 = oracle.render(oracle.scene_analytical(), w, h, 3, seed=1)
    for r in range(world):
        got1 = np.load(os.path.join(str(tmp_path), "p%d_1.npy" % r))
        got2 = np.load(os.path.join(str(tmp_path), "p%d_2.npy" % r))
        assert np.array_equal(got1.view(np.uint32), want1.view(np.uint32)), "rank %d step 1" % r
        assert np.array_equal(got2.view(np.uint32), want2.view(np.uint32)), "rank %d step 2" % r
