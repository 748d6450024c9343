"""Row tiling of the image across the GPUs of a node — a thin caller of the library.

The tiling, the per-rank tiles, the RCCL gather to rank 0 and the scatter into the image all live behind the C ABI
(include/rpt.h: rpt_create_rank / rpt_create_multi, rpt_resident_render, rpt_resident_gather_device).  This module only
  * distributes rank 0's RCCL unique id over whatever channel the job already has (torch.distributed, any backend), and
  * keeps index helpers and a CPU-tensor untile for the world-size-2 gloo tests (which have no GPU).
"""
import ctypes as C

import torch

from . import _lib
from .api import Tracer, comm_unique_id


def tile_row_count(height, tile_rows, rank, world):
    return _lib.lib().rpt_tile_row_count(height, tile_rows, rank, world)


def tile_global_rows(height, tile_rows, rank, world):
    """Global row index of every local row of `rank` (host-side helper)."""
    n = tile_row_count(height, tile_rows, rank, world)
    f = _lib.lib().rpt_tile_global_row
    return [f(i, tile_rows, rank, world) for i in range(n)]


def padded_rows(height, tile_rows, world):
    return _lib.lib().rpt_tile_rows_padded(height, tile_rows, world)


def untile(gathered, width, height, tile_rows, world, tracer=None):
    """gathered: [world, rows_padded, width, 4] (rank-major) -> [height, width, 4].  On CUDA tensors this is the
    library's scatter kernel; on CPU tensors (gloo tests) an index permutation."""
    rows_padded = gathered.shape[1]
    if gathered.is_cuda:
        assert tracer is not None
        image = torch.empty(height, width, 4, dtype=torch.float32, device=gathered.device)
        stream = torch.cuda.current_stream(gathered.device).cuda_stream
        _lib.check(_lib.lib().rpt_untile_device(tracer._h, gathered.data_ptr(), image.data_ptr(), width, height,
                                                tile_rows, world, rows_padded, C.c_void_p(stream)), tracer._h)
        return image
    src_rank = torch.empty(height, dtype=torch.long)
    src_row = torch.empty(height, dtype=torch.long)
    for r in range(world):
        for lr, g in enumerate(tile_global_rows(height, tile_rows, r, world)):
            src_rank[g] = r
            src_row[g] = lr
    return gathered[src_rank, src_row]


def gather_tiles(tile, world, group=None, dst=0):
    """CPU-tensor stand-in of the library's gather (gloo tests): equal-size tiles -> [world, rows_padded, width, 4] on
    rank `dst`, None elsewhere."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    out = [torch.empty_like(tile) for _ in range(world)] if rank == dst else None
    dist.gather(tile, out, dst=dst, group=group)
    return torch.stack(out) if rank == dst else None


def rank_tracer(scene, local_device, seed=1, group=None, probe=None):
    """One process per GPU: build this rank's Tracer.  Rank 0 asks the library for an RCCL unique id, the job's
    torch.distributed group (gloo or nccl, it only carries 128 bytes) hands it to everybody, and every rank joins
    the library's own communicator (collective).  `probe`: tests only — a callable that stands for "can this rank load RCCL and
    open its device" (raises if not), so that the agreement protocol can be driven with mixed outcomes on a box without a GPU.

    Every rank makes the same sequence of group calls whatever fails where: what can fail on one rank alone (the
    library or RCCL not loading, no usable device) is tried first and agreed on with an all_reduce, and rank 0's id
    travels together with its error, so either every rank reaches ncclCommInitRank or every rank raises."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    why = ""
    try:
        if probe is not None:
            probe()
        else:
            comm_unique_id()                              # RCCL loads in this process (the id itself is thrown away)
            Tracer(scene, device=local_device, seed=seed).close()   # the device is there and is a gfx950
    except Exception as e:                                # noqa: BLE001 - reported on every rank below
        why = "rank %d: %s: %s" % (rank, type(e).__name__, e)
    ok = torch.tensor([0 if why else 1])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    box = [None, why]
    if int(ok.item()) == 1 and rank == 0:
        try:
            box = [comm_unique_id(), ""]
        except Exception as e:                            # noqa: BLE001
            box = [None, "rank 0: %s: %s" % (type(e).__name__, e)]
    dist.broadcast_object_list(box, src=0, group=group)
    if int(ok.item()) == 0 or box[0] is None:
        raise RuntimeError("rank_tracer: the library's communicator cannot be set up (%s)" % (why or box[1] or "a failure on another rank"))
    return Tracer(scene, device=local_device, seed=seed, rank=rank, world=world, unique_id=box[0])


class TiledRender:
    """Progressive render of one image over the ranks of `tracer` (a rank tracer, a multi-device tracer or a plain one):
    render_n() accumulates into the per-rank tiles in HBM, gather() assembles the image on rank 0's device."""

    def __init__(self, tracer, width, height, tile_rows=2):
        self.tracer = tracer
        self.width, self.height, self.tile_rows = width, height, tile_rows
        self.rank, self.world, self.n_local = tracer.world()
        tracer.set_tile_rows(tile_rows)
        self.image = None

    @property
    def frames(self):
        return self.tracer.resident_frames()

    def render_n(self, spp):
        self.tracer.render_resident(self.width, self.height, spp)

    def gather_begin(self):
        """Enqueue gather + scatter behind the renders (no host wait); the image is valid after gather_end()."""
        if self.rank == 0 and self.image is None:
            self.image = torch.empty(self.height, self.width, 4, dtype=torch.float32, device=torch.device("cuda", self.tracer.device))
        self.tracer.resident_gather(self.image if self.rank == 0 else None)

    def gather_end(self):
        self.tracer.resident_sync()
        return self.image

    def gather(self):
        self.gather_begin()
        return self.gather_end()
