"""Row tiling of the image across the GPUs of a node, and the gather of the tiles.

The path shards per pixel (tracer.rs:33 touches only its own pixel), so ranks exchange
nothing while rendering.  Rows are dealt cyclically in blocks of `tile_rows` rows
(block b -> rank b % world): contiguous slabs would give the sky rows to some GPUs and
the floor/sphere rows (3-4x more work per sample) to others.  Each rank accumulates
its rows in a compact tile; one gather (RCCL over xGMI, torch.distributed backend
"nccl") brings the tiles to every rank when the image is needed.
"""
import ctypes as C

import torch

from . import _lib


def tile_row_count(height, tile_rows, rank, world):
    return _lib.lib().rpt_tile_row_count(height, tile_rows, rank, world)


def tile_global_rows(height, tile_rows, rank, world):
    """Global row index of every local row of `rank` (host-side helper)."""
    n = tile_row_count(height, tile_rows, rank, world)
    f = _lib.lib().rpt_tile_global_row
    return [f(i, tile_rows, rank, world) for i in range(n)]


def padded_rows(height, tile_rows, world):
    return max(tile_row_count(height, tile_rows, r, world) for r in range(world))


def untile(gathered, width, height, tile_rows, world, tracer=None):
    """gathered: [world, rows_padded, width, 4] (rank-major, as all_gather returns) ->
    [height, width, 4].  On CUDA tensors this is the HIP scatter kernel; on CPU tensors
    (gloo tests) it is an index permutation."""
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


def all_gather_tiles(tile, world, group=None):
    """All-gather equal-size tiles -> [world, rows_padded, width, 4] (rank-major).  The output is
    allocated in the concatenated form, which both the nccl (RCCL) and gloo backends accept."""
    import torch.distributed as dist
    out = torch.empty((world * tile.shape[0],) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    dist.all_gather_into_tensor(out, tile, group=group)
    return out.view((world,) + tuple(tile.shape))


class TiledRender:
    """Progressive render of one image over the ranks of a torch.distributed group."""

    def __init__(self, tracer, width, height, tile_rows=2, group=None, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.tracer = tracer
        self.width, self.height, self.tile_rows = width, height, tile_rows
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rows = tile_row_count(height, tile_rows, self.rank, self.world)
        self.rows_padded = padded_rows(height, tile_rows, self.world)
        self.device = device if device is not None else torch.device("cuda", tracer.device)
        self.tile = torch.zeros(self.rows_padded, width, 4, dtype=torch.float32, device=self.device)
        self.frames = 0
        self._snapshot = None       # gather_begin: the tile as it was when the gather was requested
        self._gathered = None       # all-gather destination, [world * rows_padded, width, 4]

    def render_n(self, spp):
        self.tracer.render_tile(self.tile, self.width, self.height, self.frames, spp, self.tile_rows, self.rank, self.world)
        self.frames += spp

    def gather(self):
        """All ranks receive the full image (all-gather of equal-size tiles + scatter)."""
        if self.world == 1:
            return self.tile[: self.height].clone() if self.rows == self.height else untile(
                self.tile.unsqueeze(0), self.width, self.height, self.tile_rows, 1, self.tracer)
        return self.gather_end(self.gather_begin())

    def gather_begin(self):
        """Start the all-gather of the tile as it is NOW and return a handle for gather_end().  The collective runs on
        the process group's own stream (RCCL over xGMI), so render_n() calls issued before gather_end() overlap with it;
        the tile is snapshotted first (one device-to-device copy) because the next render_n() updates it in place.
        At most one gather is in flight: call gather_end() before the next gather_begin()."""
        assert self.world > 1
        if self._snapshot is None:
            self._snapshot = torch.empty_like(self.tile)
            self._gathered = torch.empty((self.world * self.rows_padded, self.width, 4), dtype=torch.float32, device=self.device)
        self._snapshot.copy_(self.tile)
        return self.dist.all_gather_into_tensor(self._gathered, self._snapshot, group=self.group, async_op=True)

    def gather_end(self, work):
        """Wait for gather_begin()'s collective (a stream-level wait on device tensors) and scatter the tiles into the image."""
        work.wait()
        out = self._gathered.view(self.world, self.rows_padded, self.width, 4)
        return untile(out, self.width, self.height, self.tile_rows, self.world, self.tracer)
