"""The RCCL leg of the N > 1 path on the one GPU a test box has: a single-rank `nccl` process group runs the same
calls bench.py and TiledRender make (barrier, all_gather_into_tensor of the tile, all_reduce MAX of the step time)
on device tensors next to the render kernels, and the gathered + untiled image must be the plain render."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_gather_matches_plain_render(rpt):
    import torch.distributed as dist
    from rust_pathtracer_amd import tiling
    w, h, spp = 200, 90, 3
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29700 + os.getpid() % 200)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        tracer = rpt.Tracer(rpt.AnalyticalScene(), device=0, seed=1)
        job = tiling.TiledRender(tracer, w, h, tile_rows=2)
        assert (job.rank, job.world, job.rows) == (0, 1, h)
        job.render_n(spp)
        dist.barrier()
        gathered = tiling.all_gather_tiles(job.tile, 1)            # RCCL all-gather on the device tile
        img = tiling.untile(gathered, w, h, 2, 1, tracer)
        # the pipelined form bench.py uses: begin, render the next step into the tile, end -> the image of the step it
        # was begun for
        job._snapshot = torch.empty_like(job.tile)
        job._gathered = torch.empty((job.rows_padded, w, 4), dtype=torch.float32, device=job.device)
        job._snapshot.copy_(job.tile)
        work = dist.all_gather_into_tensor(job._gathered, job._snapshot, async_op=True)
        job.render_n(2)                                            # overlaps with the collective, updates the tile in place
        img_async = job.gather_end(work)
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        assert float(t.item()) == 1.5
        buf = rpt.DeviceColorBuffer(w, h)
        tracer.render_n(buf, spp)
        torch.cuda.synchronize()
        a, b = img.cpu().numpy(), buf.pixels.cpu().numpy()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert np.array_equal(img_async.cpu().numpy().view(np.uint32), b.view(np.uint32))
        tracer.close()
    finally:
        dist.destroy_process_group()
