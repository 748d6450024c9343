// tile_plan.h — the arithmetic of the cyclic row-block tiling (multi-GPU): which rows a rank owns, where they sit in its compact
// tile, and the strided copy between a top-down host image and that tile.  Plain C++ with no HIP type in it: the kernels and
// capi.hip include it through launch.h, tests/host_harness.cpp compiles it alone with g++ -fsanitize=address,undefined.
#pragma once

#include <stdint.h>
#include <string.h>

#include "../../include/rpt.h"

#if defined(__HIPCC__)
#define RPT_TILE_HD __host__ __device__
#else
#define RPT_TILE_HD
#endif

namespace rptdev {

// block b of `tile_rows` rows -> rank b % world
RPT_TILE_HD inline uint32_t tile_global_row(uint32_t local_row, uint32_t tile_rows, uint32_t rank, uint32_t world)
{
    uint32_t lb = local_row / tile_rows;
    return (lb * world + rank) * tile_rows + (local_row % tile_rows);
}

inline uint32_t tile_row_count(uint32_t height, uint32_t tile_rows, uint32_t rank, uint32_t world)
{
    if (tile_rows == 0 || world == 0 || rank >= world) return 0;
    uint32_t nblocks = (height + tile_rows - 1) / tile_rows;        // last block may be short
    uint32_t rows = 0;
    for (uint32_t b = rank; b < nblocks; b += world) {
        uint32_t start = b * tile_rows;
        uint32_t n = (start + tile_rows <= height) ? tile_rows : (height - start);
        rows += n;
    }
    return rows;
}

// rows of the largest tile: what every rank's tile is padded to, so that the gather's counts are equal
inline uint32_t tile_rows_padded(uint32_t height, uint32_t tile_rows, uint32_t world)
{
    uint32_t m = 0;
    for (uint32_t r = 0; r < world; ++r) { const uint32_t n = tile_row_count(height, tile_rows, r, world); m = n > m ? n : m; }
    return m;
}

// include/rpt.h, rpt_tile_copy_plan
inline int tile_copy_plan(uint32_t height, uint32_t tile_rows, uint32_t rank, uint32_t world, rpt_tile_plan* out)
{
    if (!out || tile_rows == 0 || world == 0 || rank >= world || height == 0) return RPT_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    if (world == 1) {                                                // everything is one block
        out->full_blocks = 1; out->block_rows = height; out->host_row0 = 0; out->host_row_stride = height;
        return RPT_OK;
    }
    const uint32_t nblocks = (height + tile_rows - 1u) / tile_rows;
    if (rank >= nblocks) return RPT_OK;                              // this rank owns no row
    const uint32_t nb = (nblocks - rank + world - 1u) / world;       // blocks of this rank: rank, rank + world, ...
    const uint32_t last_b = rank + (nb - 1u) * world;
    const bool ragged = (last_b == nblocks - 1u) && (height % tile_rows != 0u);
    out->full_blocks = ragged ? nb - 1u : nb;
    out->block_rows = tile_rows;
    out->host_row0 = rank * tile_rows;
    out->host_row_stride = world * tile_rows;
    if (ragged) {
        out->ragged_rows = height - last_b * tile_rows;
        out->ragged_host_row0 = last_b * tile_rows;
        out->ragged_tile_row0 = (nb - 1u) * tile_rows;
    }
    return RPT_OK;
}

}  // namespace rptdev
