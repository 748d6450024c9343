// k_util.hip — what runs beside the render kernels: the dispatch order of the next launch, the scatter of gathered rank tiles, the
// ColorBuffer's u8 conversions (buffer.rs:37-89).  HBM-bound, one pass each.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rpt.h"
#include "dev_math.h"
#include "launch.h"

using namespace rptdev;

// Scatter rank-major gathered tiles into the full image (one float4 per thread).
__global__ __launch_bounds__(256) void untile_kernel(const float4* __restrict__ gathered, float4* __restrict__ image,
                                                     uint32_t width, uint32_t height, uint32_t tile_rows, uint32_t world,
                                                     uint32_t rows_padded)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = (uint64_t)width * height;
    if (idx >= total) return;
    const uint32_t grow = (uint32_t)(idx / width);
    const uint32_t col = (uint32_t)(idx % width);
    const uint32_t gb = grow / tile_rows;
    const uint32_t rank = gb % world;
    const uint32_t lrow = (gb / world) * tile_rows + (grow % tile_rows);
    image[idx] = gathered[((uint64_t)rank * rows_padded + lrow) * width + col];
}

// The dispatch order of a context's next launch from the costs its last one left (block_tile above): a counting sort of the
// tiles by cost, descending, in one workgroup.  cost[t * 4 + w]: cycles / 64 for which wave w of tile t held its slot (0: the
// wave had no pixel); a tile's cost is the sum over its waves — the slot time it takes.  Ties keep no particular order.
constexpr uint32_t kOrderBuckets = 1024;
// A tile's cost: the LONGEST time one of its waves held its slot.  (The sum over the waves — the slot time the tile takes — is the
// wrong key: a tile on a silhouette, one expensive wave and three of sky, ends as late as a tile of four expensive waves.
// configs[1]: by the sum 11.1, bottom rows first 11.4, by the maximum 11.7 Gsamples/s.)
RPT_DEV uint32_t tile_key(uint4 c)
{
    const uint32_t a = c.x > c.y ? c.x : c.y, b = c.z > c.w ? c.z : c.w;
    return a > b ? a : b;
}
__global__ __launch_bounds__(1024) void sched_order_kernel(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t n_tiles)
{
    __shared__ uint32_t s_bucket[kOrderBuckets];
    __shared__ uint32_t s_scan[kOrderBuckets];
    __shared__ uint32_t s_max;
    const uint32_t tid = threadIdx.x;
    s_bucket[tid] = 0u;
    if (tid == 0u) s_max = 0u;
    __syncthreads();
    const uint4* cost4 = reinterpret_cast<const uint4*>(cost);
    uint32_t m = 0u;
    for (uint32_t t = tid; t < n_tiles; t += 1024u) {
        const uint4 c = cost4[t];
        const uint32_t sum = tile_key(c);
        m = sum > m ? sum : m;
    }
    atomicMax(&s_max, m);
    __syncthreads();
    const uint32_t mx = s_max;
    if (mx == 0u) return;                                           // nothing was recorded: the order stays what it is
    const float scale = (float)(kOrderBuckets - 1u) / (float)mx;
    const auto bucket_of = [&](uint32_t t) {
        const uint4 c = cost4[t];
        const uint32_t sum = tile_key(c);
        uint32_t b = (uint32_t)((float)sum * scale);
        b = b > kOrderBuckets - 1u ? kOrderBuckets - 1u : b;
        return kOrderBuckets - 1u - b;                              // most expensive first
    };
    for (uint32_t t = tid; t < n_tiles; t += 1024u) atomicAdd(&s_bucket[bucket_of(t)], 1u);
    __syncthreads();
    // exclusive prefix sum over the buckets (Hillis-Steele on 1 024 entries, one per thread)
    uint32_t v = s_bucket[tid];
    const uint32_t own = v;
    s_scan[tid] = v;
    __syncthreads();
    for (uint32_t off = 1u; off < kOrderBuckets; off <<= 1) {
        const uint32_t add = tid >= off ? s_scan[tid - off] : 0u;
        __syncthreads();
        v += add;
        s_scan[tid] = v;
        __syncthreads();
    }
    s_bucket[tid] = v - own;                                        // where this bucket's tiles start
    __syncthreads();
    for (uint32_t t = tid; t < n_tiles; t += 1024u) order[atomicAdd(&s_bucket[bucket_of(t)], 1u)] = t;
}

// the order before anything is known: bottom rows first; and no costs yet
__global__ __launch_bounds__(256) void sched_init_kernel(uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t n_tiles)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_tiles) return;
    order[i] = n_tiles - 1u - i;
    reinterpret_cast<uint4*>(cost)[i] = make_uint4(0u, 0u, 0u, 0u);
}

// Rust `as u8`: saturating, NaN -> 0, truncation toward zero.
RPT_DEV uint32_t as_u8(float x)
{
    if (!(x == x)) return 0u;
    if (x <= 0.0f) return 0u;
    if (x >= 255.0f) return 255u;
    return (uint32_t)x;
}

// ColorBuffer::convert_to_u8, buffer.rs:55-64
__global__ __launch_bounds__(256) void convert_to_u8_kernel(const float4* __restrict__ pixels, uint32_t* __restrict__ out, uint64_t n)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float4 p = pixels[idx];
    const uint32_t r = as_u8(rpt_powf(p.x, 0.4545f) * 255.0f);
    const uint32_t g = as_u8(rpt_powf(p.y, 0.4545f) * 255.0f);
    const uint32_t b = as_u8(rpt_powf(p.z, 0.4545f) * 255.0f);
    const uint32_t a = as_u8(p.w * 255.0f);
    out[idx] = r | (g << 8) | (b << 16) | (a << 24);
}

// ColorBuffer::convert_to_u8_at, buffer.rs:67-89 (one thread per destination pixel)
__global__ __launch_bounds__(256) void convert_to_u8_at_kernel(const float4* __restrict__ pixels, uint32_t bw, uint32_t bh,
                                                             uint32_t* __restrict__ frame, uint32_t at0, uint32_t at1,
                                                             uint32_t width, uint32_t height)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint64_t)width * height) return;
    const uint32_t row = (uint32_t)(idx / width);
    const uint64_t x = idx % width;
    const uint64_t y = (uint64_t)row + 1u;                            // y = height - j with j = height - 1 - row
    if (x > at0 && x < (uint64_t)at0 + bw && y > at1 && y < (uint64_t)at1 + bh) {
        const float4 p = pixels[(y - at1) * bw + (x - at0)];
        frame[idx] = as_u8(p.x * 255.0f) | (as_u8(p.y * 255.0f) << 8) | (as_u8(p.z * 255.0f) << 16) | (as_u8(p.w * 255.0f) << 24);
    }
}

namespace rptlaunch {

hipError_t sched_init(uint32_t* cost, uint32_t* order, uint32_t n_tiles, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(sched_init_kernel, dim3((n_tiles + 255u) / 256u), dim3(256), 0, st, cost, order, n_tiles);
    return hipGetLastError();
}

hipError_t sched_order(const uint32_t* cost, uint32_t* order, uint32_t n_tiles, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(sched_order_kernel, dim3(1), dim3(1024), 0, st, cost, order, n_tiles);
    return hipGetLastError();
}

hipError_t untile(const float* gathered, float* image, uint32_t width, uint32_t height, uint32_t tile_rows, uint32_t world,
                  uint32_t rows_padded, hipStream_t st)
{
    const uint64_t total = (uint64_t)width * height;
    (void)hipGetLastError();
    hipLaunchKernelGGL(untile_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, (const float4*)gathered, (float4*)image,
                       width, height, tile_rows, world, rows_padded);
    return hipGetLastError();
}

hipError_t convert_to_u8(const float* pixels, uint8_t* out, uint64_t n_pixels, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(convert_to_u8_kernel, dim3((uint32_t)((n_pixels + 255) / 256)), dim3(256), 0, st, (const float4*)pixels, (uint32_t*)out, n_pixels);
    return hipGetLastError();
}

hipError_t convert_to_u8_at(const float* pixels, uint32_t bw, uint32_t bh, uint8_t* frame, uint32_t at0, uint32_t at1, uint32_t width,
                            uint32_t height, hipStream_t st)
{
    const uint64_t n = (uint64_t)width * height;
    (void)hipGetLastError();
    hipLaunchKernelGGL(convert_to_u8_at_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (const float4*)pixels, bw, bh,
                       (uint32_t*)frame, at0, at1, width, height);
    return hipGetLastError();
}

}  // namespace rptlaunch
