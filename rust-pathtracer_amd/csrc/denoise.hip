// denoise.hip — the denoiser of include/rpt.h ("denoiser": project-defined, the reference only lists one as a Todo,
// Readme.md:14): an edge-avoiding a-trous filter over the colour buffer, in a compressed colour space.  A separate HBM pass
// over a ColorBuffer, not part of the render path.  Specification and parity oracle: oracle/rpt_oracle.hpp, denoise().
//
// Roofline: HBM — algorithmically every iteration reads and writes the buffer once (16 B + 16 B per pixel).  Round 5: the first
// iterations (steps 1, 2, 4: up to three) are FUSED — one kernel stages a 32 x 32 tile plus a halo of 1 + 2 + 4 pixels in LDS, runs
// the iterations there on shrinking regions (46^2 loaded -> 44^2 -> 40^2 -> 32^2 written) and writes once: 2.07 x 16 B read + 16 B
// written per pixel for three iterations where three passes moved 1.66 x 96 B (profiles/r4/denoise_4k: halos re-fetched per pass, the
// ping-pong buffers never in cache); the price is 1.48 x the tap arithmetic (halo pixels of the earlier iterations are computed by
// every tile that needs them — the same operations on the same values: bit-identical).  Later iterations (steps 8 ...: the halo
// would be as large as the nine taps) read their taps through L1 / L2.  -DRPT_DENOISE_UNFUSED: one pass per iteration (A/B).
#include <hip/hip_runtime.h>

#include "dev_math.h"
#include "launch.h"

using namespace rptdev;

namespace {

struct DnSum {
    v3 acc;
    float wsum;
};

// one tap of the filter (rpt.h: d2, the NaN skip, the edge-stopping weight)
RPT_DEV void dn_tap(DnSum& s, v3 cp, v3 cq, float hw, float k)
{
    const float d0 = cp.x - cq.x, d1 = cp.y - cq.y, d2c = cp.z - cq.z;
    const float d2 = __builtin_fmaf(d0, d0, __builtin_fmaf(d1, d1, d2c * d2c));     // (explicit fmas: the specification's, rpt.h)
    if (!(d2 == d2)) return;
    const float t = __builtin_fmaf(-d2, k, 1.0f);
    const float g = t > 0.0f ? t : 0.0f;
    const float wt = hw * (g * g);
    s.acc.x = __builtin_fmaf(cq.x, wt, s.acc.x);
    s.acc.y = __builtin_fmaf(cq.y, wt, s.acc.y);
    s.acc.z = __builtin_fmaf(cq.z, wt, s.acc.z);
    s.wsum = s.wsum + wt;
}

RPT_DEV float dn_h(int d) { return d == 0 ? 0.5f : 0.25f; }

RPT_DEV float4 dn_finish(const DnSum& s, v3 cp, bool last, float4 orig)
{
    const bool ok = s.wsum > 0.0f;
    const v3 m = divs3(s.acc, s.wsum);                               // (dev_math.h: three quotients, one reciprocal — the IEEE quotients)
    const v3 o = mk3(ok ? m.x : cp.x, ok ? m.y : cp.y, ok ? m.z : cp.z);
    if (!last) return make_float4(o.x, o.y, o.z, 0.0f);
    const float inf = __builtin_inff();
    const bool finite = (__builtin_fabsf(orig.x) < inf) && (__builtin_fabsf(orig.y) < inf) && (__builtin_fabsf(orig.z) < inf);
    if (!finite) return orig;
    return make_float4(fdiv(o.x, 1.0f - o.x), fdiv(o.y, 1.0f - o.y), fdiv(o.z, 1.0f - o.z), orig.w);
}

// Steps 1, 2 and 4 through LDS: the workgroup's 16 x 16 pixels plus a halo of STEP — (16 + 2 STEP)^2 loads for 256 pixels (1.3, 1.6,
// 2.3 per pixel) instead of nine taps each from L1 / L2, which is what bounded the first version (2.5 TB/s of HBM-equivalent at both
// 1080p and 4K: the taps' L2 traffic, not the arithmetic).  FIRST: the source is the caller's buffer and c' = c / (1 + c) is computed
// once per LOADED pixel (three divides), not once per tap.
#ifndef RPT_DENOISE_TILE
#define RPT_DENOISE_TILE 16         // pixels per side of a workgroup's tile (one thread per pixel): 16 -> 256 threads, 32 -> 1 024
#endif
constexpr int kDnTile = RPT_DENOISE_TILE;
constexpr uint32_t kDnThreads = (uint32_t)(kDnTile * kDnTile);
constexpr uint32_t kDnShift = kDnTile == 32 ? 5u : 4u;

template <int STEP, bool FIRST>
__global__ __launch_bounds__(kDnThreads) void denoise_tile_kernel(const float4* __restrict__ src, const float4* __restrict__ orig_in, float4* __restrict__ out,
                                                           uint32_t w, uint32_t h, float k, uint32_t last)
{
    constexpr int T = kDnTile + 2 * STEP;
    __shared__ float s_c[3][T * T];
    const int x0 = (int)blockIdx.x * kDnTile, y0 = (int)blockIdx.y * kDnTile;
    for (uint32_t e = threadIdx.x; e < (uint32_t)(T * T); e += kDnThreads) {
        const int lx = (int)(e % (uint32_t)T), ly = (int)(e / (uint32_t)T);
        const int gx = x0 + lx - STEP, gy = y0 + ly - STEP;
        float c0 = __builtin_nanf(""), c1 = c0, c2 = c0;              // outside the image: NaN, i.e. a tap that is skipped
        if (gx >= 0 && gy >= 0 && gx < (int)w && gy < (int)h) {
            const float4 v = src[(size_t)gy * w + (size_t)gx];
            if (FIRST) { c0 = fdiv(v.x, 1.0f + v.x); c1 = fdiv(v.y, 1.0f + v.y); c2 = fdiv(v.z, 1.0f + v.z); }
            else { c0 = v.x; c1 = v.y; c2 = v.z; }
        }
        s_c[0][e] = c0; s_c[1][e] = c1; s_c[2][e] = c2;
    }
    __syncthreads();
    const uint32_t tx = threadIdx.x & (uint32_t)(kDnTile - 1), ty = threadIdx.x >> kDnShift;
    const uint32_t x = (uint32_t)x0 + tx, y = (uint32_t)y0 + ty;
    if (x >= w || y >= h) return;
    const int ce = (int)((ty + (uint32_t)STEP) * (uint32_t)T + tx + (uint32_t)STEP);
    const v3 cp = mk3(s_c[0][ce], s_c[1][ce], s_c[2][ce]);
    DnSum s{mk3(0.0f, 0.0f, 0.0f), 0.0f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int e = ce + dy * STEP * T + dx * STEP;
            dn_tap(s, cp, mk3(s_c[0][e], s_c[1][e], s_c[2][e]), dn_h(dy) * dn_h(dx), k);
        }
    const size_t p = (size_t)y * w + x;
    float4 orig = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (last) orig = orig_in[p];
    out[p] = dn_finish(s, cp, last != 0u, orig);
}

// The first K iterations (K = 1, 2, 3; steps 1, 2, 4) in one kernel.  Stage 0 is the loaded region (colours compressed once per loaded
// pixel), stage j the output of iteration j - 1 on the region later iterations still need; positions outside the image hold NaN at every
// stage — a tap there is skipped, exactly as a tap outside the image is in the one-pass kernels.
constexpr int kFuseTile = 32;
template <int K> struct FuseGeom {
    static constexpr int kHalo = (1 << K) - 1;                       // 1, 3, 7
    static constexpr int side(int stage) { return kFuseTile + 2 * (kHalo - ((1 << stage) - 1)); }      // stage 0: tile + 2 halo ... stage K: tile
    // Two buffers, used in turn: stage j lives in buffer j & 1 (stage j + 1 is written while stage j is read; stage j - 1 is dead by then).
    static constexpr int cells(int buf) { return K > buf ? side(buf) * side(buf) : 0; }     // (the largest stage of a buffer is its first)
};
struct DnCell { float x, y, z; };                                    // a compressed colour in LDS: one ds_read2_b32 + one ds_read_b32 per tap

template <int K>
__global__ __launch_bounds__(1024) void denoise_fused_kernel(const float4* __restrict__ src, float4* __restrict__ out, uint32_t w, uint32_t h, float k0, uint32_t last)
{
    typedef FuseGeom<K> G;
    __shared__ DnCell s_a[G::cells(0)];                              // K = 3: 46^2 and 44^2 cells = 48.6 KB: three workgroups per CU
    __shared__ DnCell s_b[G::cells(1) > 0 ? G::cells(1) : 1];
    const int x0 = (int)blockIdx.x * kFuseTile, y0 = (int)blockIdx.y * kFuseTile;
    const float nan = __builtin_nanf("");
    {   // stage 0: load + compress
        constexpr int T = G::side(0), R = G::kHalo;
        for (uint32_t e = threadIdx.x; e < (uint32_t)(T * T); e += 1024u) {
            const int lx = (int)(e % (uint32_t)T), ly = (int)(e / (uint32_t)T);
            const int gx = x0 + lx - R, gy = y0 + ly - R;
            DnCell c{nan, nan, nan};                                 // outside the image: NaN, i.e. a tap that is skipped
            if (gx >= 0 && gy >= 0 && gx < (int)w && gy < (int)h) {
                const float4 v = src[(size_t)gy * w + (size_t)gx];
                c = DnCell{fdiv(v.x, 1.0f + v.x), fdiv(v.y, 1.0f + v.y), fdiv(v.z, 1.0f + v.z)};
            }
            s_a[e] = c;
        }
    }
    __syncthreads();
    float k = k0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const DnCell* from = (j & 1) ? s_b : s_a;
        DnCell* to = (j & 1) ? s_a : s_b;
        const int Ts = G::side(j), Td = G::side(j + 1), step = 1 << j;       // source stage j -> destination stage j + 1 (the tile itself when j + 1 == K)
        const int off = (Ts - Td) / 2;                                       // = step
        const int Rd = G::kHalo - ((1 << (j + 1)) - 1);                      // halo of the destination region
        const bool to_global = j + 1 == K;
        for (uint32_t e = threadIdx.x; e < (uint32_t)(Td * Td); e += 1024u) {
            const int lx = (int)(e % (uint32_t)Td), ly = (int)(e / (uint32_t)Td);
            const int gx = x0 + lx - Rd, gy = y0 + ly - Rd;
            const bool inside = gx >= 0 && gy >= 0 && gx < (int)w && gy < (int)h;
            const int ce = (ly + off) * Ts + lx + off;
            DnCell o{nan, nan, nan};
            float4 res = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (inside) {
                const DnCell cc = from[ce];
                const v3 cp = mk3(cc.x, cc.y, cc.z);
                DnSum s{mk3(0.0f, 0.0f, 0.0f), 0.0f};
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const DnCell q = from[ce + dy * step * Ts + dx * step];
                        dn_tap(s, cp, mk3(q.x, q.y, q.z), dn_h(dy) * dn_h(dx), k);
                    }
                const bool fin = to_global && last != 0u;
                float4 orig = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (fin) orig = src[(size_t)gy * w + (size_t)gx];
                res = dn_finish(s, cp, fin, orig);
                o = DnCell{res.x, res.y, res.z};
            }
            if (to_global) {
                if (inside) out[(size_t)gy * w + (size_t)gx] = res;
            } else {
                to[e] = o;
            }
        }
        k = k * 4.0f;
        if (!to_global) __syncthreads();
    }
}

// iterations 1.. (step 2^i): taps straight from the compressed buffer
__global__ __launch_bounds__(256) void denoise_step_kernel(const float4* __restrict__ cur, const float4* __restrict__ orig_in,
                                                           float4* __restrict__ out, uint32_t w, uint32_t h, int step, float k, uint32_t last)
{
    const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    const uint32_t x = blockIdx.x * 16u + tx, y = blockIdx.y * 16u + ty;
    if (x >= w || y >= h) return;
    const size_t p = (size_t)y * w + x;
    const float4 c4 = cur[p];
    const v3 cp = mk3(c4.x, c4.y, c4.z);
    DnSum s{mk3(0.0f, 0.0f, 0.0f), 0.0f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const long long qx = (long long)x + (long long)step * dx, qy = (long long)y + (long long)step * dy;
            if (qx < 0 || qy < 0 || qx >= (long long)w || qy >= (long long)h) continue;
            const float4 q = (dx == 0 && dy == 0) ? c4 : cur[(size_t)qy * w + (size_t)qx];
            dn_tap(s, cp, mk3(q.x, q.y, q.z), dn_h(dy) * dn_h(dx), k);
        }
    float4 orig = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (last) orig = orig_in[p];
    out[p] = dn_finish(s, cp, last != 0u, orig);
}

}  // namespace

namespace rptlaunch {

// `iterations` passes: in -> t0 -> t1 -> ... -> out, alternating between `out` and `scratch` so that the last one lands in `out`.
hipError_t denoise(const float* in, float* out, float* scratch, uint32_t width, uint32_t height, uint32_t iterations, float edge_k,
                   hipStream_t st)
{
    (void)hipGetLastError();
    const dim3 grid((width + 15u) / 16u, (height + 15u) / 16u), wg(256);
    const dim3 tgrid((width + (uint32_t)kDnTile - 1u) / (uint32_t)kDnTile, (height + (uint32_t)kDnTile - 1u) / (uint32_t)kDnTile), twg(kDnThreads);
    float k = edge_k;
    const float4* cur = nullptr;
    uint32_t first = 0;
#ifndef RPT_DENOISE_UNFUSED
    {   // iterations 0 .. nf-1 in one kernel (denoise_fused_kernel)
        const uint32_t nf = iterations < 3u ? iterations : 3u;
        const bool last = nf == iterations;
        float4* dst = (float4*)(((iterations - nf) & 1u) ? scratch : out);
        const dim3 fgrid((width + (uint32_t)kFuseTile - 1u) / (uint32_t)kFuseTile, (height + (uint32_t)kFuseTile - 1u) / (uint32_t)kFuseTile), fwg(1024);
        const uint32_t l = last ? 1u : 0u;
        if (nf == 1u) hipLaunchKernelGGL((denoise_fused_kernel<1>), fgrid, fwg, 0, st, (const float4*)in, dst, width, height, k, l);
        else if (nf == 2u) hipLaunchKernelGGL((denoise_fused_kernel<2>), fgrid, fwg, 0, st, (const float4*)in, dst, width, height, k, l);
        else hipLaunchKernelGGL((denoise_fused_kernel<3>), fgrid, fwg, 0, st, (const float4*)in, dst, width, height, k, l);
        cur = dst;
        for (uint32_t i = 0; i < nf; ++i) k = k * 4.0f;
        first = nf;
    }
#endif
    for (uint32_t i = first; i < iterations; ++i) {
        const bool last = i + 1u == iterations;
        float4* dst = (float4*)(((iterations - 1u - i) & 1u) ? scratch : out);
        const float4* orig = (const float4*)in;
        const uint32_t l = last ? 1u : 0u;
        if (i == 0) hipLaunchKernelGGL((denoise_tile_kernel<1, true>), tgrid, twg, 0, st, orig, orig, dst, width, height, k, l);
        else if (i == 1) hipLaunchKernelGGL((denoise_tile_kernel<2, false>), tgrid, twg, 0, st, cur, orig, dst, width, height, k, l);
        else if (i == 2) hipLaunchKernelGGL((denoise_tile_kernel<4, false>), tgrid, twg, 0, st, cur, orig, dst, width, height, k, l);
        else hipLaunchKernelGGL(denoise_step_kernel, grid, wg, 0, st, cur, orig, dst, width, height, 1 << i, k, l);
        cur = dst;
        k = k * 4.0f;
    }
    return hipGetLastError();
}

}  // namespace rptlaunch
