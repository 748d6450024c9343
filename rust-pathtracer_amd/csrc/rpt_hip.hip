// rpt_hip.hip — HIP kernels (gfx950) and the C ABI of include/rpt.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see
// build.py).  There is NO CPU fallback in this library: without a HIP device every
// entry point that computes returns RPT_ERR_NO_DEVICE / RPT_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rpt.h"
#include "dev_integrator.h"
#include "dev_scene_large.h"

using namespace rptdev;

// ---------------------------------------------------------------------------
// cyclic row-block tiling (multi-GPU): block b of `tile_rows` rows -> rank b % world
// ---------------------------------------------------------------------------
__host__ __device__ static inline uint32_t tile_global_row(uint32_t local_row, uint32_t tile_rows, uint32_t rank, uint32_t world)
{
    uint32_t lb = local_row / tile_rows;
    return (lb * world + rank) * tile_rows + (local_row % tile_rows);
}

static uint32_t tile_row_count(uint32_t height, uint32_t tile_rows, uint32_t rank, uint32_t world)
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

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------

// Per-pixel setup shared by both render kernels: tracer.rs:34-46.
struct PixelSetup {
    bool valid;
    uint32_t pixel_index;
    float px, py;                     // coord of tracer.rs:46
    size_t pix_offset;                // index of this pixel's float4 in the tile buffer
};

RPT_DEV PixelSetup pixel_setup(const RenderParams& rp)
{
    // A wave covers an 8x8 pixel block (coherent paths), a 256-thread workgroup 16x16.
    PixelSetup ps;
    // Bottom rows are dispatched first: in the usual outdoor framing they are the expensive
    // ones (floor / objects), so the cheap sky tiles fill the tail of the launch (+3 %).
    const uint32_t tile = gridDim.x - 1u - blockIdx.x;
    const uint32_t tx = tile % rp.tiles_x;
    const uint32_t ty = tile / rp.tiles_x;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t col = tx * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t lrow = ty * 16u + (wave >> 1) * 8u + (lane >> 3);
    ps.valid = (col < rp.width) && (lrow < rp.rows_local);
    const uint32_t grow = tile_global_row(lrow, rp.tile_rows, rp.rank, rp.world);
    // j counts rows from the bottom (par_rchunks, tracer.rs:29-37)
    const float W = (float)rp.width;
    const float H = (float)rp.height;
    const uint32_t j = rp.height - 1u - grow;
    const float x = (float)col;
    const float y = H - (float)j;
    const float xx = x / W;
    const float yy = y / H;
    ps.px = xx;
    ps.py = 1.0f - yy;
    ps.pixel_index = grow * rp.width + col;
    ps.pix_offset = (size_t)lrow * rp.width + col;
    return ps;
}

// mix_color, tracer.rs:108-113, with color = [r, g, b, 1.0] (tracer.rs:59,105)
RPT_DEV void blend(float4& acc, v3 rad, float v)
{
    acc.x = (1.0f - v) * acc.x + rad.x * v;
    acc.y = (1.0f - v) * acc.y + rad.y * v;
    acc.z = (1.0f - v) * acc.z + rad.z * v;
    acc.w = (1.0f - v) * acc.w + 1.0f * v;
}

// Megakernel, one thread per pixel, `spp` samples per launch, nested-loop form
// (sample loop outside, bounce loop inside; lanes whose path ended idle until the
// wave's longest path ends).  Kept as the A/B baseline for the regenerating kernel.
// The running mean of tracer.rs:105-117 is carried in registers across the launch's
// samples and updated with the reference's own expression once per sample, so one
// launch of S samples is bit-identical to S reference render() calls; the framebuffer
// is read and written once per launch as float4 (16 B per lane, 128 B per 8-pixel row).
template <class S>
RPT_DEV void render_nested_body(const S& sc, const RenderParams& rp)
{
    const PixelSetup ps = pixel_setup(rp);
    if (!ps.valid) return;
    float4* pix = reinterpret_cast<float4*>(rp.pixels) + ps.pix_offset;
    float4 acc = *pix;
    for (uint32_t s = 0; s < rp.spp; ++s) {
        const uint64_t frames = rp.frames_done + s;
        const uint32_t fkey = frame_key_hd(rp.seed, frames);
        const float v = 1.0f / (float)(frames + 1);                 // tracer.rs:115
        const v3 rad = trace_sample(sc, ps.px, ps.py, fkey, ps.pixel_index);
        blend(acc, rad, v);
    }
    *pix = acc;
}

__global__ __launch_bounds__(256) void render_small_nested_kernel(const SceneSmall sc, const RenderParams rp) { render_nested_body(sc, rp); }
__global__ __launch_bounds__(256) void render_large_nested_kernel(const SceneLarge sc, const RenderParams rp) { render_nested_body(sc, rp); }
__global__ __launch_bounds__(256) void render_sdf_nested_kernel(const SceneSmallSdf sc, const RenderParams rp) { render_nested_body(sc, rp); }

// The production megakernel.  Same arithmetic per sample, different schedule:
//  * each lane runs its pixel's whole sample loop as a state machine (dev_integrator.h,
//    PathRegs); when its path ends it blends the sample into its running mean and starts
//    the next camera path at once (path regeneration);
//  * a bounce is split into TRACE (closest hit + miss/emitter exits, cheap) and SHADE
//    (next-event estimation + Disney BSDF sampling, ~3x the instructions).  A lane that
//    hits a surface parks its SurfaceHit in registers and waits; the wave runs SHADE only
//    when at least `shade_threshold` lanes are parked (wave ballot + popcount), or nobody
//    is left to trace.  The expensive block therefore executes with most lanes active,
//    while the cheap one absorbs the divergence.
// The per-sample frame key and blend weight 1/(frames+1) are per-lane values now (lanes
// drift apart in sample index), so the workgroup stages them once in LDS tables.
constexpr uint32_t kMaxSppPerLaunch = 512;

// Minimum waves per SIMD the register allocator must leave room for (2nd argument of
// __launch_bounds__ = waves per SIMD on gfx950); see DESIGN.md for the measurements.
#ifndef RPT_WAVES_PER_SIMD
#define RPT_WAVES_PER_SIMD 5
#endif

enum : uint32_t { ST_TRACE = 0u, ST_SHADE = 1u, ST_DONE = 2u };

template <class S>
RPT_DEV void render_regen_body(const S& sc, const RenderParams& rp)
{
    __shared__ uint32_t s_fkey[kMaxSppPerLaunch];
    __shared__ float s_weight[kMaxSppPerLaunch];
    for (uint32_t i = threadIdx.x; i < rp.spp; i += 256u) {
        const uint64_t frames = rp.frames_done + i;
        s_fkey[i] = frame_key_hd(rp.seed, frames);
        s_weight[i] = 1.0f / (float)(frames + 1);                   // tracer.rs:115
    }
    __syncthreads();

    // Cold per-lane state lives in LDS, not in VGPRs: the pixel's running mean and its
    // constants are touched only when a sample ends (once per ~2 bounces), and the seven
    // registers they would pin are what separates 4 from 5 resident waves per SIMD.
    __shared__ float4 s_acc[256];                                   // running mean, tracer.rs:105-117
    __shared__ float4 s_pix[256];                                   // {coord.x, coord.y, bits(pixel_index), -}
    __shared__ float4 s_hit[256];                                   // parked SurfaceHitCold {fhp, eta}
    const uint32_t tid = threadIdx.x;
    {
        const PixelSetup ps = pixel_setup(rp);
        if (!ps.valid) return;
        s_acc[tid] = *(reinterpret_cast<const float4*>(rp.pixels) + ps.pix_offset);
        s_pix[tid] = make_float4(ps.px, ps.py, rpt_u2f(ps.pixel_index), 0.0f);
    }

    if (sc.max_depth == 0) {                                        // no bounce loop at all: radiance is zero
        float4 acc = s_acc[tid];
        for (uint32_t s = 0; s < rp.spp; ++s) blend(acc, mk3(0.0f, 0.0f, 0.0f), s_weight[s]);
        *(reinterpret_cast<float4*>(rp.pixels) + pixel_setup(rp).pix_offset) = acc;
        return;
    }

    uint32_t s = 0;
    uint32_t state = ST_TRACE;
    PathRegs p;
    SurfaceHit sh;
    {
        const float4 c = s_pix[tid];
        path_begin(sc, p, c.x, c.y, s_fkey[0], rpt_f2u(c.z));
    }

    // blend the finished sample into the running mean and start the next one (or retire)
    auto finish_sample = [&]() {
        float4 acc = s_acc[tid];
        blend(acc, p.radiance, s_weight[s]);
        s_acc[tid] = acc;
        s += 1;
        if (s >= rp.spp) {
            state = ST_DONE;
        } else {
            const float4 c = s_pix[tid];
            path_begin(sc, p, c.x, c.y, s_fkey[s], rpt_f2u(c.z));
            state = ST_TRACE;
        }
    };

    for (;;) {
        if (state == ST_TRACE) {
            SurfaceHitCold shc;
            if (path_trace(sc, p, sh, shc)) {
                s_hit[tid] = make_float4(shc.fhp.x, shc.fhp.y, shc.fhp.z, shc.eta);
                state = ST_SHADE;
            } else {
                finish_sample();
            }
        }
        const uint64_t m_shade = __ballot(state == ST_SHADE);
        const uint64_t m_trace = __ballot(state == ST_TRACE);
        if ((m_shade | m_trace) == 0ull) break;
        if ((uint32_t)__popcll(m_shade) >= rp.shade_threshold || m_trace == 0ull) {
            if (state == ST_SHADE) {
                state = ST_TRACE;
                if (path_shade(sc, p, sh, &s_hit[tid])) finish_sample();
            }
        }
    }
    *(reinterpret_cast<float4*>(rp.pixels) + pixel_setup(rp).pix_offset) = s_acc[tid];
}

__global__ __launch_bounds__(256, RPT_WAVES_PER_SIMD) void render_small_regen_kernel(const SceneSmall sc, const RenderParams rp) { render_regen_body(sc, rp); }
// Large scenes: same schedule; the scene tables are streamed from HBM (dev_scene_large.h).
__global__ __launch_bounds__(256, RPT_WAVES_PER_SIMD) void render_large_regen_kernel(const SceneLarge sc, const RenderParams rp) { render_regen_body(sc, rp); }
// Small scenes with the procedural SDF object (sphere marching inside closest_hit / any_hit).
__global__ __launch_bounds__(256, RPT_WAVES_PER_SIMD) void render_sdf_regen_kernel(const SceneSmallSdf sc, const RenderParams rp) { render_regen_body(sc, rp); }

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

__global__ __launch_bounds__(256) void probe_math_kernel(uint32_t fn, const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.0f;
    switch (fn) {
    case RPT_PROBE_SIN: r = rpt_sinf(a[i]); break;
    case RPT_PROBE_COS: r = rpt_cosf(a[i]); break;
    case RPT_PROBE_LOG2: r = rpt_log2f(a[i]); break;
    case RPT_PROBE_POW: r = rpt_powf(a[i], b[i]); break;
    case RPT_PROBE_DIV: r = a[i] / b[i]; break;
    case RPT_PROBE_SQRT: r = __builtin_sqrtf(a[i]); break;
    case RPT_PROBE_RNG: {                                            // a = seed bits, b = frame bits, i = pixel; first draw
        Rng rng;
        rng.init(frame_key_hd((uint64_t)rpt_f2u(a[i]), (uint64_t)rpt_f2u(b[i])), (uint32_t)i);
        r = rng.gen();
        break;
    }
    default: break;
    }
    out[i] = r;
}

__global__ __launch_bounds__(256) void probe_rays_kernel(const SceneLarge sc, const float* __restrict__ rays, uint32_t* __restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + i * 7;
    RayD ray{mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5])};
    float dist = 3.40282347e+38f;
    uint32_t best = 0xFFFFFFFFu;
    bool hit = false;
    bool any;
    if (sc.use_grid) {
        grid_closest_sphere(sc, ray, dist, best, hit);
        any = grid_any_sphere(sc, ray, true, r[6]);
    } else {
        any = false;
        for (uint32_t k = 0; k < sc.n_spheres; ++k) {
            const float4 s = sphere_uniform(sc, k);
            float t;
            bool h = hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t);
            if (h && (k == 0 || t < dist)) { dist = t; best = k; hit = true; }
            any = any || (h && t < r[6]);
        }
    }
    out[i * 3 + 0] = rpt_f2u(dist);
    out[i * 3 + 1] = best;
    out[i * 3 + 2] = any ? 1u : 0u;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct rpt_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool has_scene = false;
    bool large = false;               // scene exceeds the kernarg tables: SceneLarge + device tables
    SceneSmallSdf scene;              // camera part is filled per launch (depends on width/height); sdf.n_prims == 0: plain
    SceneLarge scene_large;
    void* tables = nullptr;           // one device allocation holding the large scene's tables
    rpt_camera camera;
    float* fb = nullptr;              // device framebuffer for the host-pointer API
    size_t fb_bytes = 0;
    std::string err;
};

static thread_local std::string g_err;

static void set_err(rpt_ctx* ctx, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_err = buf;
}

#define RPT_HIP_CHECK(ctx, call)                                                                  \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            set_err(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RPT_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

// Frame-invariant part of Pinhole::gen_ray (camera/pinhole.rs:38-54), evaluated on the
// host with the reference's f32 operation order (this file is compiled with
// -ffp-contract=off) and the strict tan.
static DevCamera make_camera(const rpt_camera& c, float width, float height)
{
    struct h3 { float x, y, z; };
    auto sub = [](h3 a, h3 b) { return h3{a.x - b.x, a.y - b.y, a.z - b.z}; };
    auto mulf = [](h3 a, float f) { return h3{a.x * f, a.y * f, a.z * f}; };
    auto cross = [](h3 a, h3 b) { return h3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };

    const float ratio = width / height;
    const float half_width = rpt_tanf((c.fov_deg * (3.14159265358979323846f / 180.0f)) * 0.5f);   // f32::to_radians
    const float half_height = half_width / ratio;
    const h3 origin{c.origin[0], c.origin[1], c.origin[2]};
    const h3 center{c.center[0], c.center[1], c.center[2]};
    const h3 up{0.0f, 1.0f, 0.0f};
    h3 w = sub(origin, center);
    const float wl = __builtin_sqrtf(w.x * w.x + w.y * w.y + w.z * w.z);
    w = h3{w.x / wl, w.y / wl, w.z / wl};
    const h3 u = cross(up, w);
    const h3 v = cross(w, u);
    const h3 lower_left = sub(sub(sub(origin, mulf(u, half_width)), mulf(v, half_height)), w);
    const h3 horizontal = mulf(u, half_width * 2.0f);
    const h3 vertical = mulf(v, half_height * 2.0f);
    const h3 rd = sub(lower_left, origin);

    DevCamera d;
    d.ox = origin.x; d.oy = origin.y; d.oz = origin.z;
    d.rdx = rd.x; d.rdy = rd.y; d.rdz = rd.z;
    d.hx = horizontal.x; d.hy = horizontal.y; d.hz = horizontal.z;
    d.vx = vertical.x; d.vy = vertical.y; d.vz = vertical.z;
    d.psx = 1.0f / width;
    d.psy = 1.0f / height;
    return d;
}

// Uniform grid over the spheres of a large scene (dev_scene_large.h).  Cell size targets ~2 spheres
// per cell.  Every sphere is listed in each cell its PADDED bounding box overlaps; the padding is the
// distance outside the sphere at which the reference's f32 ray/sphere test (d2 = l.l - tca^2 <= r^2,
// absolute error ~4e-7 |l|^2) can still report a hit, for ray origins within `safe_r` of the grid
// centre, doubled for safety, plus 1e-3 cell sizes for the DDA's own rounding.
struct HostGrid {
    uint32_t n[3];
    float gmin[3], gmax[3], cs[3], inv_cs[3];
    float center[3], safe_r2;
    std::vector<uint32_t> cell_start, items;
};

static HostGrid build_grid(const rpt_sphere* sph, uint32_t count)
{
    HostGrid g;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i = 0; i < count; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], (double)sph[i].center[a] - sph[i].radius);
            hi[a] = std::max(hi[a], (double)sph[i].center[a] + sph[i].radius);
        }
    double ext[3], vol = 1.0;
    for (int a = 0; a < 3; ++a) {
        double pad = 1e-3 * (hi[a] - lo[a]) + 1e-3;
        lo[a] -= pad; hi[a] += pad;
        ext[a] = hi[a] - lo[a];
        vol *= ext[a];
    }
    const double target = std::cbrt(vol / (count / 2.0 + 1.0));       // cell edge for ~2 spheres per cell
    for (int a = 0; a < 3; ++a) {
        double n = std::ceil(ext[a] / target);
        n = n < 1 ? 1 : (n > 128 ? 128 : n);
        g.n[a] = (uint32_t)n;
        g.gmin[a] = (float)lo[a];
        g.cs[a] = (float)(ext[a] / n);
        g.inv_cs[a] = 1.0f / g.cs[a];
        g.gmax[a] = g.gmin[a] + (float)g.n[a] * g.cs[a];
    }
    const size_t ncell = (size_t)g.n[0] * g.n[1] * g.n[2];
    double half_diag = 0.0;
    for (int a = 0; a < 3; ++a) {
        g.center[a] = (float)(0.5 * (lo[a] + hi[a]));
        half_diag += 0.25 * ext[a] * ext[a];
    }
    half_diag = std::sqrt(half_diag);
    const double safe_r = 6.0 * half_diag;
    g.safe_r2 = (float)(safe_r * safe_r);
    const double max_l = safe_r + half_diag;                        // |sphere centre - ray origin| for usable rays
    const double d2_err = 1.2e-6 * max_l * max_l;                   // bound on the f32 error of l.l - tca*tca
    auto range = [&](const rpt_sphere& s, int a, int& c0, int& c1) {
        const double r = s.radius;
        const double pad = (std::sqrt(r * r + d2_err) - r) + 1e-3 * g.cs[a];
        c0 = (int)std::floor(((double)s.center[a] - s.radius - pad - g.gmin[a]) / g.cs[a]);
        c1 = (int)std::floor(((double)s.center[a] + s.radius + pad - g.gmin[a]) / g.cs[a]);
        c0 = std::max(0, std::min((int)g.n[a] - 1, c0));
        c1 = std::max(0, std::min((int)g.n[a] - 1, c1));
    };
    g.cell_start.assign(ncell + 1, 0);
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<uint32_t> cursor;
        if (pass == 1) {
            for (size_t c = 0; c < ncell; ++c) g.cell_start[c + 1] += g.cell_start[c];      // counts -> exclusive prefix sums
            cursor.assign(g.cell_start.begin(), g.cell_start.end() - 1);
            g.items.assign(g.cell_start[ncell], 0);
        }
        for (uint32_t i = 0; i < count; ++i) {                                               // ascending sphere index within a cell
            int x0, x1, y0, y1, z0, z1;
            range(sph[i], 0, x0, x1); range(sph[i], 1, y0, y1); range(sph[i], 2, z0, z1);
            for (int z = z0; z <= z1; ++z)
                for (int y = y0; y <= y1; ++y)
                    for (int x = x0; x <= x1; ++x) {
                        const size_t c = ((size_t)z * g.n[1] + y) * g.n[0] + x;
                        if (pass == 0) g.cell_start[c + 1] += 1;
                        else g.items[cursor[c]++] = i;
                    }
        }
    }
    return g;
}

// Lanes that must be parked on a surface hit before a wave runs its shading block (1..64).
// RPT_SHADE_THRESHOLD overrides the default for tuning runs.
static uint32_t shade_threshold()
{
    static const uint32_t v = [] {
        const char* e = getenv("RPT_SHADE_THRESHOLD");
        long t = e ? strtol(e, nullptr, 10) : 56;
        return (uint32_t)(t < 1 ? 1 : (t > 64 ? 64 : t));
    }();
    return v;
}

extern "C" {

uint32_t rpt_abi_version(void) { return RPT_ABI_VERSION; }

const char* rpt_last_error(const rpt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

// renderer/src/analytical.rs as data (see include/rpt.h)
int rpt_scene_analytical(rpt_scene_desc* out)
{
    if (!out) return RPT_ERR_INVALID_ARG;
    static rpt_sphere spheres[2];
    static rpt_plane planes[1];
    static rpt_light lights[1];
    static rpt_material mats[3];
    memset(spheres, 0, sizeof(spheres));
    memset(planes, 0, sizeof(planes));
    memset(lights, 0, sizeof(lights));
    memset(mats, 0, sizeof(mats));
    memset(out, 0, sizeof(*out));

    spheres[0] = rpt_sphere{{-1.1f, 0.0f, 0.0f}, 1.0f, 0};           // analytical.rs:41
    spheres[1] = rpt_sphere{{1.1f, 0.0f, 0.0f}, 1.0f, 1};            // analytical.rs:70
    planes[0] = rpt_plane{{0.0f, 1.0f, 0.0f}, {0.0f, -1.0f, 0.0f}, 0.0001f, 2, 0.0f};   // analytical.rs:194-198

    mats[0].mask = RPT_MAT_RGB | RPT_MAT_ROUGHNESS | RPT_MAT_METALLIC;              // analytical.rs:56-58
    mats[0].rgb[0] = mats[0].rgb[1] = mats[0].rgb[2] = 1.0f;
    mats[0].roughness = 0.05f;
    mats[0].metallic = 1.0f;
    mats[1].mask = RPT_MAT_RGB | RPT_MAT_CLEARCOAT | RPT_MAT_CLEARCOAT_GLOSS | RPT_MAT_ROUGHNESS;   // analytical.rs:82-85
    mats[1].rgb[0] = 1.0f; mats[1].rgb[1] = 0.186f; mats[1].rgb[2] = 0.0f;
    mats[1].clearcoat = 1.0f;
    mats[1].clearcoat_gloss = 1.0f;
    mats[1].roughness = 0.1f;
    mats[2].mask = RPT_MAT_ROUGHNESS;                                               // analytical.rs:107-116
    mats[2].roughness = 1.0f;
    mats[2].proc_kind = RPT_PROC_CHECKER_DIR;
    mats[2].proc_params[0] = 0.5f; mats[2].proc_params[1] = 100.0f;
    mats[2].proc_params[2] = 0.25f; mats[2].proc_params[3] = 0.1f;

    lights[0].type = RPT_LIGHT_SPHERICAL;                                           // analytical.rs:15-16
    lights[0].position[0] = 3.0f; lights[0].position[1] = 2.0f; lights[0].position[2] = 2.0f;
    lights[0].emission[0] = lights[0].emission[1] = lights[0].emission[2] = 3.0f;
    lights[0].radius = 1.0f;
    lights[0].area = 4.0f * 3.14159265358979323846f * lights[0].radius * lights[0].radius;   // light.rs:22

    out->abi_version = RPT_ABI_VERSION;
    out->flags = 0;
    out->camera.origin[2] = 3.0f;                                                   // pinhole.rs:16
    out->camera.fov_deg = 80.0f;                                                    // pinhole.rs:23
    out->background.kind = RPT_BG_GRADIENT_Y;                                       // analytical.rs:28-32
    out->background.colour_a[0] = out->background.colour_a[1] = out->background.colour_a[2] = 1.0f;
    out->background.colour_b[0] = 0.5f; out->background.colour_b[1] = 0.7f; out->background.colour_b[2] = 1.0f;
    out->background.gamma = 2.2f;
    out->background.scale = 0.5f;
    out->eps = 0.005f;                                                              // tracer.rs:16
    out->max_depth = 4;                                                             // scene.rs:29
    out->n_spheres = 2; out->spheres = spheres;
    out->n_planes = 1; out->planes = planes;
    out->n_lights = 1; out->lights = lights;
    out->n_materials = 3; out->materials = mats;
    return RPT_OK;
}

int rpt_create(rpt_ctx** out, int device_id)
{
    if (!out) { set_err(nullptr, "rpt_create: out is NULL"); return RPT_ERR_INVALID_ARG; }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_err(nullptr, "rpt_create: no HIP device (%s); this library has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return RPT_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= count) { set_err(nullptr, "rpt_create: device %d out of range [0,%d)", device_id, count); return RPT_ERR_INVALID_ARG; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { set_err(nullptr, "rpt_create: hipGetDeviceProperties failed"); return RPT_ERR_HIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(nullptr, "rpt_create: device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
        return RPT_ERR_NO_DEVICE;
    }
    rpt_ctx* ctx = new (std::nothrow) rpt_ctx();
    if (!ctx) return RPT_ERR_HIP;
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        set_err(nullptr, "rpt_create: cannot create a stream on device %d", device_id);
        delete ctx;
        return RPT_ERR_HIP;
    }
    *out = ctx;
    return RPT_OK;
}

void rpt_destroy(rpt_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->fb) (void)hipFree(ctx->fb);
    if (ctx->tables) (void)hipFree(ctx->tables);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rpt_upload_scene(rpt_ctx* ctx, const rpt_scene_desc* s)
{
    if (!ctx || !s) { set_err(ctx, "rpt_upload_scene: NULL argument"); return RPT_ERR_INVALID_ARG; }
    if (s->abi_version != RPT_ABI_VERSION) { set_err(ctx, "rpt_upload_scene: abi_version %u != %u", s->abi_version, RPT_ABI_VERSION); return RPT_ERR_INVALID_ARG; }
    if ((s->n_spheres && !s->spheres) || (s->n_planes && !s->planes) || (s->n_lights && !s->lights) || (s->n_materials && !s->materials)) {
        set_err(ctx, "rpt_upload_scene: a table pointer is NULL");
        return RPT_ERR_INVALID_ARG;
    }
    const bool large = s->n_spheres > (uint32_t)kMaxSpheres || s->n_lights > (uint32_t)kMaxLights || s->n_materials > (uint32_t)kMaxMaterials;
    if (s->n_planes > (uint32_t)kMaxPlanes) {
        set_err(ctx, "rpt_upload_scene: at most %d planes are supported", kMaxPlanes);
        return RPT_ERR_UNSUPPORTED;
    }
    for (uint32_t i = 0; i < s->n_spheres; ++i)
        if (s->spheres[i].material >= s->n_materials) { set_err(ctx, "rpt_upload_scene: sphere %u material out of range", i); return RPT_ERR_INVALID_ARG; }
    for (uint32_t i = 0; i < s->n_planes; ++i)
        if (s->planes[i].material >= s->n_materials) { set_err(ctx, "rpt_upload_scene: plane %u material out of range", i); return RPT_ERR_INVALID_ARG; }

    auto dev_plane = [](const rpt_plane& a) { return DevPlane{a.normal[0], a.normal[1], a.normal[2], a.point[0], a.point[1], a.point[2], a.min_denom, a.material, a.max_t}; };
    auto dev_light = [](const rpt_light& a) { return DevLight{a.type, a.position[0], a.position[1], a.position[2], a.emission[0], a.emission[1], a.emission[2], a.radius, a.area}; };
    auto dev_material = [](const rpt_material& a) {
        DevMaterial m;
        m.mask = a.mask; m.proc_kind = a.proc_kind;
        for (int k = 0; k < 3; ++k) { m.rgb[k] = a.rgb[k]; m.emission[k] = a.emission[k]; }
        m.anisotropic = a.anisotropic; m.metallic = a.metallic; m.roughness = a.roughness; m.subsurface = a.subsurface;
        m.specular_tint = a.specular_tint; m.sheen = a.sheen; m.sheen_tint = a.sheen_tint; m.clearcoat = a.clearcoat;
        m.clearcoat_gloss = a.clearcoat_gloss; m.spec_trans = a.spec_trans; m.ior = a.ior;
        for (int k = 0; k < 4; ++k) m.proc_params[k] = a.proc_params[k];
        return m;
    };
    auto dev_background = [](const rpt_background& b) {
        return DevBackground{b.kind, b.colour_a[0], b.colour_a[1], b.colour_a[2], b.colour_b[0], b.colour_b[1], b.colour_b[2], b.gamma, b.scale};
    };

    if (s->sdf.n_prims) {
        if (s->sdf.n_prims > (uint32_t)kMaxSdfPrims || !s->sdf.prims || s->sdf.material >= s->n_materials || !(s->sdf.smooth_k > 0.0f)) {
            set_err(ctx, "rpt_upload_scene: bad SDF object (1..%d prims, material in range, smooth_k > 0)", kMaxSdfPrims);
            return RPT_ERR_INVALID_ARG;
        }
        for (uint32_t i = 0; i < s->sdf.n_prims; ++i)
            if (s->sdf.prims[i].kind > RPT_SDF_TORUS_Y) { set_err(ctx, "rpt_upload_scene: unknown SDF primitive kind"); return RPT_ERR_INVALID_ARG; }
        if (large) { set_err(ctx, "rpt_upload_scene: the SDF object is only supported in small scenes"); return RPT_ERR_UNSUPPORTED; }
    }
    if (large) {
        // Layered patches need a bit per primitive; large scenes must use full sphere materials.
        for (uint32_t i = 0; i < s->n_spheres; ++i) {
            const rpt_material& m = s->materials[s->spheres[i].material];
            if ((m.mask & RPT_MAT_ALL) != RPT_MAT_ALL || m.proc_kind != RPT_PROC_NONE) {
                set_err(ctx, "rpt_upload_scene: scenes beyond %d spheres / %d lights / %d materials need full sphere materials "
                             "(mask == RPT_MAT_ALL, no procedural part); sphere %u does not", kMaxSpheres, kMaxLights, kMaxMaterials, i);
                return RPT_ERR_UNSUPPORTED;
            }
        }
        RPT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        const size_t sz_sph = sizeof(float4) * s->n_spheres;
        const size_t sz_smat = (sizeof(uint32_t) * s->n_spheres + 15) & ~(size_t)15;
        const size_t sz_lights = (sizeof(DevLight) * (s->n_lights ? s->n_lights : 1) + 15) & ~(size_t)15;
        const size_t sz_mats = sizeof(DevMaterial) * (s->n_materials ? s->n_materials : 1);
        const bool use_grid = s->n_spheres >= 64 && !getenv("RPT_NO_GRID");
        HostGrid grid;
        if (use_grid) grid = build_grid(s->spheres, s->n_spheres);
        const size_t sz_tables = (sz_sph + sz_smat + sz_lights + sz_mats + 15) & ~(size_t)15;
        const size_t sz_cstart = (sizeof(uint32_t) * grid.cell_start.size() + 15) & ~(size_t)15;
        const size_t sz_items = sizeof(uint32_t) * grid.items.size();
        std::vector<unsigned char> host(sz_tables + sz_cstart + sz_items, 0);
        float4* h_sph = reinterpret_cast<float4*>(host.data());
        uint32_t* h_smat = reinterpret_cast<uint32_t*>(host.data() + sz_sph);
        DevLight* h_lights = reinterpret_cast<DevLight*>(host.data() + sz_sph + sz_smat);
        DevMaterial* h_mats = reinterpret_cast<DevMaterial*>(host.data() + sz_sph + sz_smat + sz_lights);
        for (uint32_t i = 0; i < s->n_spheres; ++i) {
            h_sph[i] = make_float4(s->spheres[i].center[0], s->spheres[i].center[1], s->spheres[i].center[2], s->spheres[i].radius);
            h_smat[i] = s->spheres[i].material;
        }
        for (uint32_t i = 0; i < s->n_lights; ++i) h_lights[i] = dev_light(s->lights[i]);
        for (uint32_t i = 0; i < s->n_materials; ++i) h_mats[i] = dev_material(s->materials[i]);
        if (use_grid) {
            memcpy(host.data() + sz_tables, grid.cell_start.data(), sizeof(uint32_t) * grid.cell_start.size());
            memcpy(host.data() + sz_tables + sz_cstart, grid.items.data(), sz_items);
        }
        if (ctx->tables) { RPT_HIP_CHECK(ctx, hipFree(ctx->tables)); ctx->tables = nullptr; }
        RPT_HIP_CHECK(ctx, hipMalloc(&ctx->tables, host.size()));
        RPT_HIP_CHECK(ctx, hipMemcpy(ctx->tables, host.data(), host.size(), hipMemcpyHostToDevice));
        unsigned char* base = reinterpret_cast<unsigned char*>(ctx->tables);
        SceneLarge& L = ctx->scene_large;
        memset(&L, 0, sizeof(L));
        L.n_spheres = s->n_spheres; L.n_planes = s->n_planes; L.n_lights = s->n_lights; L.n_materials = s->n_materials;
        L.flags = s->flags; L.max_depth = s->max_depth; L.eps = s->eps; L.n_lights_f = (float)s->n_lights;
        L.bg = dev_background(s->background);
        L.spheres = reinterpret_cast<const float4*>(base);
        L.sphere_material = reinterpret_cast<const uint32_t*>(base + sz_sph);
        L.lights = reinterpret_cast<const DevLight*>(base + sz_sph + sz_smat);
        L.materials = reinterpret_cast<const DevMaterial*>(base + sz_sph + sz_smat + sz_lights);
        for (uint32_t i = 0; i < s->n_planes; ++i) L.planes[i] = dev_plane(s->planes[i]);
        L.use_grid = use_grid ? 1u : 0u;
        if (use_grid) {
            for (int a = 0; a < 3; ++a) {
                L.gn[a] = grid.n[a]; L.gmin[a] = grid.gmin[a]; L.gmax[a] = grid.gmax[a];
                L.cell_size[a] = grid.cs[a]; L.inv_cell_size[a] = grid.inv_cs[a];
                L.gcenter[a] = grid.center[a];
            }
            L.safe_r2 = grid.safe_r2;
            L.cell_start = reinterpret_cast<const uint32_t*>(base + sz_tables);
            L.cell_items = reinterpret_cast<const uint32_t*>(base + sz_tables + sz_cstart);
        }
        ctx->camera = s->camera;
        ctx->large = true;
        ctx->has_scene = true;
        return RPT_OK;
    }

    SceneSmallSdf& d = ctx->scene;
    memset(&d, 0, sizeof(d));
    d.n_spheres = s->n_spheres; d.n_planes = s->n_planes; d.n_lights = s->n_lights; d.n_materials = s->n_materials;
    d.flags = s->flags;
    d.max_depth = s->max_depth;
    d.eps = s->eps;
    d.n_lights_f = (float)s->n_lights;
    d.bg.kind = s->background.kind;
    d.bg.ax = s->background.colour_a[0]; d.bg.ay = s->background.colour_a[1]; d.bg.az = s->background.colour_a[2];
    d.bg.bx = s->background.colour_b[0]; d.bg.by = s->background.colour_b[1]; d.bg.bz = s->background.colour_b[2];
    d.bg.gamma = s->background.gamma;
    d.bg.scale = s->background.scale;
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const rpt_sphere& a = s->spheres[i];
        d.spheres[i] = DevSphere{a.center[0], a.center[1], a.center[2], a.radius, a.material};
    }
    for (uint32_t i = 0; i < s->n_planes; ++i) {
        const rpt_plane& a = s->planes[i];
        d.planes[i] = DevPlane{a.normal[0], a.normal[1], a.normal[2], a.point[0], a.point[1], a.point[2], a.min_denom, a.material, a.max_t};
    }
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        const rpt_light& a = s->lights[i];
        d.lights[i] = DevLight{a.type, a.position[0], a.position[1], a.position[2], a.emission[0], a.emission[1], a.emission[2], a.radius, a.area};
    }
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const rpt_material& a = s->materials[i];
        DevMaterial& m = d.materials[i];
        m.mask = a.mask; m.proc_kind = a.proc_kind;
        for (int k = 0; k < 3; ++k) { m.rgb[k] = a.rgb[k]; m.emission[k] = a.emission[k]; }
        m.anisotropic = a.anisotropic; m.metallic = a.metallic; m.roughness = a.roughness; m.subsurface = a.subsurface;
        m.specular_tint = a.specular_tint; m.sheen = a.sheen; m.sheen_tint = a.sheen_tint; m.clearcoat = a.clearcoat;
        m.clearcoat_gloss = a.clearcoat_gloss; m.spec_trans = a.spec_trans; m.ior = a.ior;
        for (int k = 0; k < 4; ++k) m.proc_params[k] = a.proc_params[k];
    }
    d.sdf.n_prims = s->sdf.n_prims; d.sdf.max_steps = s->sdf.max_steps; d.sdf.material = s->sdf.material;
    d.sdf.smooth_k = s->sdf.smooth_k; d.sdf.hit_eps = s->sdf.hit_eps; d.sdf.max_t = s->sdf.max_t; d.sdf.normal_eps = s->sdf.normal_eps;
    for (uint32_t i = 0; i < s->sdf.n_prims; ++i) {
        const rpt_sdf_prim& a = s->sdf.prims[i];
        d.sdf.prims[i] = DevSdfPrim{a.kind, a.center[0], a.center[1], a.center[2], a.params[0], a.params[1]};
    }
    ctx->camera = s->camera;
    ctx->large = false;
    ctx->has_scene = true;
    return RPT_OK;
}

uint32_t rpt_tile_row_count(uint32_t height, uint32_t tile_rows, uint32_t rank, uint32_t world)
{
    return tile_row_count(height, tile_rows, rank, world);
}

uint32_t rpt_tile_global_row(uint32_t local_row, uint32_t tile_rows, uint32_t rank, uint32_t world)
{
    if (tile_rows == 0 || world == 0) return 0;
    return tile_global_row(local_row, tile_rows, rank, world);
}

int rpt_render_device(rpt_ctx* ctx, float* pixels_dev, uint32_t width, uint32_t height, uint64_t frames_done, uint32_t spp,
                      uint64_t seed, uint32_t flags, uint32_t tile_rows, uint32_t rank, uint32_t world, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_render_device: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_scene) { set_err(ctx, "rpt_render_device: no scene uploaded"); return RPT_ERR_NO_SCENE; }
    if (!pixels_dev || width == 0 || height == 0 || world == 0 || rank >= world || tile_rows == 0) {
        set_err(ctx, "rpt_render_device: invalid argument (pixels=%p width=%u height=%u tile_rows=%u rank=%u world=%u)",
                (void*)pixels_dev, width, height, tile_rows, rank, world);
        return RPT_ERR_INVALID_ARG;
    }
    if ((uint64_t)width * height > 0xFFFFFFFFull) { set_err(ctx, "rpt_render_device: image too large for 32-bit pixel indices"); return RPT_ERR_INVALID_ARG; }
    if (((uintptr_t)pixels_dev & 15u) != 0) { set_err(ctx, "rpt_render_device: pixels must be 16-byte aligned"); return RPT_ERR_INVALID_ARG; }
    if (spp == 0) return RPT_OK;
    if (world == 1) tile_rows = height;                              // one block: local row == global row

    RPT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SceneSmallSdf scs = ctx->scene;
    SceneLarge scl = ctx->scene_large;
    scs.cam = scl.cam = make_camera(ctx->camera, (float)width, (float)height);
    const SceneSmall sc = scs;                                       // the plain part (slicing is intended)
    const bool has_sdf = !ctx->large && scs.sdf.n_prims > 0;

    RenderParams rp;
    rp.pixels = pixels_dev;
    rp.width = width; rp.height = height;
    rp.rows_local = tile_row_count(height, tile_rows, rank, world);
    rp.tile_rows = tile_rows; rp.rank = rank; rp.world = world;
    rp.spp = spp;
    rp.frames_done = frames_done;
    rp.seed = seed;
    rp.tiles_x = (width + 15u) / 16u;
    rp.shade_threshold = shade_threshold();
    if (rp.rows_local == 0) return RPT_OK;
    const uint32_t tiles_y = (rp.rows_local + 15u) / 16u;
    const uint64_t nblocks = (uint64_t)rp.tiles_x * tiles_y;
    if (nblocks > 0x7FFFFFFFull) { set_err(ctx, "rpt_render_device: grid too large"); return RPT_ERR_INVALID_ARG; }

    hipStream_t st = (hipStream_t)stream;
    // The LDS tables of the regenerating kernel hold kMaxSppPerLaunch samples: larger
    // batches are split into consecutive launches (the running mean carries over).
    for (uint32_t done = 0; done < spp;) {
        const uint32_t chunk = (spp - done > kMaxSppPerLaunch) ? kMaxSppPerLaunch : (spp - done);
        rp.spp = chunk;
        rp.frames_done = frames_done + done;
        const bool nested = (flags & RPT_RENDER_NESTED_LOOPS) != 0;
        if (ctx->large && nested) hipLaunchKernelGGL(render_large_nested_kernel, dim3((uint32_t)nblocks), dim3(256), 0, st, scl, rp);
        else if (ctx->large) hipLaunchKernelGGL(render_large_regen_kernel, dim3((uint32_t)nblocks), dim3(256), 0, st, scl, rp);
        else if (has_sdf && nested) hipLaunchKernelGGL(render_sdf_nested_kernel, dim3((uint32_t)nblocks), dim3(256), 0, st, scs, rp);
        else if (has_sdf) hipLaunchKernelGGL(render_sdf_regen_kernel, dim3((uint32_t)nblocks), dim3(256), 0, st, scs, rp);
        else if (nested) hipLaunchKernelGGL(render_small_nested_kernel, dim3((uint32_t)nblocks), dim3(256), 0, st, sc, rp);
        else hipLaunchKernelGGL(render_small_regen_kernel, dim3((uint32_t)nblocks), dim3(256), 0, st, sc, rp);
        RPT_HIP_CHECK(ctx, hipGetLastError());
        done += chunk;
    }
    return RPT_OK;
}

int rpt_render(rpt_ctx* ctx, float* pixels, uint32_t width, uint32_t height, uint64_t frames_done, uint32_t spp, uint64_t seed,
               uint32_t flags)
{
    if (!ctx) { set_err(nullptr, "rpt_render: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels || width == 0 || height == 0) { set_err(ctx, "rpt_render: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_scene) { set_err(ctx, "rpt_render: no scene uploaded"); return RPT_ERR_NO_SCENE; }
    RPT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)width * height * 4 * sizeof(float);
    if (bytes > ctx->fb_bytes) {
        if (ctx->fb) { RPT_HIP_CHECK(ctx, hipFree(ctx->fb)); ctx->fb = nullptr; ctx->fb_bytes = 0; }
        RPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->fb, bytes));
        ctx->fb_bytes = bytes;
    }
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->fb, pixels, bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = rpt_render_device(ctx, ctx->fb, width, height, frames_done, spp, seed, flags, height, 0, 1, ctx->stream);
    if (rc != RPT_OK) return rc;
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(pixels, ctx->fb, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return RPT_OK;
}

int rpt_untile_device(rpt_ctx* ctx, const float* gathered_dev, float* image_dev, uint32_t width, uint32_t height,
                      uint32_t tile_rows, uint32_t world, uint32_t rows_padded, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_untile_device: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!gathered_dev || !image_dev || width == 0 || height == 0 || tile_rows == 0 || world == 0) { set_err(ctx, "rpt_untile_device: invalid argument"); return RPT_ERR_INVALID_ARG; }
    for (uint32_t r = 0; r < world; ++r)
        if (tile_row_count(height, tile_rows, r, world) > rows_padded) { set_err(ctx, "rpt_untile_device: rows_padded %u too small", rows_padded); return RPT_ERR_INVALID_ARG; }
    RPT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t total = (uint64_t)width * height;
    const uint64_t nblocks = (total + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(untile_kernel, dim3((uint32_t)nblocks), dim3(256), 0, st, (const float4*)gathered_dev, (float4*)image_dev, width,
                       height, tile_rows, world, rows_padded);
    RPT_HIP_CHECK(ctx, hipGetLastError());
    return RPT_OK;
}

int rpt_convert_to_u8_device(rpt_ctx* ctx, const float* pixels_dev, uint8_t* out_dev, uint32_t width, uint32_t height, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_convert_to_u8_device: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels_dev || !out_dev || width == 0 || height == 0) { set_err(ctx, "rpt_convert_to_u8_device: invalid argument"); return RPT_ERR_INVALID_ARG; }
    RPT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t total = (uint64_t)width * height;
    const uint64_t nblocks = (total + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(convert_to_u8_kernel, dim3((uint32_t)nblocks), dim3(256), 0, st, (const float4*)pixels_dev, (uint32_t*)out_dev, total);
    RPT_HIP_CHECK(ctx, hipGetLastError());
    return RPT_OK;
}

int rpt_convert_to_u8(rpt_ctx* ctx, const float* pixels, uint8_t* frame, uint32_t width, uint32_t height)
{
    if (!ctx) { set_err(nullptr, "rpt_convert_to_u8: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels || !frame || width == 0 || height == 0) { set_err(ctx, "rpt_convert_to_u8: invalid argument"); return RPT_ERR_INVALID_ARG; }
    RPT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)width * height;
    const size_t bytes = n * 16 + n * 4;                              // f32 RGBA in, u8 RGBA out, one allocation
    if (bytes > ctx->fb_bytes) {
        if (ctx->fb) { RPT_HIP_CHECK(ctx, hipFree(ctx->fb)); ctx->fb = nullptr; ctx->fb_bytes = 0; }
        RPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->fb, bytes));
        ctx->fb_bytes = bytes;
    }
    uint8_t* out_dev = reinterpret_cast<uint8_t*>(ctx->fb) + n * 16;
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->fb, pixels, n * 16, hipMemcpyHostToDevice, ctx->stream));
    int rc = rpt_convert_to_u8_device(ctx, ctx->fb, out_dev, width, height, ctx->stream);
    if (rc != RPT_OK) return rc;
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(frame, out_dev, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return RPT_OK;
}

int rpt_synchronize(rpt_ctx* ctx, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_synchronize: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    RPT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    RPT_HIP_CHECK(ctx, hipStreamSynchronize((hipStream_t)stream));
    return RPT_OK;
}

int rpt_probe_rays(rpt_ctx* ctx, const float* rays_dev, uint32_t* out_dev, uint64_t n, uint32_t use_grid, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_probe_rays: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_scene || !ctx->large) { set_err(ctx, "rpt_probe_rays: needs an uploaded large scene"); return RPT_ERR_NO_SCENE; }
    if (!rays_dev || !out_dev) { set_err(ctx, "rpt_probe_rays: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if (n == 0) return RPT_OK;
    RPT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    SceneLarge sc = ctx->scene_large;
    if (!use_grid) sc.use_grid = 0;
    const uint64_t nblocks = (n + 255) / 256;
    hipLaunchKernelGGL(probe_rays_kernel, dim3((uint32_t)nblocks), dim3(256), 0, (hipStream_t)stream, sc, rays_dev, out_dev, n);
    RPT_HIP_CHECK(ctx, hipGetLastError());
    return RPT_OK;
}

int rpt_probe_math(rpt_ctx* ctx, uint32_t fn, const float* a_dev, const float* b_dev, float* out_dev, uint64_t n, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_probe_math: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!a_dev || !b_dev || !out_dev || fn > RPT_PROBE_RNG) { set_err(ctx, "rpt_probe_math: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if (n == 0) return RPT_OK;
    RPT_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t nblocks = (n + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(probe_math_kernel, dim3((uint32_t)nblocks), dim3(256), 0, st, fn, a_dev, b_dev, out_dev, n);
    RPT_HIP_CHECK(ctx, hipGetLastError());
    return RPT_OK;
}

}  // extern "C"
