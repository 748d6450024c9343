// kernels.hip — every __global__ kernel of the library (gfx950) and the host-side launch wrappers
// declared in launch.h.  The C ABI lives in capi.hip.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp
// (build.py).  There is NO CPU fallback anywhere in this library.
// build.py compiles this file three times for the shipped library:
//   as is                   small scenes' megakernel (+ the nested-loop baseline, untile, conversions, probes): strict arithmetic (-ffp-contract=off, correctly rounded
//                           divide / sqrt), the short sequences' range tests TRACKED (dev_math.h, RPT_MATH_MODE 2)
//   -DRPT_PEROP_BUILD       large scenes', SDF scenes' and the compacting kernel: strict arithmetic, the range tests next to every operation
//                           (RPT_MATH_MODE 1) — their walks and marches wait for scalar loads at every step, and such a wait (lgkmcnt)
//                           also waits for the trackers' LDS operations: configs[3] 3 149 against 2 961 Msamples/s, configs[4] 2 898
//                           against 2 822 (round 4; profiles/r4/experiments/range_trackers.txt)
//   -DRPT_RELAXED_BUILD     with -fno-hip-fp32-correctly-rounded-divide-sqrt -ffp-contract=fast (v_rcp / v_rsq based divide and sqrt,
//                           ~2.5 ulp, fused multiply-adds): what RPT_RENDER_FAST_MATH selects.  NOT bit-identical to the reference
//                           arithmetic — an ulp now and then flips a branch and changes a sample by O(1) — so that mode is validated
//                           statistically (tests/test_gpu_parity.py::test_fast_math_mode_is_statistically_equivalent) and never what
//                           bench.py measures.
// each under its own kernel-name suffix and launch namespace — and once, with every form, for A/B builds (-DRPT_AB_KERNELS:
// RPT_MATH_MODE 1 throughout).
#if defined(RPT_RELAXED_BUILD)
#define RPT_RENDER_KERNELS_ONLY       // untile, the u8 conversions and the test probes have no relaxed form
#define RPT_NO_MEDIA_KERNELS          // nor have scenes with participating media (RPT_ERR_UNSUPPORTED)
#define RPT_K(name) name##_fast
#define RPT_LAUNCH_NS rptlaunch_fast
#elif defined(RPT_PEROP_BUILD)
#define RPT_GUARD_PER_OP
#define RPT_RENDER_KERNELS_ONLY
#define RPT_NO_SMALL_KERNELS          // (small scenes' megakernel and nested-loop kernel: the default object)
#define RPT_K(name) name##_perop
#define RPT_LAUNCH_NS rptlaunch_perop
#else
#if !defined(RPT_AB_KERNELS) && !defined(RPT_GUARD_PER_OP)
#define RPT_NO_LARGE_SDF_KERNELS      // (they come from the RPT_PEROP_BUILD object,
#define RPT_NO_COMPACT_KERNELS        //  and so does the compacting kernel of one-sample launches)
#endif
#define RPT_K(name) name
#define RPT_LAUNCH_NS rptlaunch
#endif
#if defined(RPT_PEROP_BUILD) || defined(RPT_AB_KERNELS) || defined(RPT_RELAXED_BUILD) || defined(RPT_GUARD_PER_OP)
// (include/rpt_strict_math.h: the f64 polynomials' coefficients as literals.  In scalar registers they save small scenes' megakernel two
// vector moves per step, +2.2 %; the large-scene kernel, short of scalar registers and scalar issue as it is, loses 5 % with them.)
#define RPT_STRICT_MATH_PLAIN_HORNER
#endif
#ifndef RPT_RELAXED_BUILD
#define RPT_HAS_SIZED_KERNELS         // the instantiations that know table sizes (sized_scene, below): strict builds only
#endif
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/rpt.h"
#include "dev_integrator.h"
#include "dev_sdf_path.h"
#include "dev_scene_large.h"
#include "dev_wavefront.h"
#ifdef RPT_AB_KERNELS                 // measured-slower kernel forms kept for A/B runs only; not in the shipped library
#include "ab/dev_sdf_pool.h"
#endif
#include "launch.h"
#ifndef RPT_RENDER_KERNELS_ONLY
#include "dev_probes.h"
#endif
#if RPT_MATH_MODE == 2
// The device functions a second time, over hipcc's own divide and sqrtf (namespace rptplain; dev_math.h, "two passes"): what sample_guard
// recomputes a sample with.  (The block profiler's scopes stay in the normal pass.)
#define RPT_PLAIN_PASS
#undef RPT_NS
#define RPT_NS rptplain
#pragma push_macro("RPT_PROF")
#undef RPT_PROF
#define RPT_PROF(id) do { } while (0)
#include "dev_scene_large.h"
#ifndef RPT_RENDER_KERNELS_ONLY
#include "dev_probes.h"
#endif
#pragma pop_macro("RPT_PROF")
#undef RPT_PLAIN_PASS
#undef RPT_NS
#define RPT_NS rptdev
#undef RPT_MATH_MODE
#define RPT_MATH_MODE 2
#endif

using namespace rptdev;
#if RPT_MATH_MODE == 2
#define RPT_ROW_NS rptplain           // (material_table_row: see render_small_regen_sized_table_kernel)
#else
#define RPT_ROW_NS rptdev
#endif


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

// coord of tracer.rs:46 and the global pixel index of column `col`, local row `lrow` of this rank's tile
RPT_DEV void pixel_coords(const RenderParams& rp, uint32_t col, uint32_t lrow, float& px, float& py, uint32_t& pixel_index)
{
    const uint32_t grow = tile_global_row(lrow, rp.tile_rows, rp.rank, rp.world);
    // j counts rows from the bottom (par_rchunks, tracer.rs:29-37)
    const float W = (float)rp.width;
    const float H = (float)rp.height;
    const uint32_t j = rp.height - 1u - grow;
    const float x = (float)col;
    const float y = H - (float)j;
    const float xx = x / W;
    const float yy = y / H;
    px = xx;
    py = 1.0f - yy;
    pixel_index = grow * rp.width + col;
}

// Dispatch: units, their order, their hand-off.
//
// The hardware hands out workgroups in the order of blockIdx.x, and what a workgroup of the state-machine kernels renders is
// a UNIT: one 16x16 tile x one chunk of the launch's samples.  Two things decide how full the chip is at the end of a launch
// (tools/dispatch_timeline.py: bottom rows first and one unit per tile, the last 9 % of a 6-round launch and the last 20 % of a
// 3-round one ran at a fraction of the resident waves):
//  * the ORDER within a chunk: most expensive tile first (longest-processing-time order), the cost of a tile being the longest
//    time one of its waves held its slot in the context's previous launch of the same shape (rp.tile_cost -> sched_order_kernel
//    -> rp.tile_order; before anything is known: bottom rows first).  +2.4 % on configs[1], +12 % on configs[3] and [4].
//  * the LENGTH of a unit: a pixel's running mean is sequential, so a tile's chunks must run one after the other — but not in
//    the same workgroup.  Units are drawn from a ticket counter (chunk-major: every tile's chunk c before any tile's chunk
//    c + 1); the unit (c, T) waits until the four waves of (c - 1, T) have published their pixels (agent-scope release ->
//    counter; poll -> agent-scope acquire: the L2s of the XCDs are not coherent with each other).  Its predecessor holds an
//    EARLIER ticket, i.e. it has started and waits for nothing that comes later: every wait ends.  A launch of few rounds of
//    workgroups is cut into enough chunks for ~12 rounds of units (capi.hip, unit_chunks).
// The order and the chunking decide WHEN and WHERE a sample is computed, never its value.
__shared__ uint32_t g_unit[2];        // this workgroup's unit: tile, chunk
__shared__ uint32_t g_unit_t0[4];     // each wave's clock at its start

// the tile this workgroup renders (wave-uniform)
RPT_DEV uint32_t block_tile(const RenderParams& rp)
{
    if (rp.n_chunks != 0u) return (uint32_t)__builtin_amdgcn_readfirstlane((int)g_unit[0]);        // lane_setup put it there
    const RPT_CONST_AS uint32_t* order = (const RPT_CONST_AS uint32_t*)rp.tile_order;               // kernels without units
    return order ? order[blockIdx.x] : gridDim.x - 1u - blockIdx.x;
}

RPT_DEV uint32_t cost_clock() { return (uint32_t)wall_clock64(); }      // s_memrealtime: 100 MHz, one counter for the whole chip (s_memtime is per XCD)

constexpr uint32_t kSyncTimeout = 0u, kSyncTicket = 16u, kSyncDone = 32u;   // dwords of rp.sched_sync: "a wait timed out" (sticky), ticket counter, done[tile] from dword 32 (capi.hip, SchedLayout)

// Takes this workgroup's unit and, for a chunk other than the first, waits for the tile's previous chunk.  Returns the
// unit's share of the launch in `rp` (frames_done, spp).  Contains a barrier; the caller's next barrier (lane_setup's, behind
// the table fill) is the one that holds every wave until thread 0's acquire has completed.
RPT_DEV void unit_begin(const RenderParams& launch, RenderParams& rp)
{
    const uint32_t tid = threadIdx.x;
    if ((tid & 63u) == 0u) g_unit_t0[tid >> 6] = cost_clock();
    if (tid == 0u) {
        uint32_t unit = blockIdx.x;
        if (launch.n_chunks > 1u) unit = __hip_atomic_fetch_add(launch.sched_sync + kSyncTicket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t n_tiles = gridDim.x / launch.n_chunks;
        const uint32_t chunk = unit / n_tiles, pos = unit - chunk * n_tiles;
        g_unit[0] = launch.tile_order ? launch.tile_order[pos] : n_tiles - 1u - pos;
        g_unit[1] = chunk;
    }
    __syncthreads();
    const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_unit[1]);
    rp = launch;
    rp.frames_done = launch.frames_done + (uint64_t)chunk * launch.chunk_spp;
    const uint32_t left = launch.spp - chunk * launch.chunk_spp;
    rp.spp = left < launch.chunk_spp ? left : launch.chunk_spp;
    if (chunk != 0u && tid == 0u) {
        uint32_t* done = launch.sched_sync + kSyncDone + g_unit[0];
        const uint32_t want = 4u * chunk;                               // every wave of every earlier chunk has counted (unit_end)
        uint32_t spins = 0u;
        while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins == (1u << 27)) {                                // (minutes: a lost hand-off must end as an error, not as a hang)
                __hip_atomic_store(launch.sched_sync + kSyncTimeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// A wave's part of the end of a unit: its pixels are published for the tile's next chunk, its time is recorded.  Every wave
// of the workgroup counts exactly once, also one that has no pixel at all (lane_setup).  `stored`: the wave has written pixels.
RPT_DEV void unit_end(const RenderParams& rp, bool stored)
{
    const uint64_t act = __ballot(1);
    const bool first = __lane_id() == (uint32_t)__ffsll((unsigned long long)act) - 1u;
    const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_unit[1]);
    if (rp.n_chunks > 1u && chunk + 1u < rp.n_chunks) {
        if (stored) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's stores have left it
        if (first) {
            if (stored) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (always: the compiler may drop the fence's own wait)
            }
            __hip_atomic_fetch_add(rp.sched_sync + kSyncDone + block_tile(rp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (rp.tile_cost && first && stored) {
        const uint32_t wave = threadIdx.x >> 6;
        const uint32_t slot = block_tile(rp) * 4u + wave;
        rp.tile_cost[slot] = cost_clock() - g_unit_t0[wave];
        if (rp.tile_start) rp.tile_start[slot] = g_unit_t0[wave];
    }
}

RPT_DEV PixelSetup pixel_setup(const RenderParams& rp, uint32_t tid)
{
    // A wave covers an 8x8 pixel block (coherent paths), a 256-thread workgroup 16x16.
    PixelSetup ps;
    const uint32_t tile = block_tile(rp);
    const uint32_t tx = tile % rp.tiles_x;
    const uint32_t ty = tile / rp.tiles_x;
    const uint32_t wave = tid >> 6;
    const uint32_t lane = tid & 63u;
    const uint32_t col = tx * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t lrow = ty * 16u + (wave >> 1) * 8u + (lane >> 3);
    ps.valid = (col < rp.width) && (lrow < rp.rows_local);
    pixel_coords(rp, col, lrow, ps.px, ps.py, ps.pixel_index);
    ps.pix_offset = (size_t)lrow * rp.width + col;
    return ps;
}
RPT_DEV PixelSetup pixel_setup(const RenderParams& rp) { return pixel_setup(rp, threadIdx.x); }

// Where this lane's pixel lives, recomputed at the end of a state-machine kernel from a thread index the compiler cannot
// identify with the prologue's: otherwise the 64-bit address stays live across the whole kernel and spills.
RPT_DEV float4* pixel_address_again(const RenderParams& rp)
{
    uint32_t tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    return reinterpret_cast<float4*>(rp.pixels) + pixel_setup(rp, tid).pix_offset;
}

// mix_color, tracer.rs:108-113, with color = [r, g, b, 1.0] (tracer.rs:59,105)
RPT_DEV void blend(float4& acc, v3 rad, float v)
{
    acc.x = (1.0f - v) * acc.x + rad.x * v;
    acc.y = (1.0f - v) * acc.y + rad.y * v;
    acc.z = (1.0f - v) * acc.z + rad.z * v;
    acc.w = (1.0f - v) * acc.w + 1.0f * v;
}

// A sample is over (its radiance complete, not yet blended): under RPT_MATH_MODE 2 the short divide / square-root sequences did not
// test their operands, they tracked them (dev_math.h).  If this lane's trackers left the range since the last look, what it computed
// may be off by an ulp somewhere: the sample is computed again from its camera ray with the plain operations — the same draws (the
// stream is keyed by pixel and frame), the same arithmetic in hipcc's own divide and sqrtf — and the trackers start clean.  A vote,
// because the second computation is long and practically never needed (configs[1]: about 1 sample in 10^6, a root of exactly -0 or
// a quotient of an infinity).  HASHED as in path_begin.
template <bool HASHED, class S>
RPT_DEV void sample_guard(const S& sc, v3& radiance, float px, float py, FrameKey fkey, uint32_t pixel, uint32_t pixel_b = 0u)
{
#if RPT_MATH_MODE == 2
    const bool ok = guard_sample_ok();
    if (__builtin_expect(__ballot(!ok) != 0ull, 0)) {
        if (!ok) {
            const rptplain::v3 r = rptplain::trace_sample<HASHED>(sc, px, py, fkey, pixel, pixel_b);
            radiance = mk3(r.x, r.y, r.z);
            guard_reset();
        }
    }
#else
    (void)sc; (void)radiance; (void)px; (void)py; (void)fkey; (void)pixel; (void)pixel_b;
#endif
}
RPT_DEV void sample_guard_begin()                                   // once per lane before its first sample
{
#if RPT_MATH_MODE == 2
    guard_reset();
#endif
}

// Megakernel, one thread per pixel, `spp` samples per launch, nested-loop form
// (sample loop outside, bounce loop inside; lanes whose path ended idle until the
// wave's longest path ends).  Kept as the A/B baseline for the regenerating kernel.
// The running mean of tracer.rs:105-117 is carried in registers across the launch's
// samples and updated with the reference's own expression once per sample, so one
// launch of S samples is bit-identical to S reference render() calls; the framebuffer
// is read and written once per launch as float4 (16 B per lane, 128 B per 8-pixel row).
template <class S>
RPT_DEV void render_nested_body(const S& sc, const RenderParams& launch)
{
    RenderParams rp = launch;
    rp.n_chunks = 0u;                                               // (no units: one workgroup per tile, all samples)
    const PixelSetup ps = pixel_setup(rp);
    if (!ps.valid) return;
    float4* pix = reinterpret_cast<float4*>(rp.pixels) + ps.pix_offset;
    float4 acc = *pix;
    sample_guard_begin();
    for (uint32_t s = 0; s < rp.spp; ++s) {
        const uint64_t frames = rp.frames_done + s;
        const FrameKey fkey = frame_key_hd(rp.seed, frames);
        const float v = 1.0f / (float)(frames + 1);                 // tracer.rs:115
        v3 rad = trace_sample(sc, ps.px, ps.py, fkey, ps.pixel_index);
        sample_guard<false>(sc, rad, ps.px, ps.py, fkey, ps.pixel_index);
        blend(acc, rad, v);
    }
    *pix = acc;
}

// The scene of a render kernel is its FIRST argument, by value: the launch puts it into the kernarg segment, and the kernel reads
// it from there through a pointer the compiler cannot see through.  Read as a plain by-value argument hipcc hoists its loads into
// the prologue and spills the SGPRs (headline kernel: 16 spilled SGPRs, -2.3 %; SDF march kernel: 50, -2.3 %), and for the largest
// kernels keeps a copy of the whole 2 KB struct in every lane's scratch (SDF scenes with media: 2.6 KB per lane, -60 %).
// (The large-scene megakernels take theirs plainly: 300 B of pointers and grid parameters, 12 spilled VGPRs instead of 16, +-0 %.)
#ifndef RPT_SCENE_ARGUMENT_PLAIN
template <class S>
RPT_DEV const S& kernarg_scene(const S& by_value)
{
    (void)by_value;
    const RPT_CONST_AS S* p = (const RPT_CONST_AS S*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const S*)p;
}
#else
template <class S>
RPT_DEV const S& kernarg_scene(const S& by_value) { return by_value; }
#endif

// Table sizes known at compile time.  A small scene's tables are loops over n_spheres / n_planes / n_lights entries of the kernarg
// segment: wave-uniform loops, each entry fetched with scalar loads whose offsets the loop computes and each fetch waited for where it
// is used.  A kernel that KNOWS the sizes (an assumption on the loaded counts; the host launches it only for scenes that have exactly
// these sizes) unrolls the loops, merges the loads of neighbouring entries and keeps no loop state:
//   * 2 spheres, 1 plane, 1 light — the reference's own scene (analytical.rs:15-16, 41, 70, 194) — for the megakernel and the compacting
//     kernel: configs[1] 13.15 -> 13.65 Gsamples/s, no spilled SGPR left (26 before), one-sample launches +4 ... 6 %.  (Each count
//     matters: lights alone 13.27, lights + planes 13.48; the number of material patches does not: it stays data.)
//   * an SDF object of 1, 2, 3 or 4 primitives over 1 plane under 1 light, for the SDF march kernel, whose every march step loops over
//     the primitives: configs[3] 3.38 -> 3.64 (the primitives alone 3.53).
// Every other scene takes the kernels with the sizes as data.  Nothing about the arithmetic changes: the same functions run on the same
// values in the same order.  RPT_NO_SIZED_KERNELS=1 takes the general kernels (tests compare the two).
template <uint32_t NS, uint32_t NP, uint32_t NL>
RPT_DEV const SceneSmall& sized_scene(const SceneSmall& s)
{
    __builtin_assume(s.n_spheres == NS);
    __builtin_assume(s.n_planes == NP);
    __builtin_assume(s.n_lights == NL);
    return s;
}
#define RPT_REFERENCE_SIZES 2u, 1u, 1u
template <uint32_t NPRIMS, class S>
RPT_DEV const S& sized_sdf_scene(const S& s)
{
    __builtin_assume(s.sdf.n_prims == NPRIMS);
    __builtin_assume(s.n_planes == 1u);
    __builtin_assume(s.n_lights == 1u);
    return s;
}

// The shipped library holds ONE nested-loop kernel, the baseline of the reference's own scene class; the other scene classes' only
// in A/B builds (-DRPT_AB_KERNELS, build.py --ab), where the parity tests run every form against the oracle.
#ifndef RPT_NO_SMALL_KERNELS
__global__ __launch_bounds__(256) void RPT_K(render_small_nested_kernel)(const SceneSmall sc, const RenderParams rp) { render_nested_body(kernarg_scene(sc), rp); }
#endif
#ifdef RPT_AB_KERNELS
__global__ __launch_bounds__(256) void RPT_K(render_large_nested_kernel)(const SceneLarge sc, const RenderParams rp) { render_nested_body(kernarg_scene(sc), rp); }
__global__ __launch_bounds__(256) void RPT_K(render_sdf_nested_kernel)(const SceneSmallSdf sc, const RenderParams rp) { render_nested_body(kernarg_scene(sc), rp); }
#endif
#if defined(RPT_AB_KERNELS) && !defined(RPT_NO_MEDIA_KERNELS)
// The same kernels for scenes with participating media (dev_scene.h WithMedia, dev_media.h): every form below has one.
__global__ __launch_bounds__(256) void RPT_K(render_small_nested_media_kernel)(const WithMedia<SceneSmall> sc, const RenderParams rp) { render_nested_body(kernarg_scene(sc), rp); }
__global__ __launch_bounds__(256) void RPT_K(render_large_nested_media_kernel)(const WithMedia<SceneLarge> sc, const RenderParams rp) { render_nested_body(kernarg_scene(sc), rp); }
__global__ __launch_bounds__(256) void RPT_K(render_sdf_nested_media_kernel)(const WithMedia<SceneSmallSdf> sc, const RenderParams rp) { render_nested_body(kernarg_scene(sc), rp); }
#endif

// The production megakernel.  Same arithmetic per sample, different schedule:
//  * each lane runs its pixel's whole sample loop as a state machine (dev_integrator.h,
//    PathRegs); when its path ends it blends the sample into its running mean and starts
//    the next camera path at once (path regeneration);
//  * a bounce is split into TRACE (the geometry pass of closest_hit + the miss / emitter exits:
//    what every ray needs) and SHADE (normal, material layering, State::finalize, next-event
//    estimation, Disney BSDF sampling: what only a surface hit needs, ~5x the instructions).  A lane
//    that hits a surface parks one dword (GeomHit) and waits; the wave runs SHADE only when at least
//    `shade_threshold` lanes are parked (wave ballot + popcount), or nobody is left to trace.  The
//    expensive block therefore executes with most lanes active, while the cheap one absorbs the
//    divergence.
#ifndef RPT_MAX_SPP_PER_LAUNCH
#define RPT_MAX_SPP_PER_LAUNCH 512
#endif
constexpr uint32_t kMaxSppPerLaunch = RPT_MAX_SPP_PER_LAUNCH;
// ... of the SDF march kernel: its workgroup also parks four float4 per lane and keeps a material table (4 KB); 192 entries leave it
// within the 32 KB that let five workgroups share a CU's LDS
constexpr uint32_t kMaxSppPerLaunchSdf = RPT_MAX_SPP_PER_LAUNCH < 192 ? RPT_MAX_SPP_PER_LAUNCH : 192;

// Minimum waves per SIMD the register allocator must leave room for (2nd argument of
// __launch_bounds__ = waves per SIMD on gfx950); see DESIGN.md for the measurements.
#ifndef RPT_WAVES_PER_SIMD
#define RPT_WAVES_PER_SIMD 5
#endif
#ifndef RPT_SDF_WAVES_PER_SIMD
#define RPT_SDF_WAVES_PER_SIMD 5
#endif

// What the three state-machine kernels below share: per-launch tables and cold per-lane state in LDS.
// The per-sample frame key and blend weight 1/(frames+1) are per-lane values there (lanes drift apart in sample
// index), so the workgroup stages them once.  The pixel's running mean and its constants are touched only when a
// sample ends (once per ~2 bounces); in VGPRs the seven registers they would pin are what separates 4 from 5
// resident waves per SIMD.
struct LaneTables {
    FrameKey* fkey;            // [kMaxSppPerLaunch] frame_key(seed, frames_done + s)
    float* weight;             // [kMaxSppPerLaunch] 1 / (frames_done + s + 1), tracer.rs:115
    float4* acc;               // [256] running mean, tracer.rs:105-117
    float4* pix;               // [256] {coord.x, coord.y, bits(a), bits(b)}: a = pcg_hash(pixel_index), b = pcg_hash(a) (Rng::init)
};

// Takes the workgroup's unit (unit_begin), fills the tables and this lane's slots.  `rp`: the unit's share of the launch.
// False: the lane has no pixel, or the scene has max_depth == 0 (no bounce loop at all: every sample's radiance is zero and
// the lane's pixel is finished here) — the caller returns.
RPT_DEV bool lane_setup(const LaneTables& lt, uint32_t max_depth, const RenderParams& launch, RenderParams& rp)
{
    unit_begin(launch, rp);
    for (uint32_t i = threadIdx.x; i < rp.spp; i += 256u) {
        const uint64_t frames = rp.frames_done + i;
        lt.fkey[i] = frame_key_hd(rp.seed, frames);
        lt.weight[i] = 1.0f / (float)(frames + 1);                  // tracer.rs:115
    }
    __syncthreads();
    if (max_depth == 0) {                                           // (its own address computation: sharing the one below keeps the
        const PixelSetup ps0 = pixel_setup(rp);                     //  64-bit address live, and spilled, across the whole kernel)
        const bool any0 = __ballot(ps0.valid) != 0ull;
        if (ps0.valid) {
            float4* pixel0 = reinterpret_cast<float4*>(rp.pixels) + ps0.pix_offset;
            float4 acc = *pixel0;
            for (uint32_t s = 0; s < rp.spp; ++s) blend(acc, mk3(0.0f, 0.0f, 0.0f), lt.weight[s]);
            *pixel0 = acc;
        }
        unit_end(rp, any0);
        return false;
    }
    const PixelSetup ps = pixel_setup(rp);
    if (__ballot(ps.valid) == 0ull) unit_end(rp, false);            // a wave without a pixel still counts for the tile's next chunk
    if (!ps.valid) return false;
    float4* pixel = reinterpret_cast<float4*>(rp.pixels) + ps.pix_offset;
    lt.acc[threadIdx.x] = *pixel;
    const uint32_t pix_a = pcg_hash(ps.pixel_index);
    lt.pix[threadIdx.x] = make_float4(ps.px, ps.py, rpt_u2f(pix_a), rpt_u2f(pcg_hash(pix_a)));
    sample_guard_begin();
    return true;
}

// The end of a state-machine kernel: the lane's running mean goes back to its pixel, the wave ends its unit.
RPT_DEV void lane_finish(const RenderParams& rp, float4 acc)
{
    *pixel_address_again(rp) = acc;
    unit_end(rp, true);
}

// The workgroup's material table (dev_integrator.h, MaterialTable): one row per lane of its first wave, built with the functions SHADE
// would have called at a hit — over hipcc's own divide and sqrtf in the object that tracks operand ranges instead of testing them
// (a row is built outside any sample: there is nobody to compute it a second time).  Every lane of the workgroup must get here.
template <bool SDF, class S>
RPT_DEV MaterialTable<SDF> material_table_build(const S& sc, uint32_t ns, uint32_t np, float4* rows)
{
    if (threadIdx.x < (4u << (ns + np + (SDF ? 1u : 0u)))) RPT_ROW_NS::material_table_row<SDF>(sc, ns, np, threadIdx.x, rows);
    __syncthreads();
    MaterialTable<SDF> t;
    t.rows = rows; t.ns = ns; t.np = np;
    material_table_procedural(sc, ns, np, t);
    return t;
}
// Which scenes: the host's side of the same rule (render(), below).
template <class S>
inline bool material_table_fits(const S& sc, uint32_t sdf_material, bool has_sdf)
{
    if (sc.n_spheres + sc.n_planes + (has_sdf ? 1u : 0u) > kMatTableBits) return false;
    uint32_t n_procedural = 0;
    for (uint32_t i = 0; i < sc.n_spheres; ++i) n_procedural += sc.materials[sc.spheres[i].material].proc_kind != 0u;
    for (uint32_t k = 0; k < sc.n_planes; ++k) n_procedural += sc.materials[sc.planes[k].material].proc_kind != 0u;
    if (has_sdf) n_procedural += sc.materials[sdf_material].proc_kind != 0u;
    return n_procedural <= 1u;
}

enum : uint32_t { ST_TRACE = 0u, ST_SHADE = 1u, ST_DONE = 2u, ST_FINISH = 3u, ST_MISS = 4u };

// Three blocks, two waiting rooms.  TRACE (closest_hit's geometry pass + the emitter exit) runs at once for every lane that has a
// ray; afterwards each live lane waits in one of two rooms: SHADE (a surface was hit) or FINISH (the path is over: the
// background of a miss still to be added, blend into the pixel's running mean, the pixel's next camera path).  Per pass the wave
// runs ONE room: SHADE when `shade_threshold` lanes wait there; otherwise FINISH when `finish_threshold` lanes wait there;
// otherwise the fuller of the two (nobody can trace while both wait).  Round 2 ran FINISH un-voted at the top of every pass,
// with 34 % of the lanes, and the background inside TRACE with 46 % (profiles/r2/block_profile_c2.txt); replayed over the oracle's
// path events (tools/sched_sim2.py, sim_finish_room) thresholds 56 / 24 cost 5 % less than that.
template <class M = MaterialPerHit, class S>
RPT_DEV void render_regen_body(const S& sc, const RenderParams& launch, const M& materials = M{})
{
    RPT_PROF_INIT();
    __shared__ FrameKey s_fkey[kMaxSppPerLaunch];
    __shared__ float s_weight[kMaxSppPerLaunch];
    __shared__ float4 s_acc[256];
    __shared__ float4 s_pix[256];
    const uint32_t tid = threadIdx.x;
    RenderParams rp;                                                // this workgroup's unit of the launch
    if (!lane_setup(LaneTables{s_fkey, s_weight, s_acc, s_pix}, sc.max_depth, launch, rp)) return;

    uint32_t s = 0;
    uint32_t state = ST_TRACE;
    PathRegs p;
    GeomHit g;                                                      // what a lane waiting for SHADE parks: one dword
    g.code = 0u;
    {
        const float4 c = s_pix[tid];
        path_begin<true>(sc, p, c.x, c.y, s_fkey[0], rpt_f2u(c.z), rpt_f2u(c.w));
    }

    for (;;) {
        RPT_PROF(PB_PASS);
        if (state == ST_TRACE) {
            RPT_PROF(PB_TRACE);
            const uint32_t what = path_trace_geom_split(sc, DirectQuery{}, p, g);
            state = (what == 2u) ? ST_SHADE : ((what == 0u) ? ST_MISS : ST_FINISH);
        }
        const uint32_t n_shade = (uint32_t)__popcll(__ballot(state == ST_SHADE));
        const uint32_t n_fin = (uint32_t)__popcll(__ballot(state >= ST_FINISH));
        if ((n_shade | n_fin) == 0u) break;
        if (n_shade >= rp.shade_threshold || (n_fin < rp.finish_threshold && n_shade >= n_fin)) {
            if (state == ST_SHADE) {
                RPT_PROF(PB_SHADE);
                state = path_shade_full(sc, DirectQuery{}, p, g, nullptr, nullptr, materials) ? ST_FINISH : ST_TRACE;
            }
        } else if (state >= ST_FINISH) {
            // one site for the paths that ended in TRACE (miss, emitter) and in SHADE (pdf <= 0, depth)
            if (state == ST_MISS) {
                RPT_PROF(PB_BACKGROUND);
                p.radiance = p.radiance + background(sc, p.ray) * p.throughput;
            }
            RPT_PROF(PB_FINISH);
            float4 acc = s_acc[tid];
            { const float4 c = s_pix[tid]; sample_guard<true>(sc, p.radiance, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w)); }
            blend(acc, p.radiance, s_weight[s]);
            s_acc[tid] = acc;
            s += 1;
            if (s >= rp.spp) {
                state = ST_DONE;
            } else {
                const float4 c = s_pix[tid];
                path_begin<true>(sc, p, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w));
                state = ST_TRACE;
            }
        }
    }
    RPT_PROF_FLUSH();
    lane_finish(rp, s_acc[tid]);
}

// The same kernel with FINISH un-voted at the top of every pass and the background inside TRACE (round 2's schedule): what large
// scenes and the inline-march SDF form keep — there TRACE carries the grid walks / sphere marches, a lane parked in a finishing room
// is a lane that does not walk, and the three-room loop above measured 2-4 % SLOWER (10 k spheres, 2048^2 x 32 spp: 1 675 vs
// 1 611-1 648 Msamples/s at finishing thresholds 1-64; profiles/r3/experiments/).
template <class S>
RPT_DEV void render_regen_body_tf(const S& sc, const RenderParams& launch)
{
    RPT_PROF_INIT();
    __shared__ FrameKey s_fkey[kMaxSppPerLaunch];
    __shared__ float s_weight[kMaxSppPerLaunch];
    __shared__ float4 s_acc[256];
    __shared__ float4 s_pix[256];
    const uint32_t tid = threadIdx.x;
    RenderParams rp;                                                // this workgroup's unit of the launch
    if (!lane_setup(LaneTables{s_fkey, s_weight, s_acc, s_pix}, sc.max_depth, launch, rp)) return;

    uint32_t s = 0;
    uint32_t state = ST_TRACE;
    PathRegs p;
    GeomHit g;                                                      // what a lane waiting for SHADE parks: one dword
    g.code = 0u;
    {
        const float4 c = s_pix[tid];
        path_begin<true>(sc, p, c.x, c.y, s_fkey[0], rpt_f2u(c.z), rpt_f2u(c.w));
    }

    for (;;) {
        RPT_PROF(PB_PASS);
        if (state == ST_FINISH) {
            // blend the finished sample into the running mean and start the next one (or retire); one site for
            // the paths that ended in TRACE (miss, emitter) and in SHADE (pdf <= 0, depth)
            RPT_PROF(PB_FINISH);
            float4 acc = s_acc[tid];
            { const float4 c = s_pix[tid]; sample_guard<true>(sc, p.radiance, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w)); }
            blend(acc, p.radiance, s_weight[s]);
            s_acc[tid] = acc;
            s += 1;
            if (s >= rp.spp) {
                state = ST_DONE;
            } else {
                const float4 c = s_pix[tid];
                path_begin<true>(sc, p, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w));
                state = ST_TRACE;
            }
        }
        if (state == ST_TRACE) {
            RPT_PROF(PB_TRACE);
            state = path_trace_geom(sc, DirectQuery{}, p, g) ? ST_SHADE : ST_FINISH;
        }
        const uint64_t m_shade = __ballot(state == ST_SHADE);
        const uint64_t m_go = __ballot(state == ST_TRACE || state == ST_FINISH);
        if ((m_shade | m_go) == 0ull) break;
        if ((uint32_t)__popcll(m_shade) >= rp.shade_threshold || m_go == 0ull) {
            if (state == ST_SHADE) {
                RPT_PROF(PB_SHADE);
                state = path_shade_full(sc, DirectQuery{}, p, g) ? ST_FINISH : ST_TRACE;
            }
        }
    }
    RPT_PROF_FLUSH();
    lane_finish(rp, s_acc[tid]);
}

#ifndef RPT_SMALL_WAVES_PER_SIMD
#define RPT_SMALL_WAVES_PER_SIMD RPT_WAVES_PER_SIMD
#endif
#if defined(RPT_AB_KERNELS) && defined(RPT_SCENE_IN_LDS)
// A/B build ("why not the other shapes"): the scene tables staged in LDS and read from there (ds_read of a
// wave-uniform address, value in a VGPR) instead of from the kernarg segment (s_load, value in an SGPR).
__global__ __launch_bounds__(256, RPT_SMALL_WAVES_PER_SIMD) void RPT_K(render_small_regen_kernel)(const SceneSmall sc, const RenderParams rp)
{
    __shared__ SceneSmall s_scene;
    static_assert(sizeof(SceneSmall) % 4 == 0, "dword copy");
    for (uint32_t i = threadIdx.x; i < sizeof(SceneSmall) / 4; i += 256u) ((uint32_t*)&s_scene)[i] = ((const uint32_t*)&sc)[i];
    __syncthreads();
    render_regen_body(s_scene, rp);
}
#else
#ifndef RPT_NO_SMALL_KERNELS
__global__ __launch_bounds__(256, RPT_SMALL_WAVES_PER_SIMD) void RPT_K(render_small_regen_kernel)(const SceneSmall sc, const RenderParams rp) { render_regen_body(kernarg_scene(sc), rp); }
#endif
#ifndef RPT_NO_SMALL_KERNELS
#ifdef RPT_HAS_SIZED_KERNELS
// ... and that reads a hit's material from a table of the 2^(2 + 1 + 2) cases there are (dev_integrator.h, MaterialTable).
// RPT_NO_MATERIAL_TABLE=1: the kernel below it.
__global__ __launch_bounds__(256, RPT_SMALL_WAVES_PER_SIMD) void RPT_K(render_small_regen_sized_table_kernel)(const SceneSmall sc, const RenderParams rp)
{
    constexpr uint32_t sizes[3] = {RPT_REFERENCE_SIZES};
    __shared__ float4 s_rows[kMatTableRows * kMatRowFloat4s];
    const SceneSmall& s = sized_scene<RPT_REFERENCE_SIZES>(kernarg_scene(sc));
    render_regen_body(s, rp, material_table_build<false>(s, sizes[0], sizes[1], s_rows));
}
__global__ __launch_bounds__(256, RPT_SMALL_WAVES_PER_SIMD) void RPT_K(render_small_regen_sized_kernel)(const SceneSmall sc, const RenderParams rp)
{
    render_regen_body(sized_scene<RPT_REFERENCE_SIZES>(kernarg_scene(sc)), rp);
}
#endif
#endif
#endif
// Large scenes: same schedule; the scene tables are streamed from HBM (dev_scene_large.h).  5 waves per SIMD: 96 VGPRs, 12 of them
// spilled (44 B of scratch per lane).  With the two tiers of cell lists 5 / 6 / 7 waves run at 1 881 / 1 874 / 1 858 Msamples/s (10 k
// spheres, 2048^2 x 32 spp) — and move 0.32 / 42 / 77 GB through HBM per launch: at 6 and 7 waves (80 / 72 VGPRs, 53 / 65 spilled) the
// resident waves' scratch no longer fits the L2s (profiles/r3/c5_megakernel vs c5_megakernel_7waves).  (Before the tiers 7 waves were
// 1.5 % ahead: 5: 1 660, 6: 1 664, 7: 1 689; round 2, hipcc's divide, 2048^2 x 8: 4: 1 165, 5: 1 387, 6: 1 454, 7: 1 372, 8: 1 200.)
#ifndef RPT_LARGE_WAVES_PER_SIMD
#define RPT_LARGE_WAVES_PER_SIMD 5
#endif
#ifndef RPT_NO_LARGE_SDF_KERNELS
__global__ __launch_bounds__(256, RPT_LARGE_WAVES_PER_SIMD) void RPT_K(render_large_regen_kernel)(const SceneLarge sc, const RenderParams rp) { render_regen_body_tf(sc, rp); }
#endif
#ifdef RPT_AB_KERNELS
// Small scenes with the procedural SDF object, the sphere march inside closest_hit / any_hit (RPT_RENDER_SDF_INLINE_MARCH): A/B builds.
__global__ __launch_bounds__(256, RPT_WAVES_PER_SIMD) void RPT_K(render_sdf_regen_kernel)(const SceneSmallSdf sc, const RenderParams rp) { render_regen_body_tf(kernarg_scene(sc), rp); }
#endif
#ifndef RPT_NO_MEDIA_KERNELS
#ifndef RPT_NO_SMALL_KERNELS
__global__ __launch_bounds__(256, RPT_SMALL_WAVES_PER_SIMD) void RPT_K(render_small_regen_media_kernel)(const WithMedia<SceneSmall> sc, const RenderParams rp) { render_regen_body(kernarg_scene(sc), rp); }
#endif
#ifndef RPT_NO_LARGE_SDF_KERNELS
__global__ __launch_bounds__(256, RPT_LARGE_WAVES_PER_SIMD) void RPT_K(render_large_regen_media_kernel)(const WithMedia<SceneLarge> sc, const RenderParams rp) { render_regen_body_tf(sc, rp); }
#endif
#ifdef RPT_AB_KERNELS
__global__ __launch_bounds__(256, RPT_WAVES_PER_SIMD) void RPT_K(render_sdf_regen_media_kernel)(const WithMedia<SceneSmallSdf> sc, const RenderParams rp) { render_regen_body_tf(kernarg_scene(sc), rp); }
#endif
#endif

#ifdef RPT_AB_KERNELS
#include "ab/kernel_large_pair.h"
#include "ab/kernel_large_carry.h"
#include "ab/kernel_large_resume.h"
#endif

#ifdef RPT_AB_KERNELS    // the wavefront form of large scenes (RPT_RENDER_LARGE_WAVEFRONT): A/B builds only since round 4 (the megakernel is ahead at every size)
// Large scenes as a wavefront (dev_wavefront.h): WALK(k) walks the rays SHADE(k-1) listed, SHADE(k) does the rest of the bounce
// for every slot that still has work and lists the next rays.
#ifndef RPT_WF_WALK_WAVES_PER_SIMD
#define RPT_WF_WALK_WAVES_PER_SIMD 6
#endif
#ifndef RPT_WF_SHADE_WAVES_PER_SIMD
#define RPT_WF_SHADE_WAVES_PER_SIMD 4
#endif
__global__ __launch_bounds__(256, RPT_WF_WALK_WAVES_PER_SIMD) void RPT_K(wf_walk_kernel)(const SceneLarge sc, const WfBuffers wb, uint32_t parity, uint32_t refill_at)
{
    RPT_PROF_INIT();
#ifdef RPT_PROFILE_BLOCKS
    __syncthreads();
#endif
    // any_active[p]: raised by SHADE(k), k & 1 == p, when a slot still has work.  Once a SHADE leaves it down, every later launch of
    // the sequence returns at once (the host enqueues the worst-case number of iterations without looking).
    const bool idle = wb.any_active[parity ^ 1u] == 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) wb.any_active[parity] = 0u;
    if (idle) return;
    wf_walk_body(sc, wb, refill_at);
    RPT_PROF_FLUSH();
}

#endif

// One workgroup per 256 slots (= pixels of the tile), in four stages with the workgroup's slots RE-DEALT to its threads in
// between, so that each expensive block runs on full waves whatever the mix of outcomes:
//   G  every thread, its own slot: load the path, add the parked light sample, finish closest_hit (planes, lights);
//      outcome: surface hit -> list S, path over (miss / emitter / waiting sample) -> list F
//   S  thread t < |S|: entry t of list S: material, next-event estimation, BSDF sample; path over -> list F
//   F  thread t < |F|: entry t of list F: background, blend into the pixel's running mean, next camera path
//   P  every thread, its own slot: tests before the grid walk, store the path, this wave's ray segments
// Paths travel between stages through LDS (19 dwords each).  `first`: every slot starts sample 0 (nothing to read).
enum : uint32_t { WFF_PENDING = 4u, WFF_PARK = 8u, WFF_NEWRAY = 16u, WFF_MISS = 32u, WFF_BLEND = 64u };   // record flags above the status bits

struct WfRecords {
    float f[14][256];          // ray o, d; throughput; radiance; hit_dist; scatter pdf
    uint32_t u[5][256];        // rng state, increment; bounce; GeomHit; sample << 8 | flags | status
};

// PathRegs.bounce and .medium share a dword wherever a path is stored (bounce <= 4096, medium < 2^16); kernels without media
// never look at the upper half.
template <bool MEDIA>
RPT_DEV uint32_t pack_bounce(const PathRegs& p) { return MEDIA ? (p.bounce | (p.medium << 16)) : p.bounce; }
template <bool MEDIA>
RPT_DEV void unpack_bounce(uint32_t v, PathRegs& p)
{
    p.bounce = MEDIA ? (v & 0xFFFFu) : v;
    p.medium = MEDIA ? (v >> 16) : 0u;
}

template <bool MEDIA = false>
RPT_DEV void wf_rec_put(WfRecords& r, uint32_t i, const PathRegs& p, uint32_t gcode, uint32_t ctl)
{
    r.f[0][i] = p.ray.o.x; r.f[1][i] = p.ray.o.y; r.f[2][i] = p.ray.o.z;
    r.f[3][i] = p.ray.d.x; r.f[4][i] = p.ray.d.y; r.f[5][i] = p.ray.d.z;
    r.f[6][i] = p.throughput.x; r.f[7][i] = p.throughput.y; r.f[8][i] = p.throughput.z;
    r.f[9][i] = p.radiance.x; r.f[10][i] = p.radiance.y; r.f[11][i] = p.radiance.z;
    r.f[12][i] = p.ps.hit_dist; r.f[13][i] = p.ps.scatter_pdf;
    r.u[0][i] = p.rng.state; r.u[1][i] = p.rng.inc; r.u[2][i] = pack_bounce<MEDIA>(p); r.u[3][i] = gcode; r.u[4][i] = ctl;
}

template <bool MEDIA = false>
RPT_DEV void wf_rec_get(const WfRecords& r, uint32_t i, PathRegs& p, uint32_t& gcode, uint32_t& ctl)
{
    p.ray.o = mk3(r.f[0][i], r.f[1][i], r.f[2][i]);
    p.ray.d = mk3(r.f[3][i], r.f[4][i], r.f[5][i]);
    p.throughput = mk3(r.f[6][i], r.f[7][i], r.f[8][i]);
    p.radiance = mk3(r.f[9][i], r.f[10][i], r.f[11][i]);
    p.ps.hit_dist = r.f[12][i]; p.ps.scatter_pdf = r.f[13][i];
    p.rng.state = r.u[0][i]; p.rng.inc = r.u[1][i]; unpack_bounce<MEDIA>(r.u[2][i], p); gcode = r.u[3][i]; ctl = r.u[4][i];
}

// append `value` to a workgroup list in LDS for the lanes that `want` (one LDS atomic per wave)
template <class T>
RPT_DEV void wf_list_add(T* list, uint32_t* count, bool want, uint32_t value)
{
    const uint64_t m = __ballot(want);
    if (m == 0ull) return;
    const uint32_t lane = __lane_id();
    const uint32_t leader = (uint32_t)__ffsll((unsigned long long)m) - 1u;
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, (int)leader);
    if (want) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (T)value;
}

#ifdef RPT_AB_KERNELS
template <class S>
RPT_DEV void wf_shade_body(const S& sc, const RenderParams& rp, const WfBuffers& wb, uint32_t parity, uint32_t first)
{
    constexpr bool M = S::kMedia;
    __shared__ WfRecords rec;
    __shared__ uint32_t l_shade[256], l_fin[256];
    __shared__ uint32_t n_lists[2];
    const uint32_t tid = threadIdx.x;
    const uint32_t slot0 = blockIdx.x * 256u;
    const uint32_t slot = slot0 + tid;
    if (!first && wb.any_active[parity ^ 1u] == 0u) return;         // nothing was left after the previous SHADE (see wf_walk_kernel)
    if (slot < kWalkGroups) wb.group_next[slot * kWalkCounterStride] = 0u;     // the next WALK's segment counters (the previous WALK is over)
    if (tid < 2u) n_lists[tid] = 0u;
    __syncthreads();

    // ---- G
    const bool in_tile = slot < wb.n_slots;
    bool live = in_tile;
    {
        uint4 c = make_uint4(0u, 0u, 0u, WF_DONE);
        if (in_tile && !first) c = wb.ctl[slot];
        live = in_tile && (first || (c.w & 3u) != WF_DONE);
        bool to_shade = false, to_fin = false;
        if (live) {
            PathRegs p;
            GeomHit g;
            g.code = 0u;
            uint32_t ctl;
            if (first) {
                p.ray.o = p.ray.d = p.throughput = p.radiance = mk3(0.0f, 0.0f, 0.0f);
                p.ps.hit_dist = 0.0f; p.ps.scatter_pdf = 0.0f;
                p.rng.state = 0u; p.rng.inc = 1u; p.bounce = 0u; p.medium = 0u;
                ctl = WF_WALKING;                                   // sample 0, nothing to blend: F starts its camera path
                to_fin = true;
            } else {
                const float4 a = wb.ray_o[slot], b = wb.ray_d[slot], t = wb.thr[slot], r = wb.rad[slot];
                p.ray.o = mk3(a.x, a.y, a.z); p.ray.d = mk3(b.x, b.y, b.z);
                p.throughput = mk3(t.x, t.y, t.z); p.ps.hit_dist = t.w;
                p.radiance = mk3(r.x, r.y, r.z); p.ps.scatter_pdf = r.w;
                p.rng.state = c.x; p.rng.inc = c.y; unpack_bounce<M>(c.z, p);
                const uint32_t s = c.w >> 8;
                const uint32_t status = c.w & 3u;
                // last bounce's light sample: visible unless its walk found an occluder
                const bool lit = (c.w & WFF_PENDING) && rpt_f2u(wb.sh_d[slot].w) == 0u;
                v3 gain = mk3(0.0f, 0.0f, 0.0f);
                if (lit) { const float4 cl = wb.c_lit[slot]; gain = mk3(cl.x, cl.y, cl.z); }
                if (c.w & WFF_PARK) {
                    // that light sample belonged to the PREVIOUS sample of the pixel, which ended there: it is blended now (the
                    // current sample was started at once, in its place: tracer.rs:105-117 still sees the samples in order)
                    const float4 pr = wb.prev[slot];
                    v3 r2 = mk3(pr.x, pr.y, pr.z);
                    if (lit) r2 = r2 + gain;
                    const uint64_t frames = rp.frames_done + (s - 1u);
                    float4* pixel = reinterpret_cast<float4*>(rp.pixels) + slot;
                    float4 acc = *pixel;
                    blend(acc, r2, 1.0f / (float)(frames + 1));
                    *pixel = acc;
                } else if (lit) {
                    p.radiance = p.radiance + gain;
                }
                ctl = (s << 8) | status;
                if (status == WF_ENDING) {
                    ctl |= WFF_BLEND;
                    to_fin = true;
                } else {
                    const WaveQuery q{a.w, rpt_f2u(b.w)};
                    const uint32_t what = path_trace_geom_split(sc, q, p, g);
                    if (what == 2u) to_shade = true;
                    else { to_fin = true; ctl |= WFF_BLEND | (what == 0u ? WFF_MISS : 0u); }
                }
            }
            wf_rec_put<M>(rec, tid, p, g.code, ctl);
        }
        wf_list_add(l_shade, &n_lists[0], to_shade, tid);
        wf_list_add(l_fin, &n_lists[1], to_fin, tid);
    }
    __syncthreads();

    // ---- S
    {
        const bool mine = tid < n_lists[0];
        bool to_fin = false;
        uint32_t i = 0u;
        if (mine) {
            i = l_shade[tid];
            PathRegs p;
            uint32_t gcode, ctl;
            wf_rec_get<M>(rec, i, p, gcode, ctl);
            GeomHit g;
            g.code = gcode;
            ShadowReq sr;
            const bool over = path_shade_deferred(sc, p, g, sr);
            const uint32_t si = slot0 + i;
            if (sr.pending) {
                wb.sh_o[si] = make_float4(sr.ray.o.x, sr.ray.o.y, sr.ray.o.z, sr.max_dist);
                wb.sh_d[si] = make_float4(sr.ray.d.x, sr.ray.d.y, sr.ray.d.z, rpt_u2f(0u));
                wb.c_lit[si] = make_float4(sr.c_lit.x, sr.c_lit.y, sr.c_lit.z, 0.0f);
                ctl |= WFF_PENDING;
            }
            if (!over) {
                ctl |= WFF_NEWRAY;
            } else if (!sr.pending) {
                ctl |= WFF_BLEND;
                to_fin = true;
            } else if ((ctl >> 8) + 1u < rp.spp) {                  // over once its last shadow ray is answered: park it, start the next sample
                wb.prev[si] = make_float4(p.radiance.x, p.radiance.y, p.radiance.z, 0.0f);
                ctl = (((ctl >> 8) + 1u) << 8) | (ctl & 0xFFu) | WFF_PARK;
                to_fin = true;                                      // (no WFF_BLEND: F only starts the next camera path)
            } else {
                ctl = (ctl & ~3u) | WF_ENDING;                      // the launch's last sample waits for the answer
            }
            wf_rec_put<M>(rec, i, p, gcode, ctl);
        }
        wf_list_add(l_fin, &n_lists[1], to_fin, i);
    }
    __syncthreads();

    // ---- F
    if (tid < n_lists[1]) {
        const uint32_t i = l_fin[tid];
        PathRegs p;
        uint32_t gcode, ctl;
        wf_rec_get<M>(rec, i, p, gcode, ctl);
        uint32_t s = ctl >> 8;
        bool begin = true;
        if (ctl & WFF_MISS) p.radiance = p.radiance + background(sc, p.ray) * p.throughput;
        if (ctl & WFF_BLEND) {                                      // tracer.rs:105-117 on this pixel's running mean, then its next sample
            const uint64_t frames = rp.frames_done + s;
            float4* pixel = reinterpret_cast<float4*>(rp.pixels) + (slot0 + i);
            float4 acc = *pixel;
            blend(acc, p.radiance, 1.0f / (float)(frames + 1));
            *pixel = acc;
            s += 1u;
            begin = s < rp.spp;
        }
        ctl = (s << 8) | (ctl & (WFF_PENDING | WFF_PARK));
        if (begin) {
            float px, py;
            uint32_t pixel_index;
            const uint32_t pix = slot0 + i;
            pixel_coords(rp, pix % rp.width, pix / rp.width, px, py, pixel_index);
            path_begin(sc, p, px, py, frame_key_hd(rp.seed, rp.frames_done + s), pixel_index);
            ctl |= WF_WALKING | WFF_NEWRAY;
        } else {
            ctl |= WF_DONE;
        }
        wf_rec_put<M>(rec, i, p, gcode, ctl);
    }
    __syncthreads();

    // ---- P
    bool keep = false, qc = false, qs = false;
    if (live) {
        PathRegs p;
        uint32_t gcode, ctl;
        wf_rec_get<M>(rec, tid, p, gcode, ctl);
        float dist = 0.0f;
        uint32_t best = 0xFFFFFFFFu;
        bool walk_closest = false;
        if (ctl & WFF_NEWRAY) walk_closest = closest_before_walk(sc, p.ray, dist, best);
        keep = (ctl & 3u) != WF_DONE;
        wb.ctl[slot] = make_uint4(p.rng.state, p.rng.inc, pack_bounce<M>(p), ctl & ~(WFF_NEWRAY | WFF_MISS | WFF_BLEND));
        if (keep) {
            wb.ray_o[slot] = make_float4(p.ray.o.x, p.ray.o.y, p.ray.o.z, dist);
            wb.ray_d[slot] = make_float4(p.ray.d.x, p.ray.d.y, p.ray.d.z, rpt_u2f(best));
            wb.thr[slot] = make_float4(p.throughput.x, p.throughput.y, p.throughput.z, p.ps.hit_dist);
            wb.rad[slot] = make_float4(p.radiance.x, p.radiance.y, p.radiance.z, p.ps.scatter_pdf);
        }
        qc = keep && (ctl & WFF_NEWRAY) && walk_closest;
        qs = keep && (ctl & WFF_PENDING);
    }
    // this wave's segment of the two ray lists (dev_wavefront.h)
    const uint32_t seg = slot >> 6, lane = tid & 63u;
    const uint64_t below = (1ull << lane) - 1ull;
    const uint64_t mc = __ballot(qc);
    if (qc) wb.closest[seg * 64u + (uint32_t)__popcll(mc & below)] = slot;
    const uint64_t ms = __ballot(qs);
    if (qs) wb.shadow[seg * 64u + (uint32_t)__popcll(ms & below)] = slot;
    const uint64_t mk = __ballot(keep);
    if (lane == 0u && seg < wb.n_seg) {
        wb.cnt_closest[seg] = (uint32_t)__popcll(mc);
        wb.cnt_shadow[seg] = (uint32_t)__popcll(ms);
        if (mk != 0ull) wb.any_active[parity] = 1u;
    }
}

__global__ __launch_bounds__(256, RPT_WF_SHADE_WAVES_PER_SIMD) void RPT_K(wf_shade_kernel)(const SceneLarge sc, const RenderParams rp, const WfBuffers wb, uint32_t parity, uint32_t first)
{
    wf_shade_body(kernarg_scene(sc), rp, wb, parity, first);
}
#ifndef RPT_NO_MEDIA_KERNELS
__global__ __launch_bounds__(256, RPT_WF_SHADE_WAVES_PER_SIMD) void RPT_K(wf_shade_media_kernel)(const WithMedia<SceneLarge> sc, const RenderParams rp, const WfBuffers wb, uint32_t parity, uint32_t first)
{
    wf_shade_body(kernarg_scene(sc), rp, wb, parity, first);
}
#endif
#endif  // RPT_AB_KERNELS (wavefront form)

// Small scenes, FEW samples per launch (the reference's own usage: one render() per redraw).  With nothing to regenerate a
// wave of the megakernel drains: its lanes end one by one and the wave runs on for its longest path.  Here the workgroup's 256
// paths live in LDS and are re-dealt to the threads before every stage, so TRACE and SHADE always run on full waves (the last one
// of a list excepted) and waves that get nothing issue nothing:
//   TRACE  thread t < |T|: entry t of the trace list: closest_hit; miss / emitter -> blend, the pixel's next sample (if any) -> next T
//                                                                   surface -> S
//   SHADE  thread t < |S|: entry t of the shade list: material, light sample, BSDF; path over -> blend, next sample -> next T;
//                                                                   otherwise -> next T
// Same device functions, same per-pixel order of samples: bit-identical to the other kernels.  Built in the RPT_PEROP_BUILD object: its
// paths change lanes between stages, range trackers would have to travel with them, and its stages are bound by their barriers, not by
// instruction issue (1080p x 1 spp 0.277 ms with trackers carried in the path records, 0.2755 with the tests per operation; 800x600
// 0.0852 against 0.0808).
#ifndef RPT_COMPACT_WAVES_PER_SIMD
#define RPT_COMPACT_WAVES_PER_SIMD 5
#endif
template <class S>
RPT_DEV bool compact_finish(const S& sc, const RenderParams& rp, uint32_t i, PathRegs& p, uint32_t& s)
{
    // tracer.rs:105-117 straight on the pixel in HBM (32 B per sample; keeping the 256 running means in LDS would cost the
    // kernel two of its eight resident workgroups per CU)
    const PixelSetup ps = pixel_setup(rp, i);
    float4* pixel = reinterpret_cast<float4*>(rp.pixels) + ps.pix_offset;
    float4 acc = *pixel;
    const uint64_t frames = rp.frames_done + s;
    blend(acc, p.radiance, 1.0f / (float)(frames + 1));
    *pixel = acc;
    s += 1u;
    if (s >= rp.spp) return false;
    path_begin(sc, p, ps.px, ps.py, frame_key_hd(rp.seed, rp.frames_done + s), ps.pixel_index);
    return true;
}

// Per pass two barriers:  TRACE for the trace list (closest_hit only: surface -> S, miss / emitter -> F)  |  SHADE for the
// entries of S from thread 0 up and, at the same time, FINISH (background for a miss, blend, the pixel's next camera path)
// for the entries of F from thread 255 down — |S| + |F| <= 256, so at most one wave has both kinds.
template <class MS = MaterialPerHit, class S>
RPT_DEV void render_compact_body(const S& sc, const RenderParams& launch, const MS& materials = MS{})
{
    constexpr bool M = S::kMedia;
    RenderParams rp = launch;
    rp.n_chunks = 0u;                                               // (no units: one workgroup per tile, all samples)
    __shared__ WfRecords rec;                                       // u[4] = sample index << 1 | "the ray left the scene"
    __shared__ uint8_t l_trace[2][256], l_shade[256], l_fin[256];   // path = pixel of the tile = thread that started it
    __shared__ uint32_t n_trace[2], n_shade[2], n_fin[2];
    const uint32_t tid = threadIdx.x;
    const PixelSetup ps = pixel_setup(rp);
    if (sc.max_depth == 0) {                                        // no bounce loop: every sample's radiance is zero
        if (ps.valid) {
            float4* pixel = reinterpret_cast<float4*>(rp.pixels) + ps.pix_offset;
            float4 acc = *pixel;
            for (uint32_t k = 0; k < rp.spp; ++k) blend(acc, mk3(0.0f, 0.0f, 0.0f), 1.0f / (float)(rp.frames_done + k + 1));
            *pixel = acc;
        }
        return;
    }
    if (tid < 2u) { n_trace[tid] = 0u; n_shade[tid] = 0u; n_fin[tid] = 0u; }
    const uint32_t t0 = cost_clock();
    __syncthreads();
    if (ps.valid) {
        PathRegs p;
        path_begin(sc, p, ps.px, ps.py, frame_key_hd(rp.seed, rp.frames_done), ps.pixel_index);
        wf_rec_put<M>(rec, tid, p, 0u, 0u);
    }
    wf_list_add(l_trace[0], &n_trace[0], ps.valid, tid);
    __syncthreads();

    for (uint32_t cur = 0u;; cur ^= 1u) {
        const uint32_t n_t = n_trace[cur];
        if (n_t == 0u) break;                                       // (the same value in every thread: read behind a barrier)
        {
            bool to_shade = false, to_fin = false;
            uint32_t i = 0u;
            if (tid < n_t) {
                i = l_trace[cur][tid];
                PathRegs p;
                uint32_t gcode, ctl;
                wf_rec_get<M>(rec, i, p, gcode, ctl);
                GeomHit g;
                g.code = 0u;
                const uint32_t what = path_trace_geom_split(sc, DirectQuery{}, p, g);
                to_shade = what == 2u;
                to_fin = !to_shade;
                wf_rec_put<M>(rec, i, p, g.code, (ctl & ~1u) | (what == 0u ? 1u : 0u));
            }
            wf_list_add(l_shade, &n_shade[cur], to_shade, i);
            wf_list_add(l_fin, &n_fin[cur], to_fin, i);
        }
        __syncthreads();
        if (tid == 0u) { n_trace[cur] = 0u; n_shade[cur ^ 1u] = 0u; n_fin[cur ^ 1u] = 0u; }   // last read before this barrier, next written after the next
        {
            const uint32_t n_s = n_shade[cur], n_f = n_fin[cur];
            bool to_trace = false;
            uint32_t i = 0u;
            if (tid < n_s) {
                i = l_shade[tid];
                PathRegs p;
                uint32_t gcode, ctl;
                wf_rec_get<M>(rec, i, p, gcode, ctl);
                GeomHit g;
                g.code = gcode;
                uint32_t s = ctl >> 1;
                if (path_shade_full(sc, DirectQuery{}, p, g, nullptr, nullptr, materials)) to_trace = compact_finish(sc, rp, i, p, s);
                else to_trace = true;
                if (to_trace) wf_rec_put<M>(rec, i, p, 0u, s << 1);
            } else if (255u - tid < n_f) {
                i = l_fin[255u - tid];
                PathRegs p;
                uint32_t gcode, ctl;
                wf_rec_get<M>(rec, i, p, gcode, ctl);
                uint32_t s = ctl >> 1;
                if (ctl & 1u) p.radiance = p.radiance + background(sc, p.ray) * p.throughput;     // tracer.rs:64-68
                to_trace = compact_finish(sc, rp, i, p, s);
                if (to_trace) wf_rec_put<M>(rec, i, p, 0u, s << 1);
            }
            wf_list_add(l_trace[cur ^ 1u], &n_trace[cur ^ 1u], to_trace, i);
        }
        __syncthreads();
    }
    // the workgroup's time, for the dispatch order of the next launch (its waves end together: one figure for all four)
    if (rp.tile_cost && (tid & 63u) == 0u) rp.tile_cost[block_tile(rp) * 4u + (tid >> 6)] = cost_clock() - t0;
}

#ifndef RPT_NO_COMPACT_KERNELS
__global__ __launch_bounds__(256, RPT_COMPACT_WAVES_PER_SIMD) void RPT_K(render_small_compact_kernel)(const SceneSmall sc, const RenderParams rp) { render_compact_body(kernarg_scene(sc), rp); }
#endif
// Frames of a few thousand workgroups (the reference's 800x600 window: 1 875) are a question of how many ROUNDS of workgroups the
// chip needs: six resident per CU make that 1.2 instead of 1.5 rounds.  The price is 80 VGPRs, 35 of the kernel's ~100 live values
// in scratch (116 B per lane, L1/L2-resident at this launch size) — and it is worth it: 800x600 x 1 spp 0.0809 ms against 0.0914
// with five per CU and no spill to speak of (round 4, tools/compact_time.py); from 1080p up the five-per-CU build is 1 % faster.
#ifndef RPT_NO_COMPACT_KERNELS
__global__ __launch_bounds__(256, 6) void RPT_K(render_small_compact_dense_kernel)(const SceneSmall sc, const RenderParams rp) { render_compact_body(kernarg_scene(sc), rp); }
#ifdef RPT_HAS_SIZED_KERNELS
__global__ __launch_bounds__(256, RPT_COMPACT_WAVES_PER_SIMD) void RPT_K(render_small_compact_sized_kernel)(const SceneSmall sc, const RenderParams rp)
{
    render_compact_body(sized_scene<RPT_REFERENCE_SIZES>(kernarg_scene(sc)), rp);
}
__global__ __launch_bounds__(256, 6) void RPT_K(render_small_compact_dense_sized_kernel)(const SceneSmall sc, const RenderParams rp)
{
    render_compact_body(sized_scene<RPT_REFERENCE_SIZES>(kernarg_scene(sc)), rp);
}
// ... with the material table (dev_integrator.h, MaterialTable)
__global__ __launch_bounds__(256, RPT_COMPACT_WAVES_PER_SIMD) void RPT_K(render_small_compact_sized_table_kernel)(const SceneSmall sc, const RenderParams rp)
{
    constexpr uint32_t sizes[3] = {RPT_REFERENCE_SIZES};
    __shared__ float4 s_rows[kMatTableRows * kMatRowFloat4s];
    const SceneSmall& s = sized_scene<RPT_REFERENCE_SIZES>(kernarg_scene(sc));
    render_compact_body(s, rp, material_table_build<false>(s, sizes[0], sizes[1], s_rows));
}
__global__ __launch_bounds__(256, 6) void RPT_K(render_small_compact_dense_sized_table_kernel)(const SceneSmall sc, const RenderParams rp)
{
    constexpr uint32_t sizes[3] = {RPT_REFERENCE_SIZES};
    __shared__ float4 s_rows[kMatTableRows * kMatRowFloat4s];
    const SceneSmall& s = sized_scene<RPT_REFERENCE_SIZES>(kernarg_scene(sc));
    render_compact_body(s, rp, material_table_build<false>(s, sizes[0], sizes[1], s_rows));
}
#endif
#endif
#ifndef RPT_NO_MEDIA_KERNELS
#ifndef RPT_NO_COMPACT_KERNELS
__global__ __launch_bounds__(256, RPT_COMPACT_WAVES_PER_SIMD) void RPT_K(render_small_compact_media_kernel)(const WithMedia<SceneSmall> sc, const RenderParams rp) { render_compact_body(kernarg_scene(sc), rp); }
#endif
#endif

#ifdef RPT_AB_KERNELS
#include "ab/kernel_sdf_compact.h"
#endif

#ifdef RPT_AB_KERNELS    // round 2's three-room march kernel (RPT_RENDER_SDF_THREE_ROOM_MARCH): A/B builds only since round 4
// SDF scenes, resumable march (dev_sdf_path.h).  Per lane:
//   MARCH_P --(march over)--> RESOLVE --(miss / emitter)--> next sample: MARCH_P
//                                     --(surface)--> MARCH_S --(march over)--> SHADE --> MARCH_P
//                                     --(shadow ray needs no march)---------> SHADE
// Per wave, each pass runs ONE of the three blocks for the lanes waiting at it: march steps while at
// least `march_min_lanes` lanes are marching (or nobody waits elsewhere); once only a few stragglers are
// left, the fuller of RESOLVE / SHADE — whose lanes then start new marches next to the stragglers.
// Measured on configs[3] (MI355X): min lanes 1: 1.81, 2: 2.11, 4: 2.28, 8: 2.31, 16: 2.01, 32: 1.63
// Gsamples/s; the bounce-granular kernel: 1.97.  (A policy with RESOLVE/SHADE waiting rooms that fire when
// 24 lanes wait, as in the regeneration kernel, is slower than bounce-granular: three rooms dilute 64 lanes.)
enum : uint32_t { SM_MARCH_P = 0u, SM_MARCH_S = 1u, SM_RESOLVE = 2u, SM_SHADE = 3u, SM_DONE = 4u, SM_FINISH = 5u };

template <class S>
RPT_DEV void render_sdf_march_body(const S& sc, const RenderParams& launch)
{
    RPT_PROF_INIT();
    __shared__ FrameKey s_fkey[kMaxSppPerLaunch];
    __shared__ float s_weight[kMaxSppPerLaunch];
    __shared__ float4 s_acc[256];
    __shared__ float4 s_pix[256];
    __shared__ float4 s_hit[256];                                   // parked hit point (the shadow march borrows p.ray.o)
    const uint32_t tid = threadIdx.x;
    RenderParams rp;                                                // this workgroup's unit of the launch
    if (!lane_setup(LaneTables{s_fkey, s_weight, s_acc, s_pix}, sc.max_depth, launch, rp)) return;

    uint32_t s = 0;
    uint32_t state = SM_MARCH_P;
    PathRegs p;
    GeomHit g;                                                      // what a lane parks between RESOLVE and SHADE: the accepted
    g.code = 0u;                                                    // mask, the normal (RESOLVE needs it for the shadow ray) and,
    v3 normal = mk3(0.0f, 0.0f, 0.0f);                              // in LDS, the hit point
    MarchRegs m;
    {
        const float4 c = s_pix[tid];
        path_begin<true>(sc, p, c.x, c.y, s_fkey[0], rpt_f2u(c.z), rpt_f2u(c.w));
        march_begin_primary(sc, p, m);
    }

    for (;;) {
        RPT_PROF(PB_PASS);
        if (state == SM_FINISH) {                                   // blend, next sample of the pixel (or retire)
            RPT_PROF(PB_FINISH);
            float4 acc = s_acc[tid];
            blend(acc, p.radiance, s_weight[s]);
            s_acc[tid] = acc;
            s += 1;
            if (s >= rp.spp) {
                state = SM_DONE;
            } else {
                const float4 c = s_pix[tid];
                path_begin<true>(sc, p, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w));
                march_begin_primary(sc, p, m);
                state = SM_MARCH_P;
            }
        }
        const uint64_t w_march = __ballot(state <= SM_MARCH_S);
        const uint32_t n_resolve = (uint32_t)__popcll(__ballot(state == SM_RESOLVE));
        const uint32_t n_shade = (uint32_t)__popcll(__ballot(state == SM_SHADE));
        if (w_march == 0ull && n_resolve == 0u && n_shade == 0u) break;

        const uint32_t n_march = (uint32_t)__popcll(w_march);
        const bool waiting = (n_resolve | n_shade) != 0u;
        if (n_march >= rp.march_min_lanes || !waiting) {
            // march until too few lanes are left marching (and somebody waits) or nobody marches
            for (;;) {
                if (state <= SM_MARCH_S) {
                    RPT_PROF(PB_CLOSEST);                              // (block profile: one march step of the wave)
                    if (march_step(sc.sdf, p.ray.o, m)) state = (state == SM_MARCH_P) ? SM_RESOLVE : SM_SHADE;
                }
                const uint32_t left = (uint32_t)__popcll(__ballot(state <= SM_MARCH_S));
                if (left == 0u || left < rp.march_min_lanes) break;
            }
        } else if (n_shade >= n_resolve) {
            if (state == SM_SHADE) {
                RPT_PROF(PB_SHADE);
                const SdfInjectedQuery q{{m.hit, m.t}, {0.0f, 0u}};
                if (path_shade_full(sc, q, p, g, &normal, &s_hit[tid])) {
                    state = SM_FINISH;
                } else {
                    march_begin_primary(sc, p, m);
                    state = SM_MARCH_P;
                }
            }
        } else {
            if (state == SM_RESOLVE) {
                RPT_PROF(PB_TRACE);
                const SdfInjectedQuery q{{m.hit, m.t}, march_analytic(m)};
                if (path_trace_geom(sc, q, p, g)) {
                    bool scatters = false;
                    if constexpr (S::kMedia) scatters = (p.medium & kMediumScatterNow) != 0u;
                    if (scatters) {
                        // a scatter event inside a medium: p.ray.o is the scatter point, where the shadow ray starts
                        const v3 fhp = p.ray.o;
                        s_hit[tid] = make_float4(fhp.x, fhp.y, fhp.z, 0.0f);
                        state = march_begin_shadow<false>(sc, p, fhp, mk3(0.0f, 0.0f, 0.0f), m) ? SM_MARCH_S : SM_SHADE;
                    } else {
                        normal = hit_normal(sc, p.ray, p.ps.hit_dist, g);
                        const bool front = (dot3(normal, p.ray.d) <= 0.0f);         // State::finalize, globals.rs:53-57
                        const v3 ffnormal = mk3(front ? normal.x : -normal.x, front ? normal.y : -normal.y, front ? normal.z : -normal.z);
                        const v3 fhp = p.ray.o + p.ps.hit_dist * p.ray.d;
                        s_hit[tid] = make_float4(fhp.x, fhp.y, fhp.z, 0.0f);
                        state = march_begin_shadow(sc, p, fhp, ffnormal, m) ? SM_MARCH_S : SM_SHADE;
                    }
                } else {
                    state = SM_FINISH;
                }
            }
        }
    }
    RPT_PROF_FLUSH();
    lane_finish(rp, s_acc[tid]);
}

__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD) void RPT_K(render_sdf_march_kernel)(const SceneSmallSdf sc, const RenderParams rp) { render_sdf_march_body(kernarg_scene(sc), rp); }
#ifndef RPT_NO_MEDIA_KERNELS
__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD) void RPT_K(render_sdf_march_media_kernel)(const WithMedia<SceneSmallSdf> sc, const RenderParams rp) { render_sdf_march_body(kernarg_scene(sc), rp); }
#endif
#endif  // RPT_AB_KERNELS (three-room march)

// SDF scenes, two rooms (dev_sdf_path.h, SdfDeferredQuery).  Per lane:
//   [MARCH_S: the parked shadow ray of the bounce just shaded] -> MARCH_P: the path ray -> WAIT -> one block: add the parked light
//   sample if its ray got through; finish closest_hit; miss / emitter / path over -> blend, the pixel's next sample; surface ->
//   material, light sample (parked), BSDF, next ray -> the marches again.
// Per wave each pass either marches (while at least `march_min_lanes` lanes are marching, or nobody waits) or runs the block
// for the lanes that wait.
enum : uint32_t { S2_MARCH_S = 0u, S2_MARCH_P = 1u, S2_WAIT = 2u, S2_DONE = 3u };

template <class MS = MaterialPerHit, class S>
RPT_DEV void render_sdf_march2_body(const S& sc, const RenderParams& launch, const MS& materials = MS{})
{
    RPT_PROF_INIT();
    __shared__ FrameKey s_fkey[kMaxSppPerLaunchSdf];
    __shared__ float s_weight[kMaxSppPerLaunchSdf];
    __shared__ float4 s_acc[256];
    __shared__ float4 s_pix[256];
#ifdef RPT_SDF_PRIMS_IN_LDS
    // A/B: the SDF object's primitive records staged in LDS and read from there with broadcast reads
    // (sdf_eval_lds) instead of through the scalar cache.  Bit-identical and 1.6 % SLOWER on configs[3] (2 973-2 985 against
    // 3 026-3 031 Msamples/s, two alternating runs each: profiles/r4/experiments/sdf_prims_in_lds.txt): the records then sit in
    // VGPRs the kernel does not have (9 spilled instead of 8), and an LDS read's latency is no shorter than a scalar-cache hit's.
    __shared__ float4 s_prims[2 * kMaxSdfPrims];
    if (threadIdx.x < sc.sdf.n_prims) {                             // (lane_setup's barrier publishes them)
        const DevSdfPrim& pr = sc.sdf.prims[threadIdx.x];
        s_prims[2u * threadIdx.x] = make_float4(rpt_u2f(pr.kind), pr.cx, pr.cy, pr.cz);
        s_prims[2u * threadIdx.x + 1u] = make_float4(pr.p0, pr.p1, 0.0f, 0.0f);
    }
#endif
    __shared__ float4 s_march[256];                                 // a lane's march between passes: t, t_useful, steps (bit 31: hit), accepted
                                                                    // analytic primitives (of the path ray's march, also while the shadow ray is marched)
    float4* const s_sho = g_sdf_sho;                                // the parked shadow ray and light sample of each lane (dev_sdf_path.h);
    float4* const s_shd = g_sdf_shd;                                // gain.w: t_useful of the path ray's march while the shadow ray is marched first
    float4* const s_gain = g_sdf_gain;
    const uint32_t tid = threadIdx.x;
    RenderParams rp;                                                // this workgroup's unit of the launch
    if (!lane_setup(LaneTables{s_fkey, s_weight, s_acc, s_pix}, sc.max_depth, launch, rp)) return;

    uint32_t s = 0;
    uint32_t state = S2_MARCH_P;
    PathRegs p;
    bool pending = false;                                           // a light sample is parked, its shadow ray not answered yet
    bool lit = false;                                               // ... answered: it got through
    bool ending = false;                                            // the path is over once the parked sample is resolved
    {
        const float4 c = s_pix[tid];
        path_begin<true>(sc, p, c.x, c.y, s_fkey[0], rpt_f2u(c.z), rpt_f2u(c.w));
        MarchRegs m;
        march_begin_primary(sc, p, m);
        s_march[tid] = make_float4(0.0f, m.t_useful, rpt_u2f(0u), rpt_u2f(m.accepted));
    }
    const bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;

    for (;;) {
        RPT_PROF(PB_PASS);
        const uint32_t n_march = (uint32_t)__popcll(__ballot(state <= S2_MARCH_P));
        const uint32_t n_wait = (uint32_t)__popcll(__ballot(state == S2_WAIT));
        if (n_march == 0u && n_wait == 0u) break;
        if (n_march >= rp.march_min_lanes || n_wait == 0u) {
            // Nothing of a march is live in registers across the block: a marching lane takes its march from LDS here and puts it
            // back behind the loop (direction and origin are the path's ray or the parked shadow ray).
            const bool mine = state <= S2_MARCH_P;
            MarchRegs m;
            v3 mo = mk3(0.0f, 0.0f, 0.0f);
            m.d = mk3(0.0f, 0.0f, 0.0f); m.t = 0.0f; m.t_useful = 0.0f; m.steps = 0u; m.accepted = 0u; m.hit = false;
            if (mine) {
                const float4 r = s_march[tid];
                m.t = r.x; m.t_useful = r.y; m.steps = rpt_f2u(r.z); m.accepted = rpt_f2u(r.w);
                if (state == S2_MARCH_S) {
                    const float4 so = s_sho[tid], sd = s_shd[tid];
                    mo = mk3(so.x, so.y, so.z); m.d = mk3(sd.x, sd.y, sd.z);
                } else {
                    mo = p.ray.o; m.d = p.ray.d;
                }
            }
            for (;;) {
                if (state <= S2_MARCH_P) {
                    RPT_PROF(PB_CLOSEST);                           // (block profile: one march step of the wave)
#ifdef RPT_SDF_PRIMS_IN_LDS
                    if (march_step_lds(sc.sdf, s_prims, mo, m)) {
#else
                    if (march_step(sc.sdf, mo, m)) {
#endif
                        if (state == S2_MARCH_S) {
                            lit = !(m.hit && (!use_max || m.t < s_sho[tid].w));      // any_hit_small's SDF term
                            if (ending) state = S2_WAIT;
                            else {
                                // the path ray's march, prepared by the block (march_begin_primary's analytic part is ~400
                                // instructions: it must not run here, for the one lane of the wave whose shadow march just ended)
                                march_begin(m, p.ray.d, s_gain[tid].w);     // (m.accepted is the path ray's already)
                                mo = p.ray.o;
                                state = S2_MARCH_P;
                            }
                        } else {
                            state = S2_WAIT;
                        }
                    }
                }
                const uint32_t left = (uint32_t)__popcll(__ballot(state <= S2_MARCH_P));
                if (left == 0u || left < rp.march_min_lanes) break;
            }
            if (mine) s_march[tid] = make_float4(m.t, m.t_useful, rpt_u2f(m.steps | (m.hit ? 0x80000000u : 0u)), rpt_u2f(m.accepted));
        } else if (state == S2_WAIT) {
            RPT_PROF(PB_SHADE);
            if (pending) {                                          // last bounce's light sample: visible unless its march hit the object
                if (lit) { const float4 gn = s_gain[tid]; p.radiance = p.radiance + mk3(gn.x, gn.y, gn.z); }
                pending = false;
            }
            bool over = ending;
            ending = false;
            if (!over) {
                GeomHit g;
                g.code = 0u;
                const float4 r = s_march[tid];                      // the finished march of the path's ray
                const SdfDeferredQuery q{{(rpt_f2u(r.z) & 0x80000000u) != 0u, r.x}, AnalyticPre{r.y, rpt_f2u(r.w)}};
                const uint32_t what = path_trace_geom_split(sc, q, p, g);
                if (what == 0u) { p.radiance = p.radiance + background(sc, p.ray) * p.throughput; over = true; }
                else if (what == 1u) over = true;
                else {
                    // (pending comes back through the parked ray: the query marks it in the slot's direction.w)
                    s_shd[tid].w = 1.0f;
                    over = path_shade_full(sc, q, p, g, nullptr, nullptr, materials);
                    pending = s_shd[tid].w == 0.0f;
                }
            }
            // what comes next for this lane: [the parked shadow ray] then the path's ray (or the end of the path)
            bool new_ray = !over;
            ending = pending && over;
            if (over && !pending) {                                 // blend, next sample of the pixel (or retire)
                RPT_PROF(PB_FINISH);
                float4 acc = s_acc[tid];
                { const float4 c = s_pix[tid]; sample_guard<true>(sc, p.radiance, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w)); }
                blend(acc, p.radiance, s_weight[s]);
                s_acc[tid] = acc;
                s += 1;
                if (s >= rp.spp) {
                    state = S2_DONE;
                } else {
                    const float4 c = s_pix[tid];
                    path_begin<true>(sc, p, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w));
                    new_ray = true;
                }
            }
            float np_tu = 0.0f;                                     // the path ray's march: t_useful and the accepted analytic primitives
            uint32_t np_acc = 0u;
            if (new_ray) {                                          // march_begin_primary's analytic part, once, for every lane of the block
                AnalyticHit ah;
                analytic_closest(sc, p.ray, ah);
                np_tu = sdf_primary_t_useful(sc, ah);
                np_acc = ah.accepted;
            }
            if (pending) {
                s_march[tid] = make_float4(0.0f, sdf_shadow_t_useful(sc, s_sho[tid].w), rpt_u2f(0u), rpt_u2f(np_acc));
                s_gain[tid].w = np_tu;
                state = S2_MARCH_S;
            } else if (new_ray) {
                s_march[tid] = make_float4(0.0f, np_tu, rpt_u2f(0u), rpt_u2f(np_acc));
                state = S2_MARCH_P;
            }
        }
    }
    RPT_PROF_FLUSH();
    lane_finish(rp, s_acc[tid]);
}

#ifndef RPT_NO_LARGE_SDF_KERNELS
__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD) 
void RPT_K(render_sdf_march2_kernel)(const SceneSmallSdf sc, const RenderParams rp) { render_sdf_march2_body(kernarg_scene(sc), rp); }
#ifdef RPT_HAS_SIZED_KERNELS
template <uint32_t NPRIMS>
__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD)
void RPT_K(render_sdf_march2_sized_kernel)(const SceneSmallSdf sc, const RenderParams rp) { render_sdf_march2_body(sized_sdf_scene<NPRIMS>(kernarg_scene(sc)), rp); }
// ... with the material table (dev_integrator.h, MaterialTable): at most one analytical sphere beside the plane and the object
template <uint32_t NPRIMS>
__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD)
void RPT_K(render_sdf_march2_sized_table_kernel)(const SceneSmallSdf sc, const RenderParams rp)
{
    __shared__ float4 s_rows[kMatTableRows * kMatRowFloat4s];
    const SceneSmallSdf& s = sized_sdf_scene<NPRIMS>(kernarg_scene(sc));
    render_sdf_march2_body(s, rp, material_table_build<true>(s, s.n_spheres, 1u, s_rows));
}
#endif
#endif
#ifndef RPT_NO_MEDIA_KERNELS
#ifndef RPT_NO_LARGE_SDF_KERNELS
__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD) void RPT_K(render_sdf_march2_media_kernel)(const WithMedia<SceneSmallSdf> sc, const RenderParams rp) { render_sdf_march2_body(kernarg_scene(sc), rp); }
#endif
#endif

#ifdef RPT_AB_KERNELS
// SDF scenes, workgroup-wide march pool (dev_sdf_pool.h): the lane states are those of the march kernel above, but a
// lane in MARCH_P / MARCH_S has SUBMITTED its march and only polls for the answer; the marching itself is done by
// whichever lanes of the workgroup are serving the queue.  Per pass a wave either runs one of its own blocks — when
// `pool_block_lanes` lanes wait at it, or when there is no march work left to do meanwhile — or serves the queue.
RPT_DEV void render_sdf_pool_body(const SceneSmallSdf& sc, const RenderParams& launch)
{
    RPT_PROF_INIT();
    __shared__ FrameKey s_fkey[kMaxSppPerLaunch];
    __shared__ float s_weight[kMaxSppPerLaunch];
    __shared__ float4 s_acc[256];
    __shared__ float4 s_pix[256];
    __shared__ MarchPool pool;
    pool_init(pool);                                                // (lane_setup has the barrier)
    const uint32_t tid = threadIdx.x;
    RenderParams rp;                                                // this workgroup's unit of the launch
    if (!lane_setup(LaneTables{s_fkey, s_weight, s_acc, s_pix}, sc.max_depth, launch, rp)) return;

    uint32_t s = 0;
    uint32_t state = SM_MARCH_P;
    PathRegs p;
    GeomHit g;
    g.code = 0u;
    v3 normal = mk3(0.0f, 0.0f, 0.0f);
    MarchRegs m;                                                    // the lane's own march: what RESOLVE / SHADE need of it
    MarchJob job;                                                   // the march the lane is stepping for the workgroup
    job.owner = kPoolEmpty;
    uint32_t patience = 0;
    {
        const float4 c = s_pix[tid];
        path_begin<true>(sc, p, c.x, c.y, s_fkey[0], rpt_f2u(c.z), rpt_f2u(c.w));
        pool_begin_primary(pool, sc, p, m);
    }

    // Every wait in here is bounded (pool_take's spin, this pass count): a lost request must show up as a wrong image
    // in the parity tests, never as a wave that does not end.
    for (uint32_t pass = 0; pass < (1u << 22); ++pass) {
        RPT_PROF(PB_PASS);
        if (state == SM_FINISH) {                                   // blend, next sample of the pixel (or retire)
            RPT_PROF(PB_FINISH);
            float4 acc = s_acc[tid];
            blend(acc, p.radiance, s_weight[s]);
            s_acc[tid] = acc;
            s += 1;
            if (s >= rp.spp) {
                state = SM_DONE;
            } else {
                const float4 c = s_pix[tid];
                path_begin<true>(sc, p, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w));
                pool_begin_primary(pool, sc, p, m);
                state = SM_MARCH_P;
            }
        }
        if (state <= SM_MARCH_S && pool_poll(pool, tid, m)) state = (state == SM_MARCH_P) ? SM_RESOLVE : SM_SHADE;
        const uint32_t n_march = (uint32_t)__popcll(__ballot(state <= SM_MARCH_S));
        const uint32_t n_resolve = (uint32_t)__popcll(__ballot(state == SM_RESOLVE));
        const uint32_t n_shade = (uint32_t)__popcll(__ballot(state == SM_SHADE));
        if (n_march == 0u && n_resolve == 0u && n_shade == 0u) {    // every pixel of the wave is finished
            pool_put_back(pool, job);
            break;
        }
        // What to do this pass.  SHADE is the expensive block (~4x RESOLVE): it waits for `pool_shade_lanes` lanes, RESOLVE
        // for `pool_resolve_lanes`; until then the wave serves the queue — but only when it can fill `pool_min_batch` lanes
        // with jobs: a sleeping wave costs the SIMD nothing (other workgroups' waves issue), a quarter-full one does.  After
        // `pool_patience` idle passes it takes whatever there is, so the tail of a tile always drains.
        const uint32_t have = (uint32_t)__popcll(__ballot(job.owner != kPoolEmpty));
        const uint32_t avail = pool_avail(pool);
        const bool patient = patience < rp.pool_patience;
        uint32_t action;                                            // 0 sleep, 1 serve, 2 SHADE, 3 RESOLVE
        if (rp.pool_patience == 0u) {
            // the wave kernel's rule: march while enough of the wave's own marches are outstanding, else the fuller block
            const bool waiting = (n_resolve | n_shade) != 0u;
            if (n_march >= rp.march_min_lanes || !waiting) action = (have + avail != 0u) ? 1u : 0u;
            else action = (n_shade >= n_resolve) ? 2u : 3u;
        } else
        if (n_shade >= rp.pool_shade_lanes) action = 2u;
        else if (n_resolve >= rp.pool_resolve_lanes) action = 3u;
        else if (have + avail >= rp.pool_min_batch || (!patient && have + avail != 0u)) action = 1u;
        else if (!patient && (n_shade | n_resolve) != 0u) action = (n_shade >= n_resolve) ? 2u : 3u;
        else action = 0u;

        if (action >= 2u) {
            pool_put_back(pool, job);
            patience = 0u;
            if (action == 2u) {
                if (state == SM_SHADE) {
                    RPT_PROF(PB_SHADE);
                    const SdfInjectedQuery q{{m.hit, m.t}, {0.0f, 0u}};
                    if (path_shade_full(sc, q, p, g, &normal)) {
                        state = SM_FINISH;
                    } else {
                        pool_begin_primary(pool, sc, p, m);
                        state = SM_MARCH_P;
                    }
                }
            } else {
                if (state == SM_RESOLVE) {
                    RPT_PROF(PB_TRACE);
                    const SdfInjectedQuery q{{m.hit, m.t}, march_analytic(m)};
                    if (path_trace_geom(sc, q, p, g)) {
                        normal = hit_normal(sc, p.ray, p.ps.hit_dist, g);
                        const bool front = (dot3(normal, p.ray.d) <= 0.0f);         // State::finalize, globals.rs:53-57
                        const v3 ffnormal = mk3(front ? normal.x : -normal.x, front ? normal.y : -normal.y, front ? normal.z : -normal.z);
                        const v3 fhp = p.ray.o + p.ps.hit_dist * p.ray.d;
                        state = pool_begin_shadow(pool, sc, p, fhp, ffnormal, m) ? SM_MARCH_S : SM_SHADE;
                    } else {
                        state = SM_FINISH;
                    }
                }
            }
        } else if (action == 1u) {
            // serve the workgroup's queue until one of the wave's own blocks has filled, or too few jobs are left
            patience = patient ? 0u : patience;
            for (uint32_t it = 0; it < 256u; ++it) {
                pool_take(pool, job);
                const uint32_t jobs = (uint32_t)__popcll(__ballot(job.owner != kPoolEmpty));
                if (jobs == 0u) break;
                if (rp.pool_patience != 0u && patient && 2u * jobs < rp.pool_min_batch) break;   // hand the rest back: another wave can merge them
                pool_step(pool, sc.sdf, job);
                if (state <= SM_MARCH_S && pool_poll(pool, tid, m)) state = (state == SM_MARCH_P) ? SM_RESOLVE : SM_SHADE;
                const uint32_t r = (uint32_t)__popcll(__ballot(state == SM_RESOLVE)), sh = (uint32_t)__popcll(__ballot(state == SM_SHADE));
                if (rp.pool_patience == 0u) {
                    const uint32_t left = (uint32_t)__popcll(__ballot(state <= SM_MARCH_S));
                    if ((r | sh) != 0u && left < rp.march_min_lanes) break;
                    continue;
                }
                if (sh >= rp.pool_shade_lanes || r >= rp.pool_resolve_lanes) break;
            }
        } else {
            pool_put_back(pool, job);
            patience += 1u;
            __builtin_amdgcn_s_sleep(4);                            // nothing worth doing yet: let other waves use the SIMD
        }
    }
    RPT_PROF_FLUSH();
    lane_finish(rp, s_acc[tid]);
}

__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD) void RPT_K(render_sdf_pool_kernel)(const SceneSmallSdf sc, const RenderParams rp) { render_sdf_pool_body(kernarg_scene(sc), rp); }

#endif  // RPT_AB_KERNELS

#ifndef RPT_RENDER_KERNELS_ONLY      // (the relaxed build: only kernels that contain relaxed arithmetic are built a second time)
// Scatter rank-major gathered tiles into the full image (one float4 per thread).
__global__ __launch_bounds__(256) void RPT_K(untile_kernel)(const float4* __restrict__ gathered, float4* __restrict__ image,
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
__global__ __launch_bounds__(1024) void RPT_K(sched_order_kernel)(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t n_tiles)
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
__global__ __launch_bounds__(256) void RPT_K(sched_init_kernel)(uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t n_tiles)
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
__global__ __launch_bounds__(256) void RPT_K(convert_to_u8_kernel)(const float4* __restrict__ pixels, uint32_t* __restrict__ out, uint64_t n)
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
__global__ __launch_bounds__(256) void RPT_K(convert_to_u8_at_kernel)(const float4* __restrict__ pixels, uint32_t bw, uint32_t bh,
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

// The test probes (dev_probes.h holds the bodies).  RPT_MATH_MODE 2: like a sample of the render kernels, a record whose operands left
// the range of the short sequences is computed again with the plain operations.
RPT_DEV bool probe_begin()
{
#if RPT_MATH_MODE == 2
    guard_reset();
#endif
    return true;
}
RPT_DEV bool probe_redo()
{
#if RPT_MATH_MODE == 2
    return !guard_sample_ok();
#else
    return false;
#endif
}
#if RPT_MATH_MODE == 2
#define RPT_PROBE_PLAIN(call) rptplain::call
#else
#define RPT_PROBE_PLAIN(call) call
#endif

__global__ __launch_bounds__(256) void RPT_K(probe_math_kernel)(uint32_t fn, const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    probe_begin();
    float r = probe_math_body(fn, a[i], b[i], i);
    if (probe_redo()) r = RPT_PROBE_PLAIN(probe_math_body(fn, a[i], b[i], i));
    out[i] = r;
}

__global__ __launch_bounds__(256) void RPT_K(probe_fn_kernel)(uint32_t fn, const DevCamera cam, const float* __restrict__ in, float* __restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = in + i * RPT_PROBE_IN_STRIDE;
    float* o = out + i * RPT_PROBE_OUT_STRIDE;
    probe_begin();
    probe_fn_body(fn, cam, r, o);
    if (probe_redo()) RPT_PROBE_PLAIN(probe_fn_body(fn, cam, r, o));
}

__global__ __launch_bounds__(256) void RPT_K(probe_rays_kernel)(const SceneLarge sc, const float* __restrict__ rays, uint32_t* __restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    probe_begin();
    probe_rays_body(sc, rays + i * 7, out, i);
    if (probe_redo()) RPT_PROBE_PLAIN(probe_rays_body(sc, rays + i * 7, out, i));
}

#endif  // RPT_RENDER_KERNELS_ONLY

// ---------------------------------------------------------------------------
// launch wrappers (launch.h)
// ---------------------------------------------------------------------------
namespace RPT_LAUNCH_NS {

uint32_t max_spp_per_launch(bool sdf_object) { return sdf_object ? kMaxSppPerLaunchSdf : kMaxSppPerLaunch; }

// Which forms a build holds: the shipped library the default form of every scene class, the compacting kernel of one-sample
// launches and ONE nested-loop baseline (small scenes without media); builds with -DRPT_AB_KERNELS every form ever measured.
// A form that is not there is hipErrorNotSupported here and RPT_ERR_UNSUPPORTED at the C ABI (capi.hip checks the flags first).
hipError_t render(const SceneSmallSdf& scs, const SceneLarge& scl, bool large, bool nested, const RenderParams& rp, uint32_t nblocks, hipStream_t st,
                  const SceneSmallSdf* scs_dev, bool media)
{
    const bool has_sdf = !large && scs.sdf.n_prims > 0;
    const SceneSmall sc = scs;                                       // the plain part (slicing is intended)
    const dim3 tiles(nblocks), wg(256);
    (void)hipGetLastError();                                         // the thread's sticky error may be somebody else's (a host process's own HIP calls)
    (void)scs_dev;
    // the kernels that know the reference scene's table sizes (sized_scene, above)
    static const bool no_sized = getenv("RPT_NO_SIZED_KERNELS") && atoi(getenv("RPT_NO_SIZED_KERNELS")) != 0;
    constexpr uint32_t ref_sizes[3] = {RPT_REFERENCE_SIZES};
    const bool sized = !no_sized && !media && !large && !has_sdf && !nested && sc.n_spheres == ref_sizes[0] && sc.n_planes == ref_sizes[1] &&
                       sc.n_lights == ref_sizes[2];
    const uint32_t sized_sdf = (!no_sized && !media && has_sdf && !nested && sc.n_planes == 1u && sc.n_lights == 1u && scs.sdf.n_prims <= 4u) ? scs.sdf.n_prims : 0u;
    // ... and that read materials from a table: at most one primitive with a procedural material (MaterialTable)
    static const bool no_table = getenv("RPT_NO_MATERIAL_TABLE") && atoi(getenv("RPT_NO_MATERIAL_TABLE")) != 0;
    const bool one_procedural = !large && material_table_fits(sc, scs.sdf.material, has_sdf);
    (void)sized; (void)sized_sdf; (void)no_table; (void)one_procedural;
#ifdef RPT_NO_LARGE_SDF_KERNELS
    if (large || has_sdf || (rp.compact && !nested)) return rptlaunch_perop::render(scs, scl, large, nested, rp, nblocks, st, scs_dev, media);   // (the RPT_PEROP_BUILD object)
#endif
#ifdef RPT_NO_SMALL_KERNELS
    if (!large && !has_sdf && !(rp.compact && !nested)) return hipErrorNotSupported;
#endif
    if (media) {
#ifndef RPT_NO_MEDIA_KERNELS
        // scenes with participating media: the same forms, instantiated for WithMedia<Scene> (csrc/ab/ has none)
        const WithMedia<SceneSmall> msc(sc);
        const WithMedia<SceneSmallSdf> mscs(scs);
        const WithMedia<SceneLarge> mscl(scl);
        if (false) {}
#ifdef RPT_AB_KERNELS
        else if (large && nested) hipLaunchKernelGGL(RPT_K(render_large_nested_media_kernel), tiles, wg, 0, st, mscl, rp);
        else if (has_sdf && nested) hipLaunchKernelGGL(RPT_K(render_sdf_nested_media_kernel), tiles, wg, 0, st, mscs, rp);
        else if (nested) hipLaunchKernelGGL(RPT_K(render_small_nested_media_kernel), tiles, wg, 0, st, msc, rp);
        else if (has_sdf && rp.sdf_resumable_march == 1u) hipLaunchKernelGGL(RPT_K(render_sdf_march_media_kernel), tiles, wg, 0, st, mscs, rp);
        else if (has_sdf && rp.sdf_resumable_march == 0u) hipLaunchKernelGGL(RPT_K(render_sdf_regen_media_kernel), tiles, wg, 0, st, mscs, rp);
#else
        else if (nested || (has_sdf && rp.sdf_resumable_march != 4u)) return hipErrorNotSupported;
#endif
#ifndef RPT_NO_LARGE_SDF_KERNELS
        else if (large) hipLaunchKernelGGL(RPT_K(render_large_regen_media_kernel), tiles, wg, 0, st, mscl, rp);
        else if (has_sdf) hipLaunchKernelGGL(RPT_K(render_sdf_march2_media_kernel), tiles, wg, 0, st, mscs, rp);
#endif
#ifndef RPT_NO_COMPACT_KERNELS
        else if (rp.compact) hipLaunchKernelGGL(RPT_K(render_small_compact_media_kernel), tiles, wg, 0, st, msc, rp);
#endif
#ifndef RPT_NO_SMALL_KERNELS
        else hipLaunchKernelGGL(RPT_K(render_small_regen_media_kernel), tiles, wg, 0, st, msc, rp);
#endif
        return hipGetLastError();
#else
        return hipErrorNotSupported;
#endif
    }
    if (false) {}
#ifdef RPT_AB_KERNELS
    else if (large && nested) hipLaunchKernelGGL(RPT_K(render_large_nested_kernel), tiles, wg, 0, st, scl, rp);
    else if (large && rp.large_pair_walk && scl.use_accel) hipLaunchKernelGGL(RPT_K(render_large_pair_kernel), tiles, wg, 0, st, scl, rp);
    else if (large && rp.large_carry_walk && scl.use_accel) hipLaunchKernelGGL(RPT_K(render_large_carry_kernel), tiles, wg, 0, st, scl, rp);
    else if (large && rp.large_walk_cap != 0u && scl.use_accel) hipLaunchKernelGGL(RPT_K(render_large_resume_kernel), tiles, wg, 0, st, scl, rp);
    else if (has_sdf && nested) hipLaunchKernelGGL(RPT_K(render_sdf_nested_kernel), tiles, wg, 0, st, scs, rp);
    else if (has_sdf && rp.sdf_resumable_march == 2u) hipLaunchKernelGGL(RPT_K(render_sdf_pool_kernel), tiles, wg, 0, st, scs, rp);
    else if (has_sdf && rp.sdf_resumable_march == 3u && scs_dev) hipLaunchKernelGGL(RPT_K(render_sdf_compact_kernel), tiles, wg, 0, st, scs_dev, rp);
    else if (has_sdf && rp.sdf_resumable_march == 1u) hipLaunchKernelGGL(RPT_K(render_sdf_march_kernel), tiles, wg, 0, st, scs, rp);
    else if (has_sdf && rp.sdf_resumable_march == 0u) hipLaunchKernelGGL(RPT_K(render_sdf_regen_kernel), tiles, wg, 0, st, scs, rp);
#else
    else if ((nested && (large || has_sdf)) || (has_sdf && rp.sdf_resumable_march != 4u)) return hipErrorNotSupported;
#endif
#ifndef RPT_NO_LARGE_SDF_KERNELS
    else if (large) hipLaunchKernelGGL(RPT_K(render_large_regen_kernel), tiles, wg, 0, st, scl, rp);
#ifdef RPT_HAS_SIZED_KERNELS
    else if (sized_sdf == 1u && one_procedural && !no_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<1u>, tiles, wg, 0, st, scs, rp);
    else if (sized_sdf == 2u && one_procedural && !no_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<2u>, tiles, wg, 0, st, scs, rp);
    else if (sized_sdf == 3u && one_procedural && !no_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<3u>, tiles, wg, 0, st, scs, rp);
    else if (sized_sdf == 4u && one_procedural && !no_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<4u>, tiles, wg, 0, st, scs, rp);
    else if (sized_sdf == 1u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<1u>, tiles, wg, 0, st, scs, rp);
    else if (sized_sdf == 2u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<2u>, tiles, wg, 0, st, scs, rp);
    else if (sized_sdf == 3u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<3u>, tiles, wg, 0, st, scs, rp);
    else if (sized_sdf == 4u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<4u>, tiles, wg, 0, st, scs, rp);
#endif
    else if (has_sdf) hipLaunchKernelGGL(RPT_K(render_sdf_march2_kernel), tiles, wg, 0, st, scs, rp);
#endif
#ifndef RPT_NO_SMALL_KERNELS
    else if (nested) hipLaunchKernelGGL(RPT_K(render_small_nested_kernel), tiles, wg, 0, st, sc, rp);
#endif
#ifndef RPT_NO_COMPACT_KERNELS
#ifdef RPT_HAS_SIZED_KERNELS
    else if (rp.compact && nblocks <= 3072u && sized && one_procedural && !no_table) hipLaunchKernelGGL(RPT_K(render_small_compact_dense_sized_table_kernel), tiles, wg, 0, st, sc, rp);
    else if (rp.compact && sized && one_procedural && !no_table) hipLaunchKernelGGL(RPT_K(render_small_compact_sized_table_kernel), tiles, wg, 0, st, sc, rp);
    else if (rp.compact && nblocks <= 3072u && sized) hipLaunchKernelGGL(RPT_K(render_small_compact_dense_sized_kernel), tiles, wg, 0, st, sc, rp);
    else if (rp.compact && sized) hipLaunchKernelGGL(RPT_K(render_small_compact_sized_kernel), tiles, wg, 0, st, sc, rp);
#endif
    else if (rp.compact && nblocks <= 3072u) hipLaunchKernelGGL(RPT_K(render_small_compact_dense_kernel), tiles, wg, 0, st, sc, rp);
    else if (rp.compact) hipLaunchKernelGGL(RPT_K(render_small_compact_kernel), tiles, wg, 0, st, sc, rp);
#endif
#ifndef RPT_NO_SMALL_KERNELS
    else {
        // RPT_DEBUG_EXTRA_LDS (bytes, experiments only): pads the workgroup's LDS so that fewer waves fit a CU — how the
        // kernel's throughput depends on resident waves per SIMD (occupancy sensitivity)
        static const unsigned extra_lds = getenv("RPT_DEBUG_EXTRA_LDS") ? (unsigned)atoi(getenv("RPT_DEBUG_EXTRA_LDS")) : 0u;
#ifdef RPT_HAS_SIZED_KERNELS
        if (sized && one_procedural && !no_table) hipLaunchKernelGGL(RPT_K(render_small_regen_sized_table_kernel), tiles, wg, extra_lds, st, sc, rp);
        else if (sized) hipLaunchKernelGGL(RPT_K(render_small_regen_sized_kernel), tiles, wg, extra_lds, st, sc, rp);
        else
#endif
        hipLaunchKernelGGL(RPT_K(render_small_regen_kernel), tiles, wg, extra_lds, st, sc, rp);
    }
#endif
    return hipGetLastError();
}

#ifdef RPT_AB_KERNELS
hipError_t render_wavefront(const SceneLarge& sc, const RenderParams& rp, const WfBuffers& wb, hipStream_t st, bool media)
{
    static const uint32_t refill_at = getenv("RPT_WF_REFILL_AT") ? ((uint32_t)atoi(getenv("RPT_WF_REFILL_AT")) & 63u) : 40u;
    static const uint32_t blocks_per_group = getenv("RPT_WF_BLOCKS_PER_GROUP") ? (uint32_t)atoi(getenv("RPT_WF_BLOCKS_PER_GROUP")) : 6u;
    (void)hipGetLastError();
    const dim3 wg(256), all((wb.n_seg * 64u + 255u) / 256u), walkers(kWalkGroups * (blocks_per_group ? blocks_per_group : 1u));
#ifndef RPT_NO_MEDIA_KERNELS
    const WithMedia<SceneLarge> msc(sc);
    const auto shade = [&](uint32_t parity, uint32_t first) {
        if (media) hipLaunchKernelGGL(RPT_K(wf_shade_media_kernel), all, wg, 0, st, msc, rp, wb, parity, first);
        else hipLaunchKernelGGL(RPT_K(wf_shade_kernel), all, wg, 0, st, sc, rp, wb, parity, first);
    };
#else
    if (media) return hipErrorNotSupported;
    const auto shade = [&](uint32_t parity, uint32_t first) { hipLaunchKernelGGL(RPT_K(wf_shade_kernel), all, wg, 0, st, sc, rp, wb, parity, first); };
#endif
    shade(0u, 1u);
    // a sample takes at most max_depth walks of its path ray; its last shadow ray is walked beside the next sample's first
    // ray, except the launch's last sample's.  Launches after the last useful iteration return at once (any_active), and
    // the host enqueues at most 256 iterations without looking: after every 256th it waits for the stream and reads the
    // flag, so a long bound (deep paths that mostly end early) costs the launches of 256 idle iterations at most —
    // spp * max_depth can reach 2 M — and a bound of up to 256 iterations is enqueued without any wait.
    const uint64_t bound = (uint64_t)rp.spp * sc.max_depth + 1u;
    uint32_t host_active = 1u;
    for (uint64_t k = 1; k <= bound; ++k) {
        const uint32_t parity = (uint32_t)(k & 1u);
        hipLaunchKernelGGL(RPT_K(wf_walk_kernel), walkers, wg, 0, st, sc, wb, parity, refill_at);
        shade(parity, 0u);
        if ((k & 255u) == 0u && k < bound) {
            hipError_t e = hipMemcpyAsync(&host_active, &wb.any_active[parity], sizeof(uint32_t), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) return e;
            if (host_active == 0u) break;
        }
    }
    return hipGetLastError();
}
#else
hipError_t render_wavefront(const SceneLarge&, const RenderParams&, const WfBuffers&, hipStream_t, bool) { return hipErrorNotSupported; }
#endif

#ifndef RPT_RENDER_KERNELS_ONLY
hipError_t sched_init(uint32_t* cost, uint32_t* order, uint32_t n_tiles, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(RPT_K(sched_init_kernel), dim3((n_tiles + 255u) / 256u), dim3(256), 0, st, cost, order, n_tiles);
    return hipGetLastError();
}

hipError_t sched_order(const uint32_t* cost, uint32_t* order, uint32_t n_tiles, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(RPT_K(sched_order_kernel), dim3(1), dim3(1024), 0, st, cost, order, n_tiles);
    return hipGetLastError();
}

hipError_t untile(const float* gathered, float* image, uint32_t width, uint32_t height, uint32_t tile_rows, uint32_t world,
                  uint32_t rows_padded, hipStream_t st)
{
    const uint64_t total = (uint64_t)width * height;
    (void)hipGetLastError();
    hipLaunchKernelGGL(RPT_K(untile_kernel), dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, (const float4*)gathered, (float4*)image,
                       width, height, tile_rows, world, rows_padded);
    return hipGetLastError();
}

hipError_t convert_to_u8(const float* pixels, uint8_t* out, uint64_t n_pixels, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(RPT_K(convert_to_u8_kernel), dim3((uint32_t)((n_pixels + 255) / 256)), dim3(256), 0, st, (const float4*)pixels, (uint32_t*)out, n_pixels);
    return hipGetLastError();
}

hipError_t convert_to_u8_at(const float* pixels, uint32_t bw, uint32_t bh, uint8_t* frame, uint32_t at0, uint32_t at1, uint32_t width,
                            uint32_t height, hipStream_t st)
{
    const uint64_t n = (uint64_t)width * height;
    (void)hipGetLastError();
    hipLaunchKernelGGL(RPT_K(convert_to_u8_at_kernel), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (const float4*)pixels, bw, bh,
                       (uint32_t*)frame, at0, at1, width, height);
    return hipGetLastError();
}

hipError_t probe_math(uint32_t fn, const float* a, const float* b, float* out, uint64_t n, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(RPT_K(probe_math_kernel), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, fn, a, b, out, n);
    return hipGetLastError();
}

hipError_t probe_fn(uint32_t fn, const DevCamera& cam, const float* in, float* out, uint64_t n, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(RPT_K(probe_fn_kernel), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, fn, cam, in, out, n);
    return hipGetLastError();
}

#ifdef RPT_PROFILE_BLOCKS
// development only (dev_prof.h): copy out and clear the block counters
hipError_t prof_read(unsigned long long* out)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * PB_COUNT * 3);
    if (e != hipSuccess) return e;
    static const unsigned long long zeros[PB_COUNT * 3] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zeros, sizeof(zeros));
}
#endif

hipError_t probe_rays(const SceneLarge& sc, const float* rays, uint32_t* out, uint64_t n, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(RPT_K(probe_rays_kernel), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, sc, rays, out, n);
    return hipGetLastError();
}

#endif  // RPT_RENDER_KERNELS_ONLY

}  // namespace RPT_LAUNCH_NS
