// kernel_common.h — what the render kernels' translation units share: the build's arithmetic mode and the two passes over the
// device functions, the pixel / unit / lane scaffolding of a workgroup, the kernarg scene.  Included once by each of k_small.hip,
// k_compact.hip, k_sdf.hip and k_large.hip (one kernel class each) and by k_probes.hip (the test build's probes).
//
// Build (build.py): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -disable-machine-licm
// -mllvm -amdgpu-sched-strategy=max-ilp.  There is NO CPU fallback anywhere in this library.  A render TU is built twice:
//   strict    -ffp-contract=off, correctly rounded divide / sqrt.  k_small.hip TRACKS the range tests of the short divide / sqrt
//             sequences (dev_math.h, RPT_MATH_MODE 2: a flagged sample is computed again with the plain operations, namespace
//             rptplain — hence the second pass over the device functions below); the other three are built with -DRPT_GUARD_PER_OP
//             (RPT_MATH_MODE 1: the test next to every operation) — their walks, marches and barriers wait for scalar loads and
//             LDS at every step, and such a wait also waits for the trackers' DS operations (configs[3] 3 149 against 2 961
//             Msamples/s, configs[4] 2 898 against 2 822: profiles/r4/experiments/range_trackers.txt)
//   relaxed   -DRPT_RELAXED_BUILD -fno-hip-fp32-correctly-rounded-divide-sqrt -ffp-contract=fast (v_rcp / v_rsq based divide and
//             sqrt, ~2.5 ulp, fused multiply-adds): what RPT_RENDER_FAST_MATH selects, under the kernel-name suffix _fast and the
//             launch namespace rptlaunch_fast.  NOT bit-identical to the reference arithmetic — an ulp now and then flips a branch
//             and changes a sample by O(1) — so that mode is validated statistically (tests/test_gpu_parity.py) and bench.py reports
//             it beside the headline (`relaxed`), never as the headline.
#pragma once

#if defined(RPT_RELAXED_BUILD)
#define RPT_K(name) name##_fast
#define RPT_LAUNCH_NS rptlaunch_fast
#else
#define RPT_K(name) name
#define RPT_LAUNCH_NS rptlaunch
#endif
#if defined(RPT_GUARD_PER_OP) || defined(RPT_RELAXED_BUILD)
// (include/rpt_strict_math.h: the f64 polynomials' coefficients as literals.  In scalar registers they save small scenes' megakernel two
// vector moves per step, +2.2 %; the large-scene kernel, short of scalar registers and scalar issue as it is, loses 5 % with them.)
#define RPT_STRICT_MATH_PLAIN_HORNER
#endif
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/rpt.h"
#ifdef RPT_WITH_PROBES
#include "../../include/rpt_test.h"
#endif
#include "dev_integrator.h"
#include "dev_sdf_path.h"
#include "dev_scene_large.h"
#include "launch.h"
#ifdef RPT_WITH_PROBES
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
#ifdef RPT_WITH_PROBES
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
__shared__ uint32_t g_unit[3];        // this workgroup's unit: tile, chunk; "the wait for the previous chunk timed out"
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
// The hand-off counts LANES, not waves: every lane that has stored its pixel adds itself (unit_end: one atomic per group of lanes
// that reaches it together), and the successor waits for `pixels of the tile` x `chunks before it`.  A wave the compiler has split
// on its way out of the state machine therefore counts what it is, in two parts, instead of twice.
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
        g_unit[2] = 0u;
    }
    __syncthreads();
    const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_unit[1]);
    rp = launch;
    rp.frames_done = launch.frames_done + (uint64_t)chunk * launch.chunk_spp;
    const uint32_t left = launch.spp - chunk * launch.chunk_spp;
    rp.spp = left < launch.chunk_spp ? left : launch.chunk_spp;
    if (chunk != 0u && tid == 0u) {
        const uint32_t tile = g_unit[0];
        uint32_t* done = launch.sched_sync + kSyncDone + tile;
        const uint32_t tx = tile % launch.tiles_x, ty = tile / launch.tiles_x;
        const uint32_t cols = launch.width - tx * 16u < 16u ? launch.width - tx * 16u : 16u;
        const uint32_t rows = launch.rows_local - ty * 16u < 16u ? launch.rows_local - ty * 16u : 16u;
        const uint32_t want = cols * rows * chunk;                      // every pixel of every earlier chunk has been stored (unit_end)
        uint32_t spins = 0u;
        while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins == (1u << 27)) {                                // (minutes: a lost hand-off must end as an error, not as a hang)
                // Cannot happen by construction (the predecessor holds an earlier ticket).  If it ever does, the host reads the sticky
                // word behind its next wait (RPT_ERR_HIP) — and a caller that waits on its own stream finds NaN in this tile's pixels
                // (lane_finish), not a plausible image.
                __hip_atomic_store(launch.sched_sync + kSyncTimeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g_unit[2] = 1u;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// The end of a unit for the lanes that reach this point together (`stored`: they have written their pixels): the pixels are
// published for the tile's next chunk, the wave's time is recorded.
RPT_DEV void unit_end(const RenderParams& rp, bool stored)
{
    const uint64_t act = __ballot(1);
    const bool first = __lane_id() == (uint32_t)__ffsll((unsigned long long)act) - 1u;
    const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_unit[1]);
    if (stored && rp.n_chunks > 1u && chunk + 1u < rp.n_chunks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // these lanes' stores have left the wave
        if (first) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (always: the compiler may drop the fence's own wait)
            __hip_atomic_fetch_add(rp.sched_sync + kSyncDone + block_tile(rp), (uint32_t)__popcll(act), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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


#ifndef RPT_MAX_SPP_PER_LAUNCH
#define RPT_MAX_SPP_PER_LAUNCH 512
#endif
constexpr uint32_t kMaxSppPerLaunch = RPT_MAX_SPP_PER_LAUNCH;
// ... of the SDF march kernel: its workgroup also parks four float4 per lane, keeps a material table (4 KB) and counts each pixel's
// samples (1 KB).  LDS is handed out in pieces of 1 280 B (160 KB / 128; measured, round 5: 31 004 B per workgroup runs five workgroups
// per CU, 32 028 B four — 26 pieces x 5 > 160 KB), so five workgroups share a CU up to 25 pieces = 32 000 B: 96 entries
// (the kernel also keeps a dword per lane for marches handed from lane to lane).
#ifndef RPT_MAX_SPP_SDF
#define RPT_MAX_SPP_SDF 96
#endif
constexpr uint32_t kMaxSppPerLaunchSdf = RPT_MAX_SPP_PER_LAUNCH < RPT_MAX_SPP_SDF ? RPT_MAX_SPP_PER_LAUNCH : RPT_MAX_SPP_SDF;

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
enum : uint32_t { LANE_NOTHING = 0u, LANE_PIXEL = 1u, LANE_NO_PIXEL = 2u };
// LANE_NOTHING: the scene has no bounce loop (the whole workgroup is finished here); LANE_NO_PIXEL: this lane of a ragged tile has no
// pixel (its wave goes on: the SDF march kernel keeps such lanes as helpers of their wave's marches, k_sdf.hip).
RPT_DEV uint32_t lane_setup_ex(const LaneTables& lt, uint32_t max_depth, const RenderParams& launch, RenderParams& rp)
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
        if (ps0.valid) {
            float4* pixel0 = reinterpret_cast<float4*>(rp.pixels) + ps0.pix_offset;
            float4 acc = *pixel0;
            for (uint32_t s = 0; s < rp.spp; ++s) blend(acc, mk3(0.0f, 0.0f, 0.0f), lt.weight[s]);
            *pixel0 = acc;
            unit_end(rp, true);
        }
        return LANE_NOTHING;
    }
    const PixelSetup ps = pixel_setup(rp);
    if (!ps.valid) return LANE_NO_PIXEL;
    float4* pixel = reinterpret_cast<float4*>(rp.pixels) + ps.pix_offset;
    lt.acc[threadIdx.x] = *pixel;
    const uint32_t pix_a = pcg_hash(ps.pixel_index);
    lt.pix[threadIdx.x] = make_float4(ps.px, ps.py, rpt_u2f(pix_a), rpt_u2f(pcg_hash(pix_a)));
    sample_guard_begin();
    return LANE_PIXEL;
}
RPT_DEV bool lane_setup(const LaneTables& lt, uint32_t max_depth, const RenderParams& launch, RenderParams& rp)
{
    return lane_setup_ex(lt, max_depth, launch, rp) == LANE_PIXEL;
}

// Sharing a wave's samples among its lanes (round 5).  A lane renders its own pixel's samples one after the other; when they are all
// handed out it takes samples of ANOTHER pixel of its wave that still has some — a wave's lanes would otherwise wait for its slowest
// pixel (block profile: 4 % of the lanes' time on configs[1] at 256 spp, 9 % on configs[3], 10 % on 10 k spheres at 32 spp).  Which lane
// renders a sample never changes it (its random stream is keyed by pixel and frame).  What must not change is the ORDER in which a pixel's
// samples enter its running mean (tracer.rs:105-117 is sequential): a finished sample s of pixel q is blended only when s - 1 has
// been; until then its lane waits (a BLOCKED state the kernels count in no vote: the lane that holds s - 1 is in the same wave and
// never waits for a later sample, so every wait ends).  Per pixel one dword of LDS: samples handed out (low half; the owner's first
// is handed out at set-up) and samples blended (high half); only the pixel's own wave touches it.
RPT_DEV void share_init(uint32_t* count, bool has_pixel) { count[threadIdx.x] = has_pixel ? 1u : 0xFFFFu; }
RPT_DEV uint32_t share_handed_out(const uint32_t* count) { return count[threadIdx.x] & 0xFFFFu; }      // of this lane's own pixel
RPT_DEV bool share_my_turn(const uint32_t* count, uint32_t q, uint32_t s) { return (count[q] >> 16) == s; }
RPT_DEV void share_blended(uint32_t* count, uint32_t q) { atomicAdd(&count[q], 0x10000u); }
// The next sample for a lane that has just blended one: of its own pixel while it has any (`own`: share_handed_out at the top of the
// block, read by every lane of the wave; `needy`: the ballot of own < spp), else of a pixel of its wave that has.  False: none left.
RPT_DEV bool share_next(uint32_t* count, uint32_t spp, uint32_t own, uint64_t needy, uint32_t& q, uint32_t& s)
{
    const uint32_t tid = threadIdx.x;
    uint32_t ns = spp, nq = tid;
    if (own < spp) ns = atomicAdd(&count[tid], 1u) & 0xFFFFu;
    if (ns >= spp && needy != 0ull) {
        const uint32_t from = (tid * 13u + 1u) & 63u;               // (lanes start their search at different places)
        const uint64_t rot = (needy >> from) | ((needy << 1) << (63u - from));
        nq = (tid & ~63u) | ((from + (uint32_t)__builtin_ctzll(rot)) & 63u);
        ns = atomicAdd(&count[nq], 1u) & 0xFFFFu;
    }
    if (ns >= spp) return false;
    q = nq; s = ns;
    return true;
}

// The end of a state-machine kernel: the lane's running mean goes back to its pixel, the wave ends its unit.
RPT_DEV void lane_finish(const RenderParams& rp, float4 acc)
{
    if (rp.n_chunks > 1u && g_unit[2] != 0u) { const float nan = __builtin_nanf(""); acc = make_float4(nan, nan, nan, nan); }   // (unit_begin: a hand-off timed out)
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
    material_table_procedural<SDF>(sc, ns, np, t);
    return t;
}
// ... by class of accepted set (dev_integrator.h, MaterialTableMapped): 64 rows, of which the classes there are get built; the class
// of every accepted set goes to `cls_lds`.  `map`: the launch's MatClassMap in the kernarg segment.
template <class S>
RPT_DEV MaterialTableMapped material_table_build_mapped(const S& sc, const MatClassMap& map, uint32_t ns, uint32_t np, float4* rows, uint8_t* cls_lds)
{
    const uint32_t tid = threadIdx.x;
    reinterpret_cast<uint4*>(cls_lds)[tid] = reinterpret_cast<const uint4*>(map.cls)[tid];     // 4 096 bytes from device memory, 16 per thread
    const uint32_t c = tid & (kMatClasses - 1u);
    if (tid < 4u * kMatClasses && c < map.n_classes) {
        const uint32_t set16 = map.class_set[c];                    // spheres in bits 0-7, planes in bits 8-11
        const uint32_t set = (set16 & ((1u << ns) - 1u)) | (((set16 >> kMaxSpheres) & ((1u << np) - 1u)) << ns);
        RPT_ROW_NS::material_table_row_of<false>(sc, ns, np, set, (tid >> kMatClassBits) & 1u, (tid >> (kMatClassBits + 1u)) & 1u, tid, rows);
    }
    __syncthreads();
    MaterialTableMapped t;
    t.rows = rows; t.cls = cls_lds; t.ns = ns; t.np = np;
    material_table_procedural<false>(sc, ns, np, t);
    return t;
}
// Which scenes: the host's side of the same rule (render(), below).
template <class S>
inline bool material_table_fits(const S& sc, uint32_t sdf_material, bool has_sdf, uint32_t max_bits = kMatTableBits)
{
    if (sc.n_spheres + sc.n_planes + (has_sdf ? 1u : 0u) > max_bits) return false;
    uint32_t n_procedural = 0;
    for (uint32_t i = 0; i < sc.n_spheres; ++i) n_procedural += sc.materials[sc.spheres[i].material].proc_kind != 0u;
    for (uint32_t k = 0; k < sc.n_planes; ++k) n_procedural += sc.materials[sc.planes[k].material].proc_kind != 0u;
    if (has_sdf) n_procedural += sc.materials[sdf_material].proc_kind != 0u;
    return n_procedural <= 1u;
}
