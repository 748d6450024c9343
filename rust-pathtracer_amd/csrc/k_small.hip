// k_small.hip — small scenes' megakernel (BASELINE configs[1], [2]: the headline) and the nested-loop baseline.
// Strict object: the range tests of the short divide / square root TRACKED (kernel_common.h); relaxed object: -DRPT_RELAXED_BUILD.
#include "kernel_common.h"

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

// The one nested-loop kernel of the library: the differential baseline of the reference's own scene class (RPT_RENDER_NESTED_LOOPS).
__global__ __launch_bounds__(256) void RPT_K(render_small_nested_kernel)(const SceneSmall sc, const RenderParams rp) { render_nested_body(kernarg_scene(sc), rp); }

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
enum : uint32_t { ST_TRACE = 0u, ST_SHADE = 1u, ST_DONE = 2u, ST_BLOCKED = 3u, ST_FINISH = 4u, ST_MISS = 5u };    // (the finishing room: state >= ST_FINISH)

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
    __shared__ uint32_t s_count[256];                               // share_* (kernel_common.h): each pixel's samples handed out and blended
    const uint32_t tid = threadIdx.x;
    share_init(s_count, false);                                     // (until the lane is known to have a pixel)
    RenderParams rp;                                                // this workgroup's unit of the launch
    if (!lane_setup(LaneTables{s_fkey, s_weight, s_acc, s_pix}, sc.max_depth, launch, rp)) return;
    share_init(s_count, true);

    uint32_t s = 0;
    uint32_t q = tid;                                               // the pixel this lane renders a sample of: its own while that has any
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
        RPT_PROF_ALIVE((uint32_t)__popcll(__ballot(state != ST_DONE)));
        if (__ballot(state == ST_BLOCKED) != 0ull) { if (state == ST_BLOCKED && share_my_turn(s_count, q, s)) state = ST_FINISH; }
        if (state == ST_TRACE) {
            RPT_PROF(PB_TRACE);
            const uint32_t what = path_trace_geom_split(sc, DirectQuery{}, p, g);
            state = (what == 2u) ? ST_SHADE : ((what == 0u) ? ST_MISS : ST_FINISH);
        }
        const uint32_t n_shade = (uint32_t)__popcll(__ballot(state == ST_SHADE));
        const uint32_t n_fin = (uint32_t)__popcll(__ballot(state >= ST_FINISH));
        if ((n_shade | n_fin) == 0u) break;                         // (a blocked lane waits for one that is in a room)
        if (n_shade >= rp.shade_threshold || (n_fin < rp.finish_threshold && n_shade >= n_fin)) {
            if (state == ST_SHADE) {
                RPT_PROF(PB_SHADE);
                state = path_shade_full(sc, DirectQuery{}, p, g, nullptr, nullptr, materials) ? ST_FINISH : ST_TRACE;
            }
        } else {
            const uint32_t own = share_handed_out(s_count);         // (every lane of the wave: who still has samples to hand out)
            const uint64_t needy = __ballot(own < rp.spp);
            if (state >= ST_FINISH) {
                // one site for the paths that ended in TRACE (miss, emitter) and in SHADE (pdf <= 0, depth)
                if (state == ST_MISS) {
                    RPT_PROF(PB_BACKGROUND);
                    p.radiance = p.radiance + background(sc, p.ray) * p.throughput;
                }
                RPT_PROF(PB_FINISH);
                if (!share_my_turn(s_count, q, s)) {
                    state = ST_BLOCKED;                             // an earlier sample of the pixel is still on its way
                } else {
                    float4 acc = s_acc[q];
                    { const float4 c = s_pix[q]; sample_guard<true>(sc, p.radiance, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w)); }
                    blend(acc, p.radiance, s_weight[s]);
                    s_acc[q] = acc;
                    share_blended(s_count, q);
                    if (!share_next(s_count, rp.spp, own, needy, q, s)) {
                        state = ST_DONE;
                    } else {
                        const float4 c = s_pix[q];
                        path_begin<true>(sc, p, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w));
                        state = ST_TRACE;
                    }
                }
            }
        }
    }
    RPT_PROF_FLUSH();
    lane_finish(rp, s_acc[tid]);
}

#ifndef RPT_SMALL_WAVES_PER_SIMD
#define RPT_SMALL_WAVES_PER_SIMD RPT_WAVES_PER_SIMD
#endif
__global__ __launch_bounds__(256, RPT_SMALL_WAVES_PER_SIMD) void RPT_K(render_small_regen_kernel)(const SceneSmall sc, const RenderParams rp) { render_regen_body(kernarg_scene(sc), rp); }
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
// The material table for ANY scene of at most FOUR primitives (kernel_common.h, material_table_fits; three spheres on a floor): the
// table's shape is data here, and 64 rows (8 KB) still leave five workgroups per CU.
__global__ __launch_bounds__(256, RPT_SMALL_WAVES_PER_SIMD) void RPT_K(render_small_regen_table_kernel)(const SceneSmall sc, const RenderParams rp)
{
    __shared__ float4 s_rows[kMatTableRowsWide * kMatRowFloat4s];
    const SceneSmall& s = kernarg_scene(sc);
    render_regen_body(s, rp, material_table_build<false>(s, uniform_here(s.n_spheres), uniform_here(s.n_planes), s_rows));
}
// ... and for scenes of FIVE TO TWELVE primitives — every small scene there is — whose accepted sets fall into at most 16 classes of
// equal material (launch.h, MatClassMap): the same 64 rows, indexed by class.  The map is the kernel's third argument, read from the
// kernarg segment.
__global__ __launch_bounds__(256, RPT_SMALL_WAVES_PER_SIMD) void RPT_K(render_small_regen_maptable_kernel)(const SceneSmall sc, const RenderParams rp, const MatClassMap map)
{
    __shared__ float4 s_rows[kMatTableRowsWide * kMatRowFloat4s];
    __shared__ __attribute__((aligned(16))) uint8_t s_cls[4096];
    const SceneSmall& s = kernarg_scene(sc);
    // (the map where the launch put it: the kernarg segment is laid out like a struct of the arguments; read through a pointer, its
    //  per-lane reads are plain loads — indexed as a by-value argument the compiler would copy it to every lane's scratch)
    struct Args { SceneSmall sc; RenderParams rp; MatClassMap map; };
    (void)map;
    const MatClassMap& m = *(const MatClassMap*)((const char*)(const void*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Args, map));
    render_regen_body(s, rp, material_table_build_mapped(s, m, uniform_here(s.n_spheres), uniform_here(s.n_planes), s_rows, s_cls));
}
#ifndef RPT_RELAXED_BUILD                                           // (media have no relaxed form)
__global__ __launch_bounds__(256, RPT_SMALL_WAVES_PER_SIMD) void RPT_K(render_small_regen_media_kernel)(const WithMedia<SceneSmall> sc, const RenderParams rp) { render_regen_body(kernarg_scene(sc), rp); }
#endif

namespace RPT_LAUNCH_NS {

hipError_t render_small(const SceneSmall& sc, bool media, bool nested, const RenderParams& rp, uint32_t nblocks, hipStream_t st, const KernelChoice& kc)
{
    const dim3 tiles(nblocks), wg(256);
    (void)hipGetLastError();                                         // the thread's sticky error may be somebody else's (a host process's own HIP calls)
    if (media) {
#ifdef RPT_RELAXED_BUILD
        return hipErrorNotSupported;                                 // (media have no relaxed form)
#else
        if (nested) return hipErrorNotSupported;
        hipLaunchKernelGGL(RPT_K(render_small_regen_media_kernel), tiles, wg, 0, st, WithMedia<SceneSmall>(sc), rp);
#endif
    } else if (nested) hipLaunchKernelGGL(RPT_K(render_small_nested_kernel), tiles, wg, 0, st, sc, rp);
    else if (kc.sized && kc.material_table) hipLaunchKernelGGL(RPT_K(render_small_regen_sized_table_kernel), tiles, wg, kc.extra_lds, st, sc, rp);
    else if (kc.sized) hipLaunchKernelGGL(RPT_K(render_small_regen_sized_kernel), tiles, wg, kc.extra_lds, st, sc, rp);
    else if (kc.material_table || kc.material_table_wide) hipLaunchKernelGGL(RPT_K(render_small_regen_table_kernel), tiles, wg, kc.extra_lds, st, sc, rp);
    else if (kc.material_table_mapped) hipLaunchKernelGGL(RPT_K(render_small_regen_maptable_kernel), tiles, wg, kc.extra_lds, st, sc, rp, kc.class_map);
    else hipLaunchKernelGGL(RPT_K(render_small_regen_kernel), tiles, wg, kc.extra_lds, st, sc, rp);
    return hipGetLastError();
}

#ifndef RPT_RELAXED_BUILD
bool material_table_fits_small(const SceneSmallSdf& scs, bool has_sdf, uint32_t max_bits) { return material_table_fits(static_cast<const SceneSmall&>(scs), scs.sdf.material, has_sdf, max_bits); }
#ifdef RPT_PROFILE_BLOCKS
hipError_t prof_read_small(unsigned long long* out) { return prof_read(out); }
#endif
#endif

}  // namespace RPT_LAUNCH_NS
