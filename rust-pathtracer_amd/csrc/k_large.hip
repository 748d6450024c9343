// k_large.hip — scenes beyond the kernarg tables (BASELINE configs[4]: 10 k spheres, 16 lights): tables in HBM, uniform grid
// (dev_scene_large.h).  Built with the range tests next to every operation (kernel_common.h).
#include "kernel_common.h"

enum : uint32_t { ST_TRACE = 0u, ST_SHADE = 1u, ST_DONE = 2u, ST_FINISH = 3u, ST_BLOCKED = 4u };

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
        const uint32_t own = share_handed_out(s_count);             // (every lane of the wave: who still has samples to hand out)
        const uint64_t needy = __ballot(own < rp.spp);
        if (state == ST_FINISH || state == ST_BLOCKED) {
            // blend the finished sample into the running mean and start the next one (or retire); one site for
            // the paths that ended in TRACE (miss, emitter) and in SHADE (pdf <= 0, depth)
            RPT_PROF(PB_FINISH);
            if (!share_my_turn(s_count, q, s)) {
                state = ST_BLOCKED;                                 // an earlier sample of the pixel is still on its way: asked again every pass
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
        if (state == ST_TRACE) {
            RPT_PROF(PB_TRACE);
            state = path_trace_geom(sc, DirectQuery{}, p, g) ? ST_SHADE : ST_FINISH;
        }
        const uint64_t m_shade = __ballot(state == ST_SHADE);
        const uint64_t m_go = __ballot(state == ST_TRACE || state == ST_FINISH);
        // (a blocked lane is in neither vote; it keeps the loop alive: the lane it waited for may have blended in this very pass, after
        //  both looked at the count)
        if ((m_shade | m_go | __ballot(state == ST_BLOCKED)) == 0ull) break;
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

// Large scenes: same schedule; the scene tables are streamed from HBM (dev_scene_large.h).  5 waves per SIMD: 96 VGPRs, 12 of them
// spilled (44 B of scratch per lane).  With the two tiers of cell lists 5 / 6 / 7 waves run at 1 881 / 1 874 / 1 858 Msamples/s (10 k
// spheres, 2048^2 x 32 spp) — and move 0.32 / 42 / 77 GB through HBM per launch: at 6 and 7 waves (80 / 72 VGPRs, 53 / 65 spilled) the
// resident waves' scratch no longer fits the L2s (profiles/r3/c5_megakernel vs c5_megakernel_7waves).  (Before the tiers 7 waves were
// 1.5 % ahead: 5: 1 660, 6: 1 664, 7: 1 689; round 2, hipcc's divide, 2048^2 x 8: 4: 1 165, 5: 1 387, 6: 1 454, 7: 1 372, 8: 1 200.)
#ifndef RPT_LARGE_WAVES_PER_SIMD
#define RPT_LARGE_WAVES_PER_SIMD 5
#endif
__global__ __launch_bounds__(256, RPT_LARGE_WAVES_PER_SIMD) void RPT_K(render_large_regen_kernel)(const SceneLarge sc, const RenderParams rp) { render_regen_body_tf(sc, rp); }
#ifndef RPT_RELAXED_BUILD
__global__ __launch_bounds__(256, RPT_LARGE_WAVES_PER_SIMD) void RPT_K(render_large_regen_media_kernel)(const WithMedia<SceneLarge> sc, const RenderParams rp) { render_regen_body_tf(sc, rp); }
#endif

namespace RPT_LAUNCH_NS {

#if defined(RPT_PROFILE_BLOCKS) && !defined(RPT_RELAXED_BUILD)
hipError_t prof_read_large(unsigned long long* out) { return prof_read(out); }
#endif

hipError_t render_large(const SceneLarge& scl, bool media, const RenderParams& rp, uint32_t nblocks, hipStream_t st)
{
    const dim3 tiles(nblocks), wg(256);
    (void)hipGetLastError();
#ifdef RPT_RELAXED_BUILD
    if (media) return hipErrorNotSupported;
#else
    if (media) hipLaunchKernelGGL(RPT_K(render_large_regen_media_kernel), tiles, wg, 0, st, WithMedia<SceneLarge>(scl), rp);
    else
#endif
    hipLaunchKernelGGL(RPT_K(render_large_regen_kernel), tiles, wg, 0, st, scl, rp);
    return hipGetLastError();
}

}  // namespace RPT_LAUNCH_NS
