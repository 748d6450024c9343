// k_sdf.hip — scenes with the procedural SDF object (BASELINE configs[3]): the sphere march as a scheduling state of the lane.
// Built with the range tests next to every operation (kernel_common.h).
#include "kernel_common.h"

// SDF scenes, two rooms (dev_sdf_path.h, SdfDeferredQuery).  Per lane:
//   [MARCH_S: the parked shadow ray of the bounce just shaded] -> MARCH_P: the path ray -> WAIT -> one block: add the parked light
//   sample if its ray got through; finish closest_hit; miss / emitter / path over -> blend, the pixel's next sample; surface ->
//   material, light sample (parked), BSDF, next ray -> the marches again.
// Per wave each pass either marches (while at least `march_min_lanes` lanes are marching, or nobody waits) or runs the block
// for the lanes that wait.
// A lane whose pixel has no sample left takes samples of another pixel of its wave (kernel_common.h, share_next; `q`: the pixel a lane
// works for); a sample that is finished before its predecessor has been blended waits in S2_BLOCKED.
enum : uint32_t { S2_MARCH_S = 0u, S2_MARCH_P = 1u, S2_WAIT = 2u, S2_DONE = 3u, S2_BLOCKED = 4u };

template <class MS = MaterialPerHit, class S>
RPT_DEV void render_sdf_march2_body(const S& sc, const RenderParams& launch, const MS& materials = MS{})
{
    RPT_PROF_INIT();
    __shared__ FrameKey s_fkey[kMaxSppPerLaunchSdf];
    __shared__ float s_weight[kMaxSppPerLaunchSdf];
    __shared__ float4 s_acc[256];
    __shared__ float4 s_pix[256];
    __shared__ float4 s_march[256];                                 // a lane's march between passes: t, t_useful, steps (bit 31: hit), accepted
                                                                    // analytic primitives (of the path ray's march, also while the shadow ray is marched)
    float4* const s_sho = g_sdf_sho;                                // the parked shadow ray and light sample of each lane (dev_sdf_path.h);
    float4* const s_shd = g_sdf_shd;                                // gain.w: t_useful of the path ray's march while the shadow ray is marched first
    float4* const s_gain = g_sdf_gain;
    __shared__ uint32_t s_count[256];                               // share_*: each pixel's samples handed out and blended
    const uint32_t tid = threadIdx.x;
    share_init(s_count, false);                                     // (until the lane is known to have a pixel)
    RenderParams rp;                                                // this workgroup's unit of the launch
    if (!lane_setup(LaneTables{s_fkey, s_weight, s_acc, s_pix}, sc.max_depth, launch, rp)) return;
    share_init(s_count, true);

    uint32_t s = 0;
    uint32_t q = tid;                                               // the pixel this lane renders a sample of
    uint32_t state = S2_MARCH_P;
    PathRegs p;
    bool pending = false;                                           // a light sample is parked, its shadow ray not answered yet
    bool lit = false;                                               // ... answered: it got through
    bool ending = false;                                            // the path is over once the parked sample is resolved
    bool blend_only = false;                                        // the sample is complete: it waits for its turn to be blended
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
        RPT_PROF_ALIVE((uint32_t)__popcll(__ballot(state != S2_DONE)));
        if (__ballot(state == S2_BLOCKED) != 0ull) { if (state == S2_BLOCKED && share_my_turn(s_count, q, s)) state = S2_WAIT; }
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
                    if (march_step(sc.sdf, mo, m)) {
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
        } else {
          const uint32_t own_started = share_handed_out(s_count);              // (every lane of the wave: who still has samples to hand out)
          const uint64_t needy = __ballot(own_started < rp.spp);
          if (state == S2_WAIT) {
            RPT_PROF(PB_SHADE);
            if (pending) {                                          // last bounce's light sample: visible unless its march hit the object
                if (lit) { const float4 gn = s_gain[tid]; p.radiance = p.radiance + mk3(gn.x, gn.y, gn.z); }
                pending = false;
            }
            bool over = ending || blend_only;
            ending = false;
            if (!over) {
                GeomHit g;
                g.code = 0u;
                const float4 r = s_march[tid];                      // the finished march of the path's ray
                const SdfDeferredQuery query{{(rpt_f2u(r.z) & 0x80000000u) != 0u, r.x}, AnalyticPre{r.y, rpt_f2u(r.w)}};
                const uint32_t what = path_trace_geom_split(sc, query, p, g);
                if (what == 0u) { p.radiance = p.radiance + background(sc, p.ray) * p.throughput; over = true; }
                else if (what == 1u) over = true;
                else {
                    // (pending comes back through the parked ray: the query marks it in the slot's direction.w)
                    s_shd[tid].w = 1.0f;
                    over = path_shade_full(sc, query, p, g, nullptr, nullptr, materials);
                    pending = s_shd[tid].w == 0.0f;
                }
            }
            // what comes next for this lane: [the parked shadow ray] then the path's ray (or the end of the path)
            bool new_ray = !over;
            ending = pending && over;
            if (over && !pending) {                                 // blend, the next sample (or retire)
                RPT_PROF(PB_FINISH);
                blend_only = !share_my_turn(s_count, q, s);                    // an earlier sample of the pixel is still on its way
                if (blend_only) {
                    state = S2_BLOCKED;
                } else {
                    float4 acc = s_acc[q];
                    { const float4 c = s_pix[q]; sample_guard<true>(sc, p.radiance, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w)); }
                    blend(acc, p.radiance, s_weight[s]);
                    s_acc[q] = acc;
                    share_blended(s_count, q);
                    if (!share_next(s_count, rp.spp, own_started, needy, q, s)) {
                        state = S2_DONE;
                    } else {
                        const float4 c = s_pix[q];
                        path_begin<true>(sc, p, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w));
                        new_ray = true;
                    }
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
    }
    RPT_PROF_FLUSH();
    lane_finish(rp, s_acc[tid]);
}

__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD)
void RPT_K(render_sdf_march2_kernel)(const SceneSmallSdf sc, const RenderParams rp) { render_sdf_march2_body(kernarg_scene(sc), rp); }
#ifndef RPT_RELAXED_BUILD
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
__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD) void RPT_K(render_sdf_march2_media_kernel)(const WithMedia<SceneSmallSdf> sc, const RenderParams rp) { render_sdf_march2_body(kernarg_scene(sc), rp); }
#endif

namespace RPT_LAUNCH_NS {

#ifndef RPT_RELAXED_BUILD
uint32_t max_spp_per_launch(bool sdf_object) { return sdf_object ? kMaxSppPerLaunchSdf : kMaxSppPerLaunch; }
#ifdef RPT_PROFILE_BLOCKS
hipError_t prof_read_sdf(unsigned long long* out) { return prof_read(out); }
#endif
#endif

hipError_t render_sdf(const SceneSmallSdf& scs, bool media, const RenderParams& rp, uint32_t nblocks, hipStream_t st, const KernelChoice& kc)
{
    const dim3 tiles(nblocks), wg(256);
    (void)hipGetLastError();
#ifdef RPT_RELAXED_BUILD
    (void)kc;
    if (media) return hipErrorNotSupported;
#else
    if (media) hipLaunchKernelGGL(RPT_K(render_sdf_march2_media_kernel), tiles, wg, 0, st, WithMedia<SceneSmallSdf>(scs), rp);
    else if (kc.sized_sdf == 1u && kc.material_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<1u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 2u && kc.material_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<2u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 3u && kc.material_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<3u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 4u && kc.material_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<4u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 1u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<1u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 2u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<2u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 3u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<3u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 4u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<4u>, tiles, wg, 0, st, scs, rp);
    else
#endif
    hipLaunchKernelGGL(RPT_K(render_sdf_march2_kernel), tiles, wg, 0, st, scs, rp);
    return hipGetLastError();
}

}  // namespace RPT_LAUNCH_NS
