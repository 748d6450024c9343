// k_sdf.hip — scenes with the procedural SDF object (BASELINE configs[3]): the sphere march as a scheduling state of the lane.
// Built with the range tests next to every operation (kernel_common.h).
#include "kernel_common.h"

// SDF scenes (dev_sdf_path.h, SdfDeferredQuery).  Per lane:
//   [MARCH_S: the parked shadow ray of the bounce just shaded] -> MARCH_P: the path ray -> WAIT -> first block: add the parked light
//   sample if its ray got through; finish closest_hit; miss / emitter / path over -> blend, the pixel's next sample -> the marches
//   again; surface -> SHADE (the second room) -> second block: material, light sample (parked), BSDF, next ray -> the marches again.
// Per wave each pass either marches (while at least `march_min_lanes` lanes are marching, or nobody waits) or runs ONE of the two
// blocks between marches for the lanes that wait for it (round 6: below, S2_SHADE).
// A lane whose pixel has no sample left takes samples of another pixel of its wave (kernel_common.h, share_next; `q`: the pixel a lane
// works for); a sample that is finished before its predecessor has been blended waits in S2_BLOCKED.
enum : uint32_t { S2_MARCH_S = 0u, S2_MARCH_P = 1u, S2_WAIT = 2u, S2_DONE = 3u, S2_BLOCKED = 4u, S2_SHADE = 5u };
// Round 6: the block is cut behind closest_hit's acceptance.  A lane that hit a SURFACE waits in a second room (S2_SHADE, one dword
// parked: GeomHit) until rp.shade_threshold lanes do, or nobody marches or waits for the first part; the first part — the parked light
// sample, closest_hit's acceptance, and for a path that is over the background, the blend and the pixel's next camera path — runs as
// before.  A block used to mix both kinds (shading ran with 54 % of the lanes, finishing with 43 %); lanes that wait for the second room
// are idle lanes of the march phases, i.e. helpers.  (Threshold 1 is round 5's one block cut in two passes.)
// A lane's two marches between passes.  Its shadow march (the lane's own work always): s_march = {t, t_useful, steps, -}.  Its path
// march, which ANY lane of the wave may run: t in the parked ray's direction.w, t_useful in the parked gain's .w, s_pjob = steps | hit | over;
// s_march.w: the analytic primitives accepted before it.
constexpr uint32_t kPjobSteps = 0x1FFFFu, kPjobHit = 1u << 30, kPjobDone = 1u << 31;

template <uint32_t NPRIMS = 0u, class MS = MaterialPerHit, class S>
RPT_DEV void render_sdf_march2_body(const S& sc, const RenderParams& launch, const MS& materials = MS{})
{
    RPT_PROF_INIT();
    __shared__ FrameKey s_fkey[kMaxSppPerLaunchSdf];
    __shared__ float s_weight[kMaxSppPerLaunchSdf];
    __shared__ float4 s_acc[256];
    __shared__ float4 s_pix[256];
    __shared__ float4 s_march[256];
    float4* const s_sho = g_sdf_sho;                                // the parked shadow ray and light sample of each lane (dev_sdf_path.h);
    float4* const s_shd = g_sdf_shd;
    float4* const s_gain = g_sdf_gain;
    __shared__ uint32_t s_count[256];                               // share_*: each pixel's samples handed out and blended
    __shared__ uint32_t s_pjob[256];                                // kPjob*
    const uint32_t tid = threadIdx.x;
    share_init(s_count, false);                                     // (until the lane is known to have a pixel)
    RenderParams rp;                                                // this workgroup's unit of the launch
    // A lane of a ragged tile that has no pixel STAYS in its wave, as S2_DONE from the start: it owns no sample and shares none
    // (its count says "none left"), but it is an idle lane like any other for the marches handed from lane to lane below — and the
    // pairing there ranks lanes with ballots and moves the offers through lane registers, which is only right while every lane of the
    // wave is alive (a ds_bpermute that selects a lane that has left the kernel returns 0, i.e. lane 0's job: ADVICE r5).
    const uint32_t have = lane_setup_ex(LaneTables{s_fkey, s_weight, s_acc, s_pix}, sc.max_depth, launch, rp);
    if (have == LANE_NOTHING) return;
    const bool has_pixel = have == LANE_PIXEL;
    if (has_pixel) share_init(s_count, true);

    uint32_t s = 0;
    uint32_t q = tid;                                               // the pixel this lane renders a sample of
    uint32_t state = has_pixel ? S2_MARCH_P : S2_DONE;
    PathRegs p;
    GeomHit g_hit;                                                  // what closest_hit's acceptance found: the dword a lane waiting for the second room parks
    g_hit.code = 0u;
    bool pending = false;                                           // a light sample is parked, its shadow ray not answered yet
    bool lit = false;                                               // ... answered: it got through
    bool ending = false;                                            // the path is over once the parked sample is resolved
    bool blend_only = false;                                        // the sample is complete: it waits for its turn to be blended
    bool p_done = false;                                            // this lane's path march is over (its own work or a helper's)
    if (has_pixel) {
        const float4 c = s_pix[tid];
        path_begin<true>(sc, p, c.x, c.y, s_fkey[0], rpt_f2u(c.z), rpt_f2u(c.w));
        MarchRegs m;
        march_begin_primary(sc, p, m);
        s_march[tid] = make_float4(0.0f, 0.0f, rpt_u2f(0u), rpt_u2f(m.accepted));
        s_shd[tid].w = 0.0f; s_gain[tid].w = m.t_useful; s_pjob[tid] = 0u;
    } else {
        p.ray.o = mk3(0.0f, 0.0f, 0.0f); p.ray.d = mk3(0.0f, 0.0f, 0.0f);     // (read by nobody: a lane without a pixel never offers a march)
    }
    const bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;

    for (;;) {
        RPT_PROF(PB_PASS);
        RPT_PROF_ALIVE((uint32_t)__popcll(__ballot(state != S2_DONE)));
        if (__ballot(state == S2_BLOCKED) != 0ull) { if (state == S2_BLOCKED && share_my_turn(s_count, q, s)) state = S2_WAIT; }
        const uint32_t n_march = (uint32_t)__popcll(__ballot(state <= S2_MARCH_P));
        uint32_t n_wait = (uint32_t)__popcll(__ballot(state == S2_WAIT));
        const uint32_t shade_room = rp.shade_threshold;             // (>= 1: capi.hip)
        bool run_shade_room = false;
        {
            const uint32_t n_shade = (uint32_t)__popcll(__ballot(state == S2_SHADE));
            if (n_march == 0u && n_wait == 0u && n_shade == 0u) break;
            // the second room runs when it is full enough, or when nobody marches and nobody waits for the first part
            run_shade_room = n_shade >= shade_room || (n_shade != 0u && n_wait == 0u && n_march < rp.march_min_lanes);
            if (run_shade_room) n_wait = n_shade;                   // (the phase decision below: "somebody waits for a block")
        }
        const uint32_t own_started = share_handed_out(s_count);     // (every lane of the wave: who still has samples to hand out)
        const uint64_t needy = __ballot(own_started < rp.spp);
        if (!run_shade_room && (n_march >= rp.march_min_lanes || n_wait == 0u)) {
            // Nothing of a march is live in registers across the block: a marching lane takes its march from LDS here and puts it
            // back behind the loop (direction and origin are the path's ray or the parked shadow ray).
            // A lane that marches its shadow ray has a SECOND march waiting behind it, its path ray's.  Lanes with nothing to march
            // (their marches are over, they wait for the block; or they are done) take such marches over: every kHelpEvery steps idle
            // lanes and waiting path marches are paired by rank, the helper fetches the ray from its owner's registers (ds_bpermute)
            // and the march's state from the owner's slots, and leaves the outcome — or, when the phase ends first, how far it got — there.
            // A march is a function of its ray alone: who runs it, and in how many pieces, changes nothing.
#ifndef RPT_SDF_HELP_EVERY
#define RPT_SDF_HELP_EVERY 4
#endif
#ifndef RPT_SDF_HELP_MIN_IDLE
#define RPT_SDF_HELP_MIN_IDLE 1
#endif
            constexpr uint32_t kHelpEvery = RPT_SDF_HELP_EVERY, kHelpMinIdle = RPT_SDF_HELP_MIN_IDLE;
            const uint32_t lane = tid & 63u, base = tid & ~63u;
            uint32_t work = 0u;                                     // 0 idle, 1 the lane's shadow march, 2 a path march: lane `job`'s
            uint32_t job = tid;
            bool p_given = false;                                   // this lane's waiting path march is with a helper (for this phase)
            bool s_over = false;                                    // this lane's shadow march ended in this phase
            MarchRegs m;
            v3 mo = mk3(0.0f, 0.0f, 0.0f);
            m.d = mk3(0.0f, 0.0f, 0.0f); m.t = 0.0f; m.t_useful = 0.0f; m.steps = 0u; m.accepted = 0u; m.hit = false;
            if (state == S2_MARCH_S) {
                const float4 r = s_march[tid], so = s_sho[tid], sd = s_shd[tid];
                m.t = r.x; m.t_useful = r.y; m.steps = rpt_f2u(r.z);
                mo = mk3(so.x, so.y, so.z); m.d = mk3(sd.x, sd.y, sd.z);
                work = 1u;
            } else if (state == S2_MARCH_P) {
                mo = p.ray.o; m.d = p.ray.d;
                m.t = s_shd[tid].w; m.t_useful = s_gain[tid].w; m.steps = s_pjob[tid] & kPjobSteps;
                work = 2u;
            }
            // (a kernel that knows the number of primitives keeps their records in scalar registers for the phase)
            SdfRegs<(NPRIMS != 0u ? NPRIMS : 1u)> sdf_regs;
            if constexpr (NPRIMS != 0u) sdf_regs_load(sc.sdf, sdf_regs);
            for (uint32_t step = 0u;; ++step) {
                if ((step & (kHelpEvery - 1u)) == 0u) {
                    const bool offers = work == 1u && !ending && !p_given && !p_done;      // (a path march nobody has started behind a shadow march in flight)
                    const uint64_t D = __ballot(offers), I = __ballot(work == 0u);
                    if (D != 0ull && (uint32_t)__popcll(I) >= kHelpMinIdle) {
                        const uint32_t nD = (uint32_t)__popcll(D), nI = (uint32_t)__popcll(I);
                        const uint32_t dr = __builtin_amdgcn_mbcnt_hi((uint32_t)(D >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)D, 0u));
                        const uint32_t ir = __builtin_amdgcn_mbcnt_hi((uint32_t)(I >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)I, 0u));
                        // lane k receives the lane of the k-th offer (the others push theirs to lane 63, which then is no offer's rank)
                        const int tbl = __builtin_amdgcn_ds_permute((int)((offers ? dr : 63u) * 4u), (int)lane);
                        const uint32_t src = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ir * 4u), tbl) & 63u;
                        const float ox = rpt_u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(src * 4u), (int)rpt_f2u(p.ray.o.x)));
                        const float oy = rpt_u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(src * 4u), (int)rpt_f2u(p.ray.o.y)));
                        const float oz = rpt_u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(src * 4u), (int)rpt_f2u(p.ray.o.z)));
                        const float dx = rpt_u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(src * 4u), (int)rpt_f2u(p.ray.d.x)));
                        const float dy = rpt_u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(src * 4u), (int)rpt_f2u(p.ray.d.y)));
                        const float dz = rpt_u2f((uint32_t)__builtin_amdgcn_ds_bpermute((int)(src * 4u), (int)rpt_f2u(p.ray.d.z)));
                        if (work == 0u && ir < nD) {
                            job = base + src;
                            mo = mk3(ox, oy, oz); m.d = mk3(dx, dy, dz);
                            m.t = s_shd[job].w; m.t_useful = s_gain[job].w; m.steps = s_pjob[job] & kPjobSteps; m.hit = false;
                            work = 2u;
                        }
                        if (offers && dr < nI) p_given = true;
                    }
                }
                if (work != 0u) {
                    RPT_PROF(PB_CLOSEST);                           // (block profile: one march step of the wave)
                    bool over;
                    if constexpr (NPRIMS != 0u) over = march_step(sc.sdf, sdf_regs, mo, m);
                    else over = march_step(sc.sdf, mo, m);
                    if (over) {
                        if (work == 1u) {
                            lit = !(m.hit && (!use_max || m.t < s_sho[tid].w));      // any_hit_small's SDF term
                            s_over = true;
                            work = 0u;
                            if (!ending && !p_given && !p_done) {
                                // the path ray's march, prepared by the block (march_begin_primary's analytic part is ~400
                                // instructions: it must not run here, for the one lane of the wave whose shadow march just ended);
                                // from where a helper of an earlier phase left it
                                job = tid;
                                mo = p.ray.o; m.d = p.ray.d;
                                m.t = s_shd[tid].w; m.t_useful = s_gain[tid].w; m.steps = s_pjob[tid] & kPjobSteps; m.hit = false;
                                work = 2u;
                            }
                        } else {
                            s_shd[job].w = m.t; s_pjob[job] = m.steps | (m.hit ? kPjobHit : 0u) | kPjobDone;
                            work = 0u;
                        }
                    }
                }
                const uint32_t left = (uint32_t)__popcll(__ballot(work != 0u));
                if (left == 0u) break;
                // Few lanes still march: the phase ends for the block — if somebody is waiting for it.  Lanes that waited when the
                // phase began, or a marcher whose marches have ended since (not one whose path march is out with a helper).  With
                // nobody to run the block for, leaving would only re-enter this phase at its full price (the reload of every march,
                // the records' scalar loads, the pairing) once per step: the wave's tail marches on instead.
                if (left < rp.march_min_lanes && (n_wait != 0u || __ballot(state <= S2_MARCH_P && work == 0u && !p_given) != 0ull)) break;
            }
            // what is unfinished goes back to its slot; then every lane learns where its own marches stand
            if (work == 1u) { s_march[tid].x = m.t; s_march[tid].z = rpt_u2f(m.steps); }
            if (work == 2u) { s_shd[job].w = m.t; s_pjob[job] = m.steps; }
            if (state <= S2_MARCH_P) {
                p_done = (s_pjob[tid] & kPjobDone) != 0u;
                if (state == S2_MARCH_P || s_over) state = (p_done || ending) ? S2_WAIT : S2_MARCH_P;
            }
        } else if (state == (run_shade_room ? S2_SHADE : S2_WAIT)) {
            RPT_PROF(PB_SHADE);
            bool over = false, shade_now = state == S2_SHADE, to_room = false;
            // the finished march of the path's ray
            const SdfDeferredQuery query{{(s_pjob[tid] & kPjobHit) != 0u, s_shd[tid].w}, AnalyticPre{s_gain[tid].w, rpt_f2u(s_march[tid].w)}};
            if (!shade_now) {
                if (pending) {                                      // last bounce's light sample: visible unless its march hit the object
                    if (lit) { const float4 gn = s_gain[tid]; p.radiance = p.radiance + mk3(gn.x, gn.y, gn.z); }
                    pending = false;
                }
                over = ending || blend_only;
                ending = false;
                if (!over) {
                    g_hit.code = 0u;
                    const uint32_t what = path_trace_geom_split(sc, query, p, g_hit);
                    if (what == 0u) { p.radiance = p.radiance + background(sc, p.ray) * p.throughput; over = true; }
                    else if (what == 1u) over = true;
                    else { state = S2_SHADE; to_room = true; }      // the surface waits for its room
                }
            }
            if (shade_now) {
                // (pending comes back through the parked ray: the query marks it in the slot's direction.w)
                s_shd[tid].w = 1.0f;
                over = path_shade_full(sc, query, p, g_hit, nullptr, nullptr, materials);
                pending = s_shd[tid].w == 0.0f;
            }
            if (!to_room) {
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
            if (new_ray) {
                s_march[tid].w = rpt_u2f(np_acc);
                s_shd[tid].w = 0.0f; s_gain[tid].w = np_tu; s_pjob[tid] = 0u;
                p_done = false;
                state = S2_MARCH_P;
            }
            if (pending) {
                s_march[tid].x = 0.0f; s_march[tid].y = sdf_shadow_t_useful(sc, s_sho[tid].w); s_march[tid].z = rpt_u2f(0u);
                state = S2_MARCH_S;
            }
            }
        }
    }
    RPT_PROF_FLUSH();
    if (!has_pixel) return;                                         // (before lane_finish: the hand-off counts the lanes that stored a pixel)
    lane_finish(rp, s_acc[tid]);
}

__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD)
void RPT_K(render_sdf_march2_kernel)(const SceneSmallSdf sc, const RenderParams rp) { render_sdf_march2_body(kernarg_scene(sc), rp); }
template <uint32_t NPRIMS>
__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD)
void RPT_K(render_sdf_march2_sized_kernel)(const SceneSmallSdf sc, const RenderParams rp) { render_sdf_march2_body<NPRIMS>(sized_sdf_scene<NPRIMS>(kernarg_scene(sc)), rp); }
// ... with the material table (dev_integrator.h, MaterialTable): at most one analytical sphere beside the plane and the object
template <uint32_t NPRIMS>
__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD)
void RPT_K(render_sdf_march2_sized_table_kernel)(const SceneSmallSdf sc, const RenderParams rp)
{
    __shared__ float4 s_rows[kMatTableRows * kMatRowFloat4s];
    const SceneSmallSdf& s = sized_sdf_scene<NPRIMS>(kernarg_scene(sc));
    render_sdf_march2_body<NPRIMS>(s, rp, material_table_build<true>(s, s.n_spheres, 1u, s_rows));
}
// ... for any scene of at most three primitives, the object included: the table's shape is data
__global__ __launch_bounds__(256, RPT_SDF_WAVES_PER_SIMD)
void RPT_K(render_sdf_march2_table_kernel)(const SceneSmallSdf sc, const RenderParams rp)
{
    __shared__ float4 s_rows[kMatTableRows * kMatRowFloat4s];
    const SceneSmallSdf& s = kernarg_scene(sc);
    render_sdf_march2_body(s, rp, material_table_build<true>(s, uniform_here(s.n_spheres), uniform_here(s.n_planes), s_rows));
}
#ifndef RPT_RELAXED_BUILD                                           // (media have no relaxed form)
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
    if (media) {
#ifdef RPT_RELAXED_BUILD
        return hipErrorNotSupported;
#else
        hipLaunchKernelGGL(RPT_K(render_sdf_march2_media_kernel), tiles, wg, 0, st, WithMedia<SceneSmallSdf>(scs), rp);
#endif
    } else if (kc.sized_sdf == 1u && kc.material_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<1u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 2u && kc.material_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<2u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 3u && kc.material_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<3u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 4u && kc.material_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_table_kernel)<4u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 1u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<1u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 2u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<2u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 3u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<3u>, tiles, wg, 0, st, scs, rp);
    else if (kc.sized_sdf == 4u) hipLaunchKernelGGL(RPT_K(render_sdf_march2_sized_kernel)<4u>, tiles, wg, 0, st, scs, rp);
    else if (kc.material_table) hipLaunchKernelGGL(RPT_K(render_sdf_march2_table_kernel), tiles, wg, 0, st, scs, rp);
    else hipLaunchKernelGGL(RPT_K(render_sdf_march2_kernel), tiles, wg, 0, st, scs, rp);
    return hipGetLastError();
}

}  // namespace RPT_LAUNCH_NS
