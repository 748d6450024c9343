// Resumable sphere march for scenes with the SDF object (include/rpt.h rpt_sdf).
//
// In dev_integrator.h the march runs inside closest_hit / any_hit: a wave leaves it only when its
// slowest lane has (measured on BASELINE.json configs[3]: 34 % lane utilisation, march lengths range
// from 2 to `max_steps`).  Here the march is a scheduling state of the lane, like TRACE and SHADE in
// the regeneration kernel: a lane marches in chunks, and when its march ends it waits for enough
// lanes with the same next stage, while the lanes still marching are joined by lanes that started a
// new march (the next bounce, the next sample of the pixel, a shadow ray).
//
// Nothing about the arithmetic changes: the march below is sdf_march() one iteration at a time, and
// its outcome is handed to the same closest_geom_small / any_hit_small code through SdfMarchResult,
// so images are bit-identical to the bounce-granular kernels (tests/test_gpu_parity.py, SDF cases).
#pragma once
#include "dev_integrator.h"

namespace rptdev {

// One lane's march in flight.  The ray origin is p.ray.o for both kinds: after RESOLVE the path's own
// origin is dead (the next origin is rebuilt from the parked hit point), so a shadow march borrows it.
struct MarchRegs {
    v3 d;                      // direction being marched (the path's for a closest_hit march, the light's for a shadow march)
    float t, t_useful;
    uint32_t steps;
    uint32_t accepted;         // closest_hit march: the analytic primitives accepted before the march (analytic_closest)
    bool hit;
};

RPT_DEV void march_begin(MarchRegs& m, v3 dir, float t_useful)
{
    m.d = dir;
    m.t = 0.0f;
    m.t_useful = t_useful;
    m.steps = 0;
    m.hit = false;
}

// One iteration of sdf_march()'s loop; true when the march is over (m.hit / m.t hold its result).  `eval`: the object — a DevSdf, or
// its records in registers (SdfRegs).
template <class E>
RPT_DEV bool march_step(const DevSdf& sd, const E& eval, v3 origin, MarchRegs& m)
{
    if (m.steps >= sd.max_steps) return true;
    float dist = sdf_eval(eval, origin + m.t * m.d);
    if (dist < sd.hit_eps * m.t) { m.hit = true; return true; }
    m.t = m.t + dist;
    m.steps += 1;
    return (m.t > sd.max_t) || (m.t > m.t_useful);
}
RPT_DEV bool march_step(const DevSdf& sd, v3 origin, MarchRegs& m) { return march_step(sd, sd, origin, m); }

// Scene queries answered from a finished march.
struct SdfInjectedQuery {
    SdfMarchResult r;
    AnalyticPre a;             // closest_hit: what march_begin_primary found among the analytic primitives
    RPT_DEV bool geom(const SceneSmallSdf& sc, const RayD& ray, PathState& ps, GeomHit& g, EmitterHit& e) const
    {
        return closest_geom_small<true>(sc, &sc.sdf, ray, ps, g, e, &r, &a);
    }
    RPT_DEV bool any(const SceneSmallSdf& sc, const RayD& ray, float max_dist, v3, bool& pending) const
    {
        pending = false;
        return any_hit_small<true>(sc, &sc.sdf, ray, max_dist, &r);
    }
    RPT_DEV void park(v3) const {}
};

// Start the closest_hit march of the path's current ray.
RPT_DEV void march_begin_primary(const SceneSmallSdf& sc, const PathRegs& p, MarchRegs& m)
{
    AnalyticHit a;
    analytic_closest(sc, p.ray, a);
    march_begin(m, p.ray.d, sdf_primary_t_useful(sc, a));
    m.accepted = a.accepted;
}

// The analytic part of closest_hit as march_begin_primary left it.  (With no analytic primitive t_useful is +inf
// where analytic_closest says F::MAX; closest_geom_small does not read the distance in that case.)
RPT_DEV AnalyticPre march_analytic(const MarchRegs& m) { return AnalyticPre{m.t_useful, m.accepted}; }

// After a surface hit: find out whether next-event estimation will march a shadow ray, without
// consuming the path's random numbers (SHADE replays the same draws from p.rng).  Returns true when a
// march was started (origin in p.ray.o); false when any_hit's answer does not depend on the SDF object,
// in which case m.hit is set to what the march would be allowed to report (nothing).
// OFFSET false (media kernels): `fhp` is a scatter point inside a medium, the shadow ray starts exactly there.
template <bool OFFSET = true>
RPT_DEV bool march_begin_shadow(const SceneSmallSdf& sc, PathRegs& p, v3 fhp, v3 ffnormal, MarchRegs& m)
{
    m.hit = false;
    m.t = 0.0f;
    if (sc.n_lights == 0) return false;
    Rng rng = p.rng;
    v3 scatter_pos;
    float light_area;
    LightSample ls;
    if (!nee_sample<OFFSET>(sc, fhp, ffnormal, rng, scatter_pos, light_area, ls)) return false;
    const float max_dist = ls.dist - sc.eps;
    const RayD shadow{scatter_pos, ls.direction};
    if (any_hit_analytic(sc, shadow, max_dist)) return false;      // occluded whatever the march says
    p.ray.o = scatter_pos;
    march_begin(m, ls.direction, sdf_shadow_t_useful(sc, max_dist));
    return true;
}

// ---- two rooms instead of three (render_sdf_march2_kernel) -------------------------------------------------------------
// The shadow ray of next-event estimation is marched one bounce LATE, right before the next path ray, the way the wavefront
// form of large scenes walked its shadow rays (rounds 2-4): SHADE computes the light sample's contribution as if the
// light were visible and parks it; once the shadow march has answered, it is added — or not — before anything else touches
// the radiance, so the additions and their order are the reference's.  A bounce is then ONE block (finish closest_hit,
// shade, set up the next ray) between marches, not RESOLVE -> march -> SHADE: two waiting rooms for a 64-lane wave instead of
// three, and next-event estimation is sampled once instead of twice (the three-room kernel replays it to find the shadow ray).
// Each lane's parked shadow ray (o.w: max_dist; d.w: 0 once a ray is parked) and light sample.  File-scope LDS: a query object
// that carried three pointers to its lane's slots kept six VGPRs alive through the whole shading block.
__shared__ float4 g_sdf_sho[256], g_sdf_shd[256], g_sdf_gain[256];

struct SdfDeferredQuery {
    SdfMarchResult r;          // the path ray's march
    AnalyticPre a;
    RPT_DEV bool geom(const SceneSmallSdf& sc, const RayD& ray, PathState& ps, GeomHit& g, EmitterHit& e) const
    {
        return closest_geom_small<true>(sc, &sc.sdf, ray, ps, g, e, &r, &a);
    }
    // any_hit_small<true> taken apart: the analytic primitives now, the SDF object's march later
    RPT_DEV bool any(const SceneSmallSdf& sc, const RayD& shadow, float max_dist, v3 th, bool& pending) const
    {
        pending = false;
        if (any_hit_analytic(sc, shadow, max_dist)) return true;
        const bool finite = (__builtin_fabsf(th.x) < __builtin_inff()) && (__builtin_fabsf(th.y) < __builtin_inff()) &&
                            (__builtin_fabsf(th.z) < __builtin_inff());
        if (!finite) {
            // "add nothing" and "add 0 x throughput" differ for this lane: answer on the spot (rare)
            const bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
            float t;
            const bool h = sdf_march(sc.sdf, shadow, sdf_shadow_t_useful(sc, max_dist), t);
            return h && (!use_max || t < max_dist);
        }
        pending = true;
        g_sdf_sho[threadIdx.x] = make_float4(shadow.o.x, shadow.o.y, shadow.o.z, max_dist);
        g_sdf_shd[threadIdx.x] = make_float4(shadow.d.x, shadow.d.y, shadow.d.z, 0.0f);
        return false;
    }
    RPT_DEV void park(v3 c) const { g_sdf_gain[threadIdx.x] = make_float4(c.x, c.y, c.z, 0.0f); }
};

}  // namespace rptdev
