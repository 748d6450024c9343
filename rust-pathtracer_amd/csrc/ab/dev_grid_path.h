// Resumable grid walk for large analytical scenes (BASELINE.json configs[4]: 10 k spheres, 16 lights).
//
// In dev_scene_large.h the 3D-DDA walk runs inside closest_hit / any_hit: a wave leaves it when its longest walk
// ends (measured: 85 % of the kernel's time is in the two walks, at 23 % VALU lane utilisation; walks cross 1 to
// ~40 cells and the cells hold 0 to ~6 spheres).  Here the walk is a scheduling state of the lane, like the sphere
// march of dev_sdf_path.h: a lane walks one cell per step, and when its walk ends it waits for enough lanes with the
// same next stage, while the lanes still walking are joined by lanes that started a new walk (next bounce, next
// sample, a shadow ray).
//
// MEASURED SLOWER than the walk inside the bounce (10 k spheres, 2048^2 x 8 spp: 1.12 vs 1.40 Gsamples/s, best of
// min-lanes 1..48 and 3..5 waves/SIMD): the walk is bound by the latency of its dependent loads, which lane
// utilisation does not shorten, and the walk state costs registers (224 B of scratch at 5 waves/SIMD).  The kernel is
// kept behind RPT_RENDER_GRID_RESUMABLE_WALK for A/B; the default is the bounce-granular kernel.
//
// Nothing about the arithmetic changes: walk_step() is one iteration of grid_closest_sphere's / grid_any_sphere's
// loop, and the outcome is handed to the unchanged plane / light / shading code through GridInjectedQuery, so images
// are bit-identical to the bounce-granular kernel (tests/test_gpu_parity.py, large-scene cases).
#pragma once
#include "../dev_scene_large.h"

namespace rptdev {

// One lane's walk in flight.  `g` is the cell whose sphere list [k0, k1) has not been tested yet.
struct WalkRegs {
    RayD ray;                  // the ray being walked: the path's (closest hit) or the shadow ray (any hit)
    GridWalk g;
    uint32_t k0, k1;
    uint32_t guard;            // a DDA crosses at most nx+ny+nz cells: every walk ends
    float dist;                // closest: nearest hit so far; any: max_dist
    uint32_t best;             // closest: its sphere
    bool hit;                  // closest: anything hit; any: occluded
};

// Start grid_closest_sphere(); true when the answer is already known (w.dist / best / hit), false when w has to walk.
RPT_DEV bool walk_begin_closest(const SceneLarge& sc, const RayD& ray, WalkRegs& w)
{
    w.ray = ray;
    w.dist = 3.40282347e+38f;
    w.best = 0xFFFFFFFFu;
    w.hit = false;
    if (!grid_usable(sc, ray)) { brute_closest_sphere(sc, ray, w.dist, w.best, w.hit); return true; }
    {   // sphere 0: accepted whenever it is hit (analytical.rs:43)
        const float4 s = sphere_uniform(sc, 0);
        float t;
        if (hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t)) { w.dist = t; w.best = 0; w.hit = true; }
    }
    w.g = grid_begin(sc, ray);
    if (!w.g.alive) return true;
    const uint32_t c = grid_cell_index(sc, w.g);
    w.k0 = sc.cell_start[c];
    w.k1 = sc.cell_start[c + 1];
    w.guard = sc.gn[0] + sc.gn[1] + sc.gn[2] + 3u;
    return false;
}

// Start grid_any_sphere(); as above (w.hit = occluded).
RPT_DEV bool walk_begin_any(const SceneLarge& sc, const RayD& ray, float max_dist, WalkRegs& w)
{
    const bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
    w.ray = ray;
    w.dist = max_dist;
    w.hit = false;
    if (!grid_usable(sc, ray)) { w.hit = brute_any_sphere(sc, ray, use_max, max_dist); return true; }
    w.g = grid_begin(sc, ray);
    if (!w.g.alive) return true;
    const uint32_t c = grid_cell_index(sc, w.g);
    w.k0 = sc.cell_start[c];
    w.k1 = sc.cell_start[c + 1];
    w.guard = sc.gn[0] + sc.gn[1] + sc.gn[2] + 3u;
    return false;
}

// One iteration of the loop of grid_closest_sphere (`any` false) / grid_any_sphere (`any` true); true when the walk
// is over.  `any` is a per-lane value: lanes on a primary walk and lanes on a shadow walk step together and share
// the sphere tests.
RPT_DEV bool walk_step(const SceneLarge& sc, WalkRegs& w, bool any)
{
    const bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
    const float t_exit = grid_cell_exit(w.g);                       // of the cell whose list is [k0, k1)
    grid_step(sc, w.g);                                             // g is the NEXT cell from here on
    uint32_t n0 = 0, n1 = 0;
    if (w.g.alive) { const uint32_t c = grid_cell_index(sc, w.g); n0 = sc.cell_start[c]; n1 = sc.cell_start[c + 1]; }
    bool occluded = false;
    for (uint32_t k = w.k0; k < w.k1 && !occluded; ++k) {
        const float4 s = sc.cell_spheres[k];
        float t;
        if (hit_sphere(w.ray, mk3(s.x, s.y, s.z), s.w, t)) {
            if (any) {
                occluded = (!use_max || t < w.dist);
            } else {
                const uint32_t i = sc.cell_items[k];
                if (i != 0u && (t < w.dist || (t == w.dist && i < w.best))) { w.dist = t; w.best = i; w.hit = true; }
            }
        }
    }
    if (occluded) { w.hit = true; return true; }
    if (!any && w.hit && w.dist <= t_exit) return true;             // nothing beyond this cell can be nearer
    if (t_exit > w.g.t_end) return true;
    if (any && use_max && t_exit > w.dist) return true;             // a sphere entirely beyond max_dist cannot occlude
    w.k0 = n0; w.k1 = n1;
    w.guard -= 1u;
    return !(w.g.alive && w.guard != 0u);
}

// Scene queries answered from finished walks.
struct GridInjectedQuery {
    float dist;
    uint32_t best;
    bool hit;
    bool occluded;
    RPT_DEV bool geom(const SceneLarge& sc, const RayD& ray, PathState& ps, GeomHit& g, EmitterHit& e) const
    {
        return closest_geom_finish(sc, ray, ps, dist, best, hit, g, e);
    }
    RPT_DEV bool any(const SceneLarge& sc, const RayD& ray, float max_dist) const { return any_hit_finish(sc, ray, max_dist, occluded); }
};

// After a surface hit: find out whether next-event estimation will walk a shadow ray, without consuming the path's
// random numbers (SHADE replays the same draws from p.rng).  Returns true when a walk was started; false when
// any_hit's sphere part is already known (w.hit) or will not be asked (the light sample does not face the point).
RPT_DEV bool walk_begin_shadow(const SceneLarge& sc, const PathRegs& p, const GeomHit& g, WalkRegs& w)
{
    w.hit = false;
    if (sc.n_lights == 0) return false;
    const v3 n = normal_large(sc, p.ray, p.ps.hit_dist, g);
    const bool front = (dot3(n, p.ray.d) <= 0.0f);                   // State::finalize, globals.rs:53-57
    const v3 ffnormal = mk3(front ? n.x : -n.x, front ? n.y : -n.y, front ? n.z : -n.z);
    const v3 fhp = p.ray.o + p.ps.hit_dist * p.ray.d;
    Rng rng = p.rng;
    v3 scatter_pos;
    float light_area;
    LightSample ls;
    if (!nee_sample(sc, fhp, ffnormal, rng, scatter_pos, light_area, ls)) return false;
    return !walk_begin_any(sc, RayD{scatter_pos, ls.direction}, ls.dist - sc.eps, w);
}

}  // namespace rptdev
