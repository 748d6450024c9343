// dev_wavefront.h — large scenes (uniform grid, dev_scene_large.h) as a WAVEFRONT: the grid walks run in their own small
// kernel over ray lists in HBM, everything else of a bounce in a shading kernel (kernels.hip: wf_walk_kernel,
// wf_shade_kernel; host loop: RPT_LAUNCH_NS::render_wavefront).
//
// Why: inside the megakernel a wave leaves a grid walk when its LONGEST walk ends — 19.5 % of the lanes active on the
// 10 k-sphere scene (profiles/r2/block_profile_c5.txt) — and handing finished lanes new walks inside that kernel lost to
// the register state it had to keep.  A kernel that ONLY walks keeps 80 VGPRs per lane, none of them spilled:
// a lane whose walk ends takes the next ray of the list, whatever pixel it belongs to, at 6 waves per SIMD.
//
// Per pixel the arithmetic and its order are the megakernel's (same device functions), so the image is bit-identical:
//   * a pixel's samples are still traced one after the other (a slot per pixel, path regeneration in the slot);
//   * closest_hit = sphere 0 + oversize spheres (SHADE, every lane) -> grid walk (WALK) -> planes, lights (SHADE);
//   * the shadow ray of next-event estimation is walked one iteration LATER, together with the next bounce's ray:
//     SHADE computes the contribution as if the light were visible and parks it (c_lit); the next SHADE adds it, or not,
//     before anything else touches the radiance — the same additions in the same order.  (With a non-finite throughput
//     "add nothing" and "add 0 * throughput" differ: those lanes resolve their shadow ray on the spot.)
#pragma once

#include "dev_scene_large.h"

namespace rptdev {

enum : uint32_t { WF_WALKING = 0u, WF_ENDING = 1u, WF_DONE = 2u };      // slot status (ctl.w bits 0-1)

// Ray lists without a shared counter: every wave of SHADE (64 consecutive slots) owns a 64-entry SEGMENT of each list and
// compacts its rays into it with a ballot — no atomics.  (A first version appended to one global list with one atomicAdd per
// wave; ~200 k same-address atomics per launch took 3.4 ms, against 0.1 ms of arithmetic.)  The walk kernel's waves are split
// into kWalkGroups groups; group g owns segments g, g + G, g + 2G, ... (a uniform sample of the image, so the groups carry equal
// loads) and its waves take whole segments from the group's own counter (its own cache line: 2 * n_seg / G atomics each).
constexpr uint32_t kWalkGroups = 256u;
constexpr uint32_t kWalkCounterStride = 32u;       // dwords between two groups' counters (128 B)
constexpr uint32_t kWalkSteal = 3u;                // a wave whose group has run dry takes segments of this many following groups

struct WfBuffers {
    float4* ray_o;             // [n_slots] path ray origin; w: nearest sphere distance (in: after sphere 0 / oversize, out: after the walk)
    float4* ray_d;             //           direction; w: bits of the nearest sphere's index (0xFFFFFFFF none)
    float4* thr;               //           throughput; w: State.hit_dist
    float4* rad;               //           radiance; w: previous scatter pdf
    float4* sh_o;              //           shadow ray origin; w: max_dist
    float4* sh_d;              //           direction; w: bits of "occluded" (written by WALK)
    float4* c_lit;             //           the parked next-event contribution
    float4* prev;              //           radiance of the pixel's previous sample while its last shadow ray is out (ctl.w bit 3)
    uint4* ctl;                //           rng state, rng increment, bounce, sample << 8 | previous parked << 3 | shadow pending << 2 | status
    uint32_t* closest;         // [n_seg * 64] slots whose path ray needs a grid walk, per segment
    uint32_t* shadow;          // [n_seg * 64] slots whose shadow ray needs a grid walk
    uint32_t* cnt_closest;     // [n_seg] entries in each segment
    uint32_t* cnt_shadow;      // [n_seg]
    uint32_t* group_next;      // [kWalkGroups * kWalkCounterStride] the walk groups' segment counters
    uint32_t* any_active;      // [2] set by SHADE(k) (parity k & 1) when a slot still has work
    uint32_t n_slots, n_seg;
};

// closest_hit's sphere part arrives from the walk kernel
struct WaveQuery {
    float dist;
    uint32_t best;
    RPT_DEV bool geom(const SceneLarge& sc, const RayD& ray, PathState& ps, GeomHit& g, EmitterHit& e) const
    {
        return closest_geom_finish(sc, ray, ps, dist, best, best != 0xFFFFFFFFu, g, e);
    }
};

// Everything of grid_closest_sphere before the walk.  False: the ray is outside the grid's reach and was answered by the
// brute-force loop right here (dist / best final).
RPT_DEV bool closest_before_walk(const SceneLarge& sc, const RayD& ray, float& dist, uint32_t& best)
{
    dist = 3.40282347e+38f;
    best = 0xFFFFFFFFu;
    bool hit = false;
    if (!grid_usable(sc, ray)) { brute_closest_sphere(sc, ray, dist, best, hit); return false; }
    {
        const float4 s = sphere_uniform(sc, 0);
        float t;
        if (hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t)) { dist = t; best = 0; }
    }
    for (uint32_t j = 0; j < sc.n_oversize; ++j) {
        const uint32_t i = ((cuint_p)sc.oversize)[j];
        const float4 s = sphere_uniform(sc, i);
        float t;
        if (i != 0u && hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t) && (t < dist || (t == dist && i < best))) { dist = t; best = i; }
    }
    return true;
}

struct ShadowReq {
    bool pending;              // the shadow ray still needs its grid walk
    RayD ray;
    float max_dist;
    v3 c_lit;                  // what the radiance gains if the walk finds nothing
};

// The shadow query of path_shade_full (dev_integrator.h) with the grid walk left to the walk kernel: any_hit taken apart —
// planes, reach test and oversize spheres here; when those leave the question open the ray is handed to the caller (`sr`) and
// SHADE goes on as if the light were visible; what that adds to the radiance is parked (c_lit) until the walk has answered.
// A lane whose throughput is not finite resolves its shadow ray on the spot: there "add nothing" and "add 0 x throughput" differ.
struct DeferredQuery {
    ShadowReq* sr;
    RPT_DEV bool any(const SceneLarge& sc, const RayD& shadow, float max_dist, v3 th, bool& pending) const
    {
        pending = false;
        const bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
        bool occluded = any_hit_finish(sc, shadow, max_dist, false);                          // the planes
        if (!occluded) {
            const bool finite = (__builtin_fabsf(th.x) < __builtin_inff()) && (__builtin_fabsf(th.y) < __builtin_inff()) &&
                                (__builtin_fabsf(th.z) < __builtin_inff());
            if (!grid_usable(sc, shadow) || !finite) {
                occluded = grid_any_sphere(sc, shadow, use_max, max_dist);
            } else {
                for (uint32_t j = 0; j < sc.n_oversize; ++j) {
                    const float4 s = sphere_uniform(sc, ((cuint_p)sc.oversize)[j]);
                    float t;
                    if (hit_sphere(shadow, mk3(s.x, s.y, s.z), s.w, t) && (!use_max || t < max_dist)) occluded = true;
                }
                if (!occluded) { pending = true; sr->pending = true; sr->ray = shadow; sr->max_dist = max_dist; }
            }
        }
        return occluded;
    }
    RPT_DEV void park(v3 gain) const { sr->c_lit = gain; }
};

// path_shade_full with the shadow ray's grid walk left to the walk kernel.
template <class S>
RPT_DEV bool path_shade_deferred(const S& sc, PathRegs& p, const GeomHit& g, ShadowReq& sr)
{
    sr.pending = false;
    return path_shade_full(sc, DeferredQuery{&sr}, p, g);
}

// One cell of grid_closest_sphere / grid_any_sphere (dev_scene_large.h) for the ray a lane is walking; true: the walk is over.
// The cell's list, four entries per trip: the loads go out together and the discriminants are computed branch-free.  A candidate
// (the line meets the sphere: few) is parked — hit_sphere's tca and radius2 - d2 — and its square-root half runs once per cell for
// all lanes that have one, not per entry for one lane in twenty.  Acceptance is order-independent (nearest t, lowest index on
// ties; "any" for shadow rays), so neither batching nor parking changes the result.
//   shadow walks: `occluded` is raised on a hit (within max_dist when the scene honours it); path walks: dist / best are updated.
RPT_DEV bool walk_cell(const SceneLarge& sc, const RayD& ray, GridWalk& g, uint32_t& k0, uint32_t& k1, uint32_t& guard, bool shadow, bool use_max,
                       float max_dist, float& dist, uint32_t& best, bool& occluded)
{
    const float t_exit = grid_cell_exit(g);
    grid_step(sc, g);
    uint32_t n0 = 0, n1 = 0;
    if (g.alive) cell_bounds(sc, grid_cell_index(sc, g), n0, n1);
    bool done = false;
    float c_tca0 = 0.0f, c_rd0 = 0.0f, c_tca1 = 0.0f, c_rd1 = 0.0f;
    uint32_t c_k0 = 0u, c_k1 = 0u, nc = 0u;
    auto resolve = [&](float tca, float rd, uint32_t kk) {              // hit_sphere's second half + acceptance
        const float thc = fsqrt(rd);
        float t0 = tca - thc;
        float t1 = tca + thc;
        if (t0 > t1) { const float tmp = t0; t0 = t1; t1 = tmp; }
        bool ok = true;
        if (t0 < 0.0f) {
            t0 = t1;
            if (t0 < 0.0f) ok = false;
        }
        if (ok) {
            if (shadow) {
                if (!use_max || t0 < max_dist) done = true;
            } else {
                const uint32_t idx = sc.cell_items[kk];
                if (idx != 0u && (t0 < dist || (t0 == dist && idx < best))) { dist = t0; best = idx; }
            }
        }
    };
    for (uint32_t k = k0; k < k1; k += 4u) {
        float4 sp[4];
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) {
            sp[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (k + i < k1) sp[i] = sc.cell_spheres[k + i];
        }
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) {
            const v3 l = mk3(sp[i].x, sp[i].y, sp[i].z) - ray.o;        // hit_sphere's first half
            const float tca = dot3(l, ray.d);
            const float d2 = dot3(l, l) - tca * tca;
            const float radius2 = sp[i].w * sp[i].w;
            const bool cand = (k + i < k1) && !(d2 > radius2);
            const float rd = radius2 - d2;
            if (cand) {
                if (nc == 0u) { c_tca0 = tca; c_rd0 = rd; c_k0 = k + i; nc = 1u; }
                else if (nc == 1u) { c_tca1 = tca; c_rd1 = rd; c_k1 = k + i; nc = 2u; }
                else resolve(tca, rd, k + i);                           // a third candidate in one cell: at once
            }
        }
    }
    if (nc >= 1u) resolve(c_tca0, c_rd0, c_k0);
    if (nc >= 2u) resolve(c_tca1, c_rd1, c_k1);
    if (shadow) {
        if (done) occluded = true;
        done = done || (t_exit > g.t_end) || (use_max && t_exit > max_dist);
    } else {
        done = (best != 0xFFFFFFFFu && dist <= t_exit) || (t_exit > g.t_end);
    }
    guard -= 1u;
    done = done || !g.alive || guard == 0u;
    k0 = n0; k1 = n1;
    return done;
}

// The two grid walks of a bounce in ONE loop of the wave (render_large_pair_kernel): every lane walks its parked shadow ray
// (if it has one) and then its path ray (if it has one); a lane that finishes a walk waits until at most `refill_at` lanes are
// still walking, then all idle lanes with a ray left set theirs up together.  Inside the megakernel a wave otherwise leaves each
// walk when its LONGEST walk ends, shadow walks and path walks separately (19.5 % of the lanes active, profiles/r2/block_profile_c5.txt);
// back to back in one loop the wave waits for the longest SUM, and idle lanes refill.
//   shadow: in has_shadow / sh / sh_max, out occluded (by the grid walk; the planes, oversize spheres and the reach test were the caller's)
//   path:   in has_path (closest_before_walk returned true), dist / best in-out
RPT_DEV void grid_walk_pair(const SceneLarge& sc, bool has_shadow, const RayD& sh, float sh_max, bool& occluded, bool has_path, const RayD& path,
                            float& dist, uint32_t& best, uint32_t refill_at)
{
    const bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
    occluded = false;
    uint32_t next = has_shadow ? 0u : (has_path ? 1u : 2u);         // the lane's next ray: 0 shadow, 1 path, 2 none
    bool has = false, shadow = false;
    uint32_t k0 = 0, k1 = 0, guard = 0;
    RayD ray{mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 0.0f)};
    GridWalk g;
    g.alive = false;
    for (;;) {
        uint32_t n_has = (uint32_t)__popcll(__ballot(has));
        uint64_t m_more = __ballot(!has && next < 2u);
        if (m_more != 0ull && n_has <= refill_at) {
            RPT_PROF(PB_GRID_BEGIN);
            if (!has && next < 2u) {
                shadow = next == 0u;
                ray = shadow ? sh : path;
                next = (shadow && has_path) ? 1u : 2u;
                g = grid_begin(sc, ray);
                if (g.alive) {
                    has = true;
                    cell_bounds(sc, grid_cell_index(sc, g), k0, k1);
                    guard = sc.gn[0] + sc.gn[1] + sc.gn[2] + 3u;
                }                                                   // else: the ray misses the grid, nothing to walk
            }
            m_more = __ballot(!has && next < 2u);
        }
        n_has = (uint32_t)__popcll(__ballot(has));
        if (n_has == 0u) {
            if (m_more == 0ull) break;
            continue;                                               // rays that missed the grid: set up the lanes' next ones
        }
        do {
            if (has) {
                RPT_PROF(PB_GRID_CELL);
                if (walk_cell(sc, ray, g, k0, k1, guard, shadow, use_max, sh_max, dist, best, occluded)) has = false;
            }
            n_has = (uint32_t)__popcll(__ballot(has));
            m_more = __ballot(!has && next < 2u);
        } while (n_has != 0u && (m_more == 0ull || n_has > refill_at));
    }
}

// ---------------------------------------------------------------------------
// The walk kernel's body: every lane walks one ray of the lists at a time and takes the next one when its walk ends.
// Refills happen for the whole wave at once, when at most `refill_at` lanes are still walking (the set-up — ray load,
// slab test, nine divides of the DDA — then runs for many lanes, not for one).
// ---------------------------------------------------------------------------
RPT_DEV void wf_walk_body(const SceneLarge& sc, const WfBuffers& wb, uint32_t refill_at)
{
    const bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
    const uint32_t lane = __lane_id();
    const uint32_t group = blockIdx.x % kWalkGroups;
    const uint32_t per_group = (wb.n_seg + kWalkGroups - 1u) / kWalkGroups;        // segments of one list a group owns
    uint32_t* next = wb.group_next + group * kWalkCounterStride;
    uint32_t cur = group, visited = 0u;

    // the wave's current segment (wave-uniform)
    uint32_t seg_base = 0u, seg_pos = 0u, seg_cnt = 0u;
    bool seg_shadow = false;

    bool has = false, shadow = false, exhausted = false;
    uint32_t slot = 0, best = 0, k0 = 0, k1 = 0, guard = 0;
    float dist = 0.0f, max_dist = 0.0f;
    RayD ray{mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 0.0f)};
    GridWalk g;
    g.alive = false;

    RPT_PROF(PB_WF_WAVE);
    for (;;) {
        uint32_t n_has = (uint32_t)__popcll(__ballot(has));
        if (!exhausted && n_has <= refill_at) {
            uint64_t m_need = __ballot(!has);
            bool got = false;
            RPT_PROF(PB_WF_FETCH);
            while (m_need != 0ull && !exhausted) {
                if (seg_pos >= seg_cnt) {                           // take the group's next segment
                    uint32_t j = 0u;
                    if (lane == 0u) j = atomicAdd(next, 1u);
                    j = (uint32_t)__shfl((int)j, 0);
                    if (j >= 2u * per_group) {                      // this group's segments are all taken: help the next groups
                        if (visited == kWalkSteal) { exhausted = true; break; }
                        visited += 1u;
                        cur = (cur + 1u) % kWalkGroups;
                        next = wb.group_next + cur * kWalkCounterStride;
                        continue;
                    }
                    seg_shadow = j >= per_group;
                    const uint32_t seg = cur + (seg_shadow ? j - per_group : j) * kWalkGroups;
                    seg_pos = 0u;
                    seg_cnt = (seg < wb.n_seg) ? (seg_shadow ? wb.cnt_shadow : wb.cnt_closest)[seg] : 0u;
                    seg_base = seg * 64u;
                    continue;
                }
                const uint32_t rank = (uint32_t)__popcll(m_need & ((1ull << lane) - 1ull));
                const uint32_t avail = seg_cnt - seg_pos;
                const bool mine = !has && !got && ((m_need >> lane) & 1ull) && rank < avail;
                if (mine) {
                    shadow = seg_shadow;
                    slot = (seg_shadow ? wb.shadow : wb.closest)[seg_base + seg_pos + rank];
                    got = true;
                }
                const uint32_t n_need = (uint32_t)__popcll(m_need);
                seg_pos += (n_need < avail) ? n_need : avail;
                m_need = __ballot(!has && !got);
            }
            if (got) {
                RPT_PROF(PB_WF_SETUP);
                const float4 o = shadow ? wb.sh_o[slot] : wb.ray_o[slot];
                const float4 d = shadow ? wb.sh_d[slot] : wb.ray_d[slot];
                ray.o = mk3(o.x, o.y, o.z);
                ray.d = mk3(d.x, d.y, d.z);
                dist = o.w; max_dist = o.w;                         // (each kind reads its own)
                best = rpt_f2u(d.w);
                g = grid_begin(sc, ray);
                if (g.alive) {
                    has = true;
                    cell_bounds(sc, grid_cell_index(sc, g), k0, k1);
                    guard = sc.gn[0] + sc.gn[1] + sc.gn[2] + 3u;
                }                                                   // else: the ray misses the grid; what SHADE wrote is the answer
            }
        }
        if (__ballot(has) == 0ull) {
            if (exhausted) break;
            continue;
        }
        do {
            if (has) {                                              // one cell of grid_closest_sphere / grid_any_sphere
                RPT_PROF(PB_WF_CELL);
                bool occluded = false;
                if (walk_cell(sc, ray, g, k0, k1, guard, shadow, use_max, max_dist, dist, best, occluded)) {
                    if (!shadow) { wb.ray_o[slot].w = dist; wb.ray_d[slot].w = rpt_u2f(best); }
                    has = false;
                }
                if (occluded) wb.sh_d[slot].w = rpt_u2f(1u);
            }
            n_has = (uint32_t)__popcll(__ballot(has));
        } while (n_has != 0u && (exhausted || n_has > refill_at));
    }
}

}  // namespace rptdev
