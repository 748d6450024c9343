// dev_scene_large.h — scene queries for analytical scenes too large for the kernarg
// tables of dev_scene.h (BASELINE.json configs[4]: thousands of spheres, tens of lights).
//
// The tables live in HBM.  Loops over primitives use a wave-uniform index and read through
// constant-address-space pointers, so hipcc emits s_load_dwordx4 and the sphere lands in
// SGPRs (scalar cache / L2 resident: 16 B per sphere, the whole table is re-read by every
// wave for every ray — brute force; the acceleration structure is the next step,
// DESIGN.md §8).  Data that depends on the lane (the winning sphere's centre, its material,
// the sampled light) is gathered with ordinary vector loads.
//
// Material layering (dev_integrator.h, apply_patch) needs one bit per primitive and does not
// scale; large scenes therefore require every SPHERE material to be a full patch
// (mask == RPT_MAT_ALL, no procedural part), which rpt_upload_scene checks.  With full
// patches the ordered-acceptance rule of analytical.rs:36-120 reduces to "the last accepted
// sphere's material", i.e. the nearest sphere's (first index on ties), so the result is still
// exactly what the ordered loop gives.  Plane materials stay arbitrary patches.
#ifndef RPT_NS                        // (the namespace of this pass: dev_math.h, "two passes")
#define RPT_NS rptdev
#endif
#if (defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_SCENE_LARGE_H_PLAIN)) || (!defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_SCENE_LARGE_H_NORMAL))
#ifdef RPT_PLAIN_PASS
#define RPT_DEV_SCENE_LARGE_H_PLAIN
#else
#define RPT_DEV_SCENE_LARGE_H_NORMAL
#endif

#include "dev_integrator.h"

namespace RPT_NS {
using namespace rptscene;


// Wave-uniform table reads: plain dwords through the constant address space, which the
// compiler merges into s_load_dwordx4/x8.
typedef const RPT_CONST_AS float* cfloat_p;
typedef const RPT_CONST_AS uint32_t* cuint_p;

// A per-lane gather from one of the scene's tables with a 32-BIT byte offset: `base + zext(index * sizeof(T))` is what a global load with
// a scalar base and a vector offset addresses (global_load ... v_off, s[base:base+1]) — one 32-bit multiply / shift and ONE address
// register per lane.  `table[index]` is base + zext(index) * sizeof(T), 34 and more bits, so the compiler builds a 64-bit address in
// two registers with v_lshl_add_u64 behind a v_mov 0 for every load of a walk.  The tables are far below 4 GiB (rpt_upload_scene checks).
template <class T>
RPT_DEV T gather32(const T* table, uint32_t index)
{
    const uint32_t off = index * (uint32_t)sizeof(T);
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(table) + off);
}

RPT_DEV float4 sphere_uniform(const SceneLarge& sc, uint32_t i)     // i wave-uniform -> scalar load
{
    cfloat_p p = (cfloat_p)sc.spheres + 4u * i;
    return make_float4(p[0], p[1], p[2], p[3]);
}

RPT_DEV DevLight light_uniform(const SceneLarge& sc, uint32_t i)
{
    static_assert(sizeof(DevLight) == 15 * 4, "DevLight is 15 dwords");
    cfloat_p p = (cfloat_p)sc.lights + 15u * i;
    DevLight L;
    L.type = ((cuint_p)p)[0];
    L.px = p[1]; L.py = p[2]; L.pz = p[3];
    L.ex = p[4]; L.ey = p[5]; L.ez = p[6];
    L.radius = p[7]; L.area = p[8];
    L.ux = p[9]; L.uy = p[10]; L.uz = p[11];
    L.vx = p[12]; L.vy = p[13]; L.vz = p[14];
    return L;
}

RPT_DEV DevMaterial material_uniform(const SceneLarge& sc, uint32_t i)
{
    static_assert(sizeof(DevMaterial) == 29 * 4, "DevMaterial is 29 dwords");
    cfloat_p p = (cfloat_p)sc.materials + 29u * i;
    DevMaterial m;
    m.mask = ((cuint_p)p)[0]; m.proc_kind = ((cuint_p)p)[1];
    for (int c = 0; c < 3; ++c) { m.rgb[c] = p[2 + c]; m.emission[c] = p[5 + c]; }
    m.anisotropic = p[8]; m.metallic = p[9]; m.roughness = p[10]; m.subsurface = p[11]; m.specular_tint = p[12];
    m.sheen = p[13]; m.sheen_tint = p[14]; m.clearcoat = p[15]; m.clearcoat_gloss = p[16]; m.spec_trans = p[17]; m.ior = p[18];
    for (int c = 0; c < 4; ++c) m.proc_params[c] = p[19 + c];
    m.medium_type = ((cuint_p)p)[23]; m.medium_density = p[24];
    for (int c = 0; c < 3; ++c) m.medium_color[c] = p[25 + c];
    m.medium_anisotropy = p[28];
    return m;
}

RPT_DEV DevLight light_at(const SceneLarge& sc, uint32_t index)     // per-lane index -> gather
{
    return gather32(sc.lights, index);
}

// ---------------------------------------------------------------------------
// Grid traversal (3D DDA).  The brute-force loop accepts sphere i when it is hit and
// (i == 0 or t < dist), i.e. it ends with the hit of smallest t, lowest index on ties
// (and sphere 0 unconditionally if it is hit first).  Visiting spheres in grid order gives the
// same winner with the acceptance rule  t < dist || (t == dist && i < best); sphere 0 is
// tested up front exactly like the loop does.  A sphere is registered in every cell its
// bounding box overlaps, and the box is PADDED by how far outside the sphere a line can pass and
// still be called a hit by the reference's f32 test: d2 = l.l - tca*tca cancels catastrophically,
// with an absolute error of about 4e-7 * |l|^2, so from 250 units away a line 0.1 outside a
// 0.3-radius sphere can "hit" it — and the brute-force loop (the semantics to reproduce) reports
// that.  The padding covers every origin within sqrt(safe_r2) of the grid centre (host:
// build_grid); rays that start farther away (grazing floor hits tens of thousands of units out,
// where the test is pure noise) take the brute-force loop instead.
// ---------------------------------------------------------------------------
#ifndef RPT_GRID_BATCH
#define RPT_GRID_BATCH 2          // list entries per trip in grid_closest_sphere ...
#endif
#ifndef RPT_GRID_BATCH_ANY
#define RPT_GRID_BATCH_ANY 2      // ... and in grid_any_sphere
#endif

struct GridWalk {
    int ix, iy, iz;
    int sx, sy, sz;
    float tmx, tmy, tmz;       // t at which the ray leaves the current cell along each axis
    float tdx, tdy, tdz;
    float t_end;               // t at which the ray leaves the grid
    uint32_t coff;             // the tier of cell lists this ray walks: 0 or near_cell_off (grid_tier)
    bool alive;
};

// The tier of cell lists that serves a ray: the padding a tier's lists carry against the reference's cancellation error grows
// with the square of the farthest origin it serves, so origins near the grid (every bounce of a camera inside the scene) get
// their own, far shorter lists.  (A NaN origin compares false: the far tier; grid_usable sends it to the brute-force loop anyway.)
RPT_DEV uint32_t grid_tier(const SceneLarge& sc, const RayD& ray)
{
    const float dx = ray.o.x - sc.gcenter[0], dy = ray.o.y - sc.gcenter[1], dz = ray.o.z - sc.gcenter[2];
    return ((dx * dx + dy * dy + dz * dz) <= sc.near_r2) ? sc.near_cell_off : 0u;
}

// Set-up of a walk: slab test against the grid box, first cell, the DDA's increments.  The grid is ours, not the reference's: what
// must be exact is the walk's ANSWER (test_grid_queries_equal_brute_force), and the lists carry 1e-3 cell sizes of allowance for
// the DDA's own rounding — so the set-up uses one hardware reciprocal per axis (v_rcp_f32, 1 ulp) where it had nine correctly
// rounded divides, and selects where it had branches (round 4: 10 k spheres +2 %).
RPT_DEV GridWalk grid_begin(const SceneLarge& sc, const RayD& ray)
{
    GridWalk g;
    g.coff = grid_tier(sc, ray);
    float t0 = 0.0f, t1 = 3.40282347e+38f;
    const float o[3] = {ray.o.x, ray.o.y, ray.o.z};
    const float d[3] = {ray.d.x, ray.d.y, ray.d.z};
    float inv[3];
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const bool flat = d[a] == 0.0f;                             // the ray runs inside one slab of cells along this axis, or misses the box
        inv[a] = __builtin_amdgcn_rcpf(flat ? 1.0f : d[a]);
        const float ta = (sc.gmin[a] - o[a]) * inv[a];
        const float tb = (sc.gmax[a] - o[a]) * inv[a];
        const float lo = ta < tb ? ta : tb;
        const float hi = ta < tb ? tb : ta;
        t0 = (!flat && lo > t0) ? lo : t0;
        t1 = (!flat && hi < t1) ? hi : t1;
        ok = ok && (!flat || ((o[a] >= sc.gmin[a]) && (o[a] <= sc.gmax[a])));
    }
    g.alive = ok && (t0 <= t1);                                     // (false for NaN)
    g.t_end = t1;
    int cell[3], step[3];
    float tmax[3], tdel[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float p = o[a] + t0 * d[a];
        int c = (int)__builtin_floorf((p - sc.gmin[a]) * sc.inv_cell_size[a]);
        const int n = (int)sc.gn[a];
        c = c < 0 ? 0 : (c > n - 1 ? n - 1 : c);
        cell[a] = c;
        const bool fwd = d[a] > 0.0f, flat = d[a] == 0.0f;
        step[a] = flat ? 0 : (fwd ? 1 : -1);
        const float edge = sc.gmin[a] + (float)(fwd ? c + 1 : c) * sc.cell_size[a];
        tmax[a] = flat ? 3.40282347e+38f : (edge - o[a]) * inv[a];
        tdel[a] = flat ? 3.40282347e+38f : sc.cell_size[a] * __builtin_fabsf(inv[a]);
    }
    g.ix = cell[0]; g.iy = cell[1]; g.iz = cell[2];
    g.sx = step[0]; g.sy = step[1]; g.sz = step[2];
    g.tmx = tmax[0]; g.tmy = tmax[1]; g.tmz = tmax[2];
    g.tdx = tdel[0]; g.tdy = tdel[1]; g.tdz = tdel[2];
    return g;
}

// cell_start[c], cell_start[c + 1] in one 8-byte load (dword-aligned)
RPT_DEV void cell_bounds(const SceneLarge& sc, uint32_t c, uint32_t& k0, uint32_t& k1)
{
    struct __attribute__((packed, aligned(4))) Pair { uint32_t a, b; };
    const Pair r = *reinterpret_cast<const Pair*>(reinterpret_cast<const char*>(sc.cell_start) + c * 4u);      // (a 32-bit byte offset: gather32)
    k0 = r.a; k1 = r.b;
}

RPT_DEV uint32_t grid_cell_index(const SceneLarge& sc, const GridWalk& g)
{
    return ((uint32_t)g.iz * sc.gn[1] + (uint32_t)g.iy) * sc.gn[0] + (uint32_t)g.ix + g.coff;
}

// t at which the ray leaves the current cell
RPT_DEV float grid_cell_exit(const GridWalk& g)
{
    float m = g.tmx < g.tmy ? g.tmx : g.tmy;
    return m < g.tmz ? m : g.tmz;
}

// The megakernel's step (round 4).  Selects, not a three-way branch: every lane of a wave takes its own way, so all three arms ran
// anyway, each behind its own exec-mask bookkeeping on the scalar unit.  And no in-range test of the new cell: the walks below end
// when the ray leaves the grid BOX (t_exit > t_end, which they test anyway) and clamp the cell index instead.  Should rounding
// let the indices leave the grid a step before the exit time says so, the walk tests the spheres of a cell it need not have
// visited, which cannot change its answer: the answer is the nearest (any) hit among ALL spheres — the reference's loop tests
// every one — and a walk is right as long as the cells it MUST visit are among those it visits.
RPT_DEV void grid_advance(GridWalk& g)
{
    const bool ax = g.tmx <= g.tmy && g.tmx <= g.tmz;
    const bool ay = !ax && g.tmy <= g.tmz;
    const bool az = !ax && !ay;
    g.ix += ax ? g.sx : 0; g.iy += ay ? g.sy : 0; g.iz += az ? g.sz : 0;
    g.tmx = ax ? g.tmx + g.tdx : g.tmx; g.tmy = ay ? g.tmy + g.tdy : g.tmy; g.tmz = az ? g.tmz + g.tdz : g.tmz;
}

RPT_DEV uint32_t grid_cell_index_clamped(const SceneLarge& sc, const GridWalk& g)
{
    const uint32_t last = sc.gn[0] * sc.gn[1] * sc.gn[2] - 1u;     // (scalar)
    const uint32_t c = ((uint32_t)g.iz * sc.gn[1] + (uint32_t)g.iy) * sc.gn[0] + (uint32_t)g.ix;     // an index outside the grid wraps or aliases: any cell will do
    return (c < last ? c : last) + g.coff;
}

// nearest sphere along the ray (dist/best in-out), equivalent to the ordered loop over all spheres
RPT_DEV bool grid_usable(const SceneLarge& sc, const RayD& ray)
{
    const float dx = ray.o.x - sc.gcenter[0], dy = ray.o.y - sc.gcenter[1], dz = ray.o.z - sc.gcenter[2];
    return (dx * dx + dy * dy + dz * dz) <= sc.safe_r2;              // false for NaN origins too
}

RPT_DEV void brute_closest_sphere(const SceneLarge& sc, const RayD& ray, float& dist, uint32_t& best, bool& hit)
{
    for (uint32_t i = 0; i < sc.n_spheres; ++i) {
        const float4 s = sphere_uniform(sc, i);
        float t;
        bool h = hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t);
        bool acc = h && (i == 0 || t < dist);                       // analytical.rs:43 / :74
        if (acc) {
            dist = t;
            best = i;
            hit = true;
        }
    }
}

RPT_DEV bool brute_any_sphere(const SceneLarge& sc, const RayD& ray, bool use_max, float max_dist)
{
    bool occluded = false;
    for (uint32_t i = 0; i < sc.n_spheres; ++i) {
        const float4 s = sphere_uniform(sc, i);
        float t;
        bool h = hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t);
        occluded = occluded || (h && (!use_max || t < max_dist));
    }
    return occluded;
}

// A closest-hit walk between two cells: the cell the ray is in, its list [k0, k1) and that list's first entries, already loaded.
struct ClosestWalk {
    GridWalk g;
    uint32_t k0, k1;
    float4 pf[RPT_GRID_BATCH];                                      // in flight across the loop's back edge
};

// The walk is bound by the latency of its dependent loads (cell -> list bounds -> spheres), not by arithmetic, so the next cell's
// list bounds are requested before this cell's spheres are tested, and the first entries of the next list with them.
RPT_DEV void closest_walk_fetch(const SceneLarge& sc, ClosestWalk& w, uint32_t cell)
{
    cell_bounds(sc, cell, w.k0, w.k1);
#pragma unroll
    for (uint32_t j = 0; j < RPT_GRID_BATCH; ++j) w.pf[j] = gather32(sc.cell_spheres, w.k0 + j);     // (the array ends in spare entries: host_grid.h)
}

// One cell of the walk: true when the walk is over (nothing beyond this cell can be nearer, or the ray leaves the grid box here).
// `hit_w` (a sphere has been accepted) and "a candidate is parked" are WORDS in vector registers: as bools the compiler keeps them as
// lane masks in SGPRs and merges them with scalar instructions at every join of the walk's nested branches — and the scalar unit, one
// per CU for its four SIMDs, issues one instruction for every two vector ones in this kernel (profiles/r3/c5_megakernel: 2.3e10
// SALU + 5.5e9 branches against 4.6e10 VALU).  10 k spheres, 2048^2 x 32 spp: 2 138 -> 2 212 Msamples/s (+3.5 %), round 4.
RPT_DEV bool closest_walk_cell(const SceneLarge& sc, const RayD& ray, ClosestWalk& w, float& dist, uint32_t& best, uint32_t& hit_w)
{
    RPT_PROF(PB_GRID_CELL);
    GridWalk& g = w.g;
    const uint32_t k0 = w.k0, k1 = w.k1;
    const float t_exit = grid_cell_exit(g);                         // of the cell whose list is [k0, k1)
    const bool last = t_exit > g.t_end;                             // the ray leaves the grid box in this cell
    grid_advance(g);                                                // g is the NEXT cell from here on
    uint32_t n0, n1;
    cell_bounds(sc, grid_cell_index_clamped(sc, g), n0, n1);        // (also when this is the last cell: the index is always a cell's, and an unconditional load needs no exec mask)
    // The cell's list, RPT_GRID_BATCH entries per trip: the loads go out together, hit_sphere's discriminant is computed
    // branch-free for all of them and only candidates (the line meets the sphere: few) take its square-root half.  The
    // acceptance rule is order-independent, so neither batching nor parking changes the winner.  (Against the plain loop
    // over hit_sphere, 10 k spheres: two-phase test +4.6 %, 2 per trip +1.8 %, one 8-byte load for the list bounds +1.4 %;
    // 4 per trip: more live registers than the kernel has, -6 %.)
    // The first candidate of a cell parks hit_sphere's tca and radius2 - d2 and its square-root half runs once, behind the
    // list, with the other lanes' (+1.4 %); further candidates of the same cell are resolved at once.
    float c_tca = 0.0f, c_rd = 0.0f;
    uint32_t c_k = 0xFFFFFFFFu;                                     // 0xFFFFFFFF: nothing parked
    // hit_sphere's second half (analytical.rs:176-189) for a candidate, in selects.  Its swap of the roots is not here: thc is a
    // square root, so t0 = tca - thc <= tca + thc = t1 unless both are NaN, and then nothing below accepts them either.  The
    // sphere's index is loaded only for a root that can still win (t <= dist).
    auto resolve = [&](float tca, float rd, uint32_t kk) {
        RPT_PROF(PB_GRID_RESOLVE);
        const float thc = fsqrt(rd);
        const float t0 = tca - thc;
        const float t1 = tca + thc;
        const float t = t0 < 0.0f ? t1 : t0;
        if (!(t < 0.0f) && t <= dist) {
            const uint32_t i = gather32(sc.cell_items, kk);
            if (i != 0u && (t < dist || i < best)) { dist = t; best = i; hit_w = 1u; }
        }
    };
    auto test_batch = [&](const float4* sp, uint32_t k) {
#pragma unroll
        for (uint32_t j = 0; j < RPT_GRID_BATCH; ++j) {
            const v3 l = mk3(sp[j].x, sp[j].y, sp[j].z) - ray.o;
            const float tca = dot3(l, ray.d);
            const float d2 = dot3(l, l) - tca * tca;
            const float radius2 = sp[j].w;                      // (the list-ordered copy holds r * r: host_grid.h)
            if ((k + j < k1) && !(d2 > radius2)) {
                if (c_k == 0xFFFFFFFFu) { c_tca = tca; c_rd = radius2 - d2; c_k = k + j; }
                else resolve(tca, radius2 - d2, k + j);
            }
        }
    };
    test_batch(w.pf, k0);                                           // the list's first entries were requested a cell ago (+1.5 %, round 4)
    for (uint32_t k = k0 + RPT_GRID_BATCH; k < k1; k += RPT_GRID_BATCH) {
        RPT_PROF(PB_GRID_EXTRA);
        float4 sp[RPT_GRID_BATCH];
#pragma unroll
        for (uint32_t j = 0; j < RPT_GRID_BATCH; ++j) sp[j] = gather32(sc.cell_spheres, k + j);
        test_batch(sp, k);
    }
    if (c_k != 0xFFFFFFFFu) resolve(c_tca, c_rd, c_k);
    if (hit_w != 0u && dist <= t_exit) return true;
    if (last) return true;
    w.k0 = n0; w.k1 = n1;
#pragma unroll
    for (uint32_t j = 0; j < RPT_GRID_BATCH; ++j) w.pf[j] = gather32(sc.cell_spheres, n0 + j);
    return false;
}

// A DDA crosses at most nx + ny + nz cells; with this many trips as its guard every wave leaves a walk loop whatever the ray holds.
RPT_DEV uint32_t grid_walk_guard(const SceneLarge& sc) { return sc.gn[0] + sc.gn[1] + sc.gn[2] + 3u; }

RPT_DEV void grid_closest_sphere(const SceneLarge& sc, const RayD& ray, float& dist, uint32_t& best, bool& hit)
{
    if (!grid_usable(sc, ray)) { brute_closest_sphere(sc, ray, dist, best, hit); return; }
    uint32_t hit_w = hit ? 1u : 0u;
    {   RPT_PROF(PB_WALK_HEAD);
    {   // sphere 0: accepted whenever it is hit (analytical.rs:43)
        const float4 s = sphere_uniform(sc, 0);
        float t;
        if (hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t)) { dist = t; best = 0; hit_w = 1u; }
    }
    for (uint32_t j = 0; j < sc.n_oversize; ++j) {                  // the spheres that are not in the grid (wave-uniform loop)
        const uint32_t i = ((cuint_p)sc.oversize)[j];
        const float4 s = sphere_uniform(sc, i);
        float t;
        if (i != 0u && hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t) && (t < dist || (t == dist && i < best))) { dist = t; best = i; hit_w = 1u; }
    }
    }
    ClosestWalk w;
    { RPT_PROF(PB_GRID_BEGIN); w.g = grid_begin(sc, ray); }
    w.k0 = 0; w.k1 = 0;
    if (w.g.alive) cell_bounds(sc, grid_cell_index(sc, w.g), w.k0, w.k1);
#pragma unroll
    for (uint32_t j = 0; j < RPT_GRID_BATCH; ++j) w.pf[j] = gather32(sc.cell_spheres, w.k0 + j);
    if (w.g.alive)
    for (uint32_t guard = grid_walk_guard(sc); guard != 0u; --guard)
        if (closest_walk_cell(sc, ray, w, dist, best, hit_w)) break;
    hit = hit_w != 0u;
}

RPT_DEV bool grid_any_sphere(const SceneLarge& sc, const RayD& ray, bool use_max, float max_dist)
{
    if (!grid_usable(sc, ray)) return brute_any_sphere(sc, ray, use_max, max_dist);
    for (uint32_t j = 0; j < sc.n_oversize; ++j) {
        const float4 s = sphere_uniform(sc, ((cuint_p)sc.oversize)[j]);
        float t;
        if (hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t) && (!use_max || t < max_dist)) return true;
    }
    GridWalk g;
    { RPT_PROF(PB_GRID_BEGIN); g = grid_begin(sc, ray); }
    uint32_t k0 = 0, k1 = 0;
    uint32_t occluded = 0u;                                         // (a word, not a bool: see grid_closest_sphere)
    if (g.alive) cell_bounds(sc, grid_cell_index(sc, g), k0, k1);
    if (g.alive)
    for (uint32_t guard = sc.gn[0] + sc.gn[1] + sc.gn[2] + 3u; guard != 0u; --guard) {
        RPT_PROF(PB_GRID_CELL_ANY);
        const float t_exit = grid_cell_exit(g);
        const bool last = t_exit > g.t_end;
        grid_advance(g);                                            // as in grid_closest_sphere
        uint32_t n0, n1;
        cell_bounds(sc, grid_cell_index_clamped(sc, g), n0, n1);
        // (as in grid_closest_sphere; parking the candidate as well, or 3 per trip, is slower here: -2 %, -4 %)
        for (uint32_t k = k0; k < k1 && occluded == 0u; k += RPT_GRID_BATCH_ANY) {
            float4 sp[RPT_GRID_BATCH_ANY];
            float c_tca[RPT_GRID_BATCH_ANY], c_rd[RPT_GRID_BATCH_ANY];
            bool cand[RPT_GRID_BATCH_ANY];
            bool any_cand = false;
#pragma unroll
            for (uint32_t j = 0; j < RPT_GRID_BATCH_ANY; ++j) sp[j] = gather32(sc.cell_spheres, k + j);     // (requested a cell ahead as in the closest walk: -6 %)
#pragma unroll
            for (uint32_t j = 0; j < RPT_GRID_BATCH_ANY; ++j) {
                const v3 l = mk3(sp[j].x, sp[j].y, sp[j].z) - ray.o;
                const float tca = dot3(l, ray.d);
                const float d2 = dot3(l, l) - tca * tca;
                const float radius2 = sp[j].w;                      // (the list-ordered copy holds r * r: host_grid.h)
                cand[j] = (k + j < k1) && !(d2 > radius2);
                any_cand = any_cand || cand[j];
                c_tca[j] = tca;
                c_rd[j] = radius2 - d2;
            }
            if (any_cand) {
#pragma unroll
                for (uint32_t j = 0; j < RPT_GRID_BATCH_ANY; ++j) {
                    if (cand[j]) {                                  // hit_sphere's second half, as in grid_closest_sphere's resolve
                        const float thc = fsqrt(c_rd[j]);
                        const float t0 = c_tca[j] - thc, t1 = c_tca[j] + thc;
                        const float t = t0 < 0.0f ? t1 : t0;
                        if (!(t < 0.0f) && (!use_max || t < max_dist)) occluded = 1u;
                    }
                }
            }
        }
        if (occluded != 0u || last) break;
        if (use_max && t_exit > max_dist) {
            // a sphere entirely beyond max_dist cannot occlude; one straddling this cell was tested
            break;
        }
        k0 = n0; k1 = n1;
    }
    return occluded != 0u;
}

// GeomHit.code of a large scene: the nearest sphere's index in the low 28 bits (kNoSphere: none), the mask of
// accepted planes above.  (rpt_upload_scene refuses tables with 2^28 spheres or more.)
constexpr uint32_t kNoSphere = 0x0FFFFFFFu;

// Geometry pass of AnalyticalScene::closest_hit + Scene::sample_lights, as in dev_integrator.h, for N spheres:
// the part after the sphere loop (dist / best / hit are the loop's result, however it was run).
RPT_DEV bool closest_geom_finish(const SceneLarge& sc, const RayD& ray, PathState& ps, float dist, uint32_t best, bool hit, GeomHit& g, EmitterHit& e)
{
    uint32_t accepted_planes = 0;
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        const DevPlane& p = sc.planes[k];
        float t;
        bool h = hit_plane(ray, p, t);
        bool acc = h && ((sc.n_spheres == 0 && k == 0) || t < dist);
        if (acc) {
            dist = t;
            hit = true;
            accepted_planes |= 1u << k;
        }
    }
    if (hit) ps.hit_dist = dist;
    g.code = (best == 0xFFFFFFFFu ? kNoSphere : best) | (accepted_planes << 28);

    // Scene::sample_lights, scene.rs:65-85
    float ldist = ps.hit_dist;
    if (sc.n_light_spheres == 0xFFFFFFFFu) {
        for (uint32_t i = 0; i < sc.n_lights; ++i) {
            const DevLight L = light_uniform(sc, i);
            hit = light_intersect(L, sc.flags, ray, ps, e, ldist) || hit;
        }
        return hit;
    }
    // The same loop for spherical lights (the other kinds do nothing in it), four lights per trip: one scalar load brings four
    // {centre, radius} records, the first half of the sphere test (scene.rs:39-50) runs for all four without a branch, and only a
    // light the ray's line actually meets — rare — takes the second half and, if it is the nearest so far, fetches the light
    // itself.  Per light the operations and their order are light_intersect's.  (Until round 4 every closest_hit loaded 16 light
    // records of 15 dwords one after the other and waited for each: a fifth of the 10 k-sphere frame's time.)
    uint32_t hit_w = hit ? 1u : 0u;
    RPT_PROF(PB_LIGHTS);
    for (uint32_t i = 0; i < sc.n_light_spheres; i += 4u) {
        cfloat_p rec = (cfloat_p)sc.light_spheres + 4u * i;
        float c_tca[4], c_rd[4];
        bool cand[4];
        bool any_cand = false;
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const v3 l = mk3(rec[4u * j], rec[4u * j + 1u], rec[4u * j + 2u]) - ray.o;
            const float tca = dot3(l, ray.d);
            const float d2 = dot3(l, l) - tca * tca;
            const float radius2 = rec[4u * j + 3u];              // ({centre, r * r}: capi.hip, rpt_upload_scene)
            cand[j] = (i + j < sc.n_light_spheres) && !(d2 > radius2);
            any_cand = any_cand || cand[j];
            c_tca[j] = tca;
            c_rd[j] = radius2 - d2;
        }
        if (any_cand) {
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                if (cand[j]) {
                    const float thc = fsqrt(c_rd[j]);
                    const float t0 = c_tca[j] - thc, t1 = c_tca[j] + thc;       // (t0 <= t1: thc is a square root)
                    const float t = t0 < 0.0f ? t1 : t0;
                    if (!(t < 0.0f) && t < ldist) {                 // light_intersect from here on
                        const DevLight L = light_uniform(sc, ((cuint_p)sc.light_sphere_ids)[i + j]);
                        const v3 pos = mk3(L.px, L.py, L.pz);
                        ldist = t;
                        const v3 hit_point = ray.o + t * ray.d;
                        const float cos_theta = dot3(-ray.d, norm3(hit_point - pos));
                        e.light_pdf = fdiv(ldist * ldist, L.area * cos_theta * 0.5f);
                        e.light_emission = mk3(L.ex, L.ey, L.ez);
                        e.is_emitter = true;
                        ps.hit_dist = t;
                        hit_w = 1u;
                    }
                }
            }
        }
    }
    return hit_w != 0u;
}

RPT_DEV bool closest_geom(const SceneLarge& sc, const RayD& ray, PathState& ps, GeomHit& g, EmitterHit& e)
{
    float dist = 3.40282347e+38f;
    bool hit = false;
    uint32_t best = 0xFFFFFFFFu;                                    // nearest sphere so far
    if (sc.use_accel) grid_closest_sphere(sc, ray, dist, best, hit);
    else brute_closest_sphere(sc, ray, dist, best, hit);
    return closest_geom_finish(sc, ray, ps, dist, best, hit, g, e);
}

// material = Material::new(), the nearest sphere's full patch, then the accepted planes' patches in order
RPT_DEV void material_large(const SceneLarge& sc, const RayD& ray, uint32_t code, Mat& mat)
{
    const uint32_t best = code & kNoSphere;
    const uint32_t accepted_planes = code >> 28;
    mat_defaults(mat);
    if (best != kNoSphere) {                                        // the nearest sphere's full patch
        const DevMaterial m = gather32(sc.materials, gather32(sc.sphere_material, best));
        mat.rgb = mk3(m.rgb[0], m.rgb[1], m.rgb[2]);
        mat.emission = mk3(m.emission[0], m.emission[1], m.emission[2]);
        mat.anisotropic = m.anisotropic; mat.metallic = m.metallic; mat.roughness = m.roughness;
        mat.subsurface = m.subsurface; mat.specular_tint = m.specular_tint; mat.sheen = m.sheen;
        mat.sheen_tint = m.sheen_tint; mat.clearcoat = m.clearcoat; mat.clearcoat_gloss = m.clearcoat_gloss;
        mat.spec_trans = m.spec_trans; mat.ior = m.ior;
    }
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        const DevMaterial pm = material_uniform(sc, sc.planes[k].material);   // wave-uniform patch
        apply_patch(mat, pm, (accepted_planes >> k) & 1u, ray.d);
    }
}

RPT_DEV v3 hit_emission(const SceneLarge& sc, const GeomHit& g)
{
    const uint32_t best = g.code & kNoSphere;
    const uint32_t accepted_planes = g.code >> 28;
    v3 em = mk3(0.0f, 0.0f, 0.0f);
    if (best != kNoSphere) {
        const DevMaterial m = gather32(sc.materials, gather32(sc.sphere_material, best));
        em = mk3(m.emission[0], m.emission[1], m.emission[2]);
    }
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        const DevMaterial pm = material_uniform(sc, sc.planes[k].material);
        apply_patch_emission(em, pm, (accepted_planes >> k) & 1u);
    }
    return em;
}

// Normal of a surface hit at `dist`.
RPT_DEV v3 normal_large(const SceneLarge& sc, const RayD& ray, float dist, const GeomHit& g)
{
    const uint32_t best = g.code & kNoSphere;
    const uint32_t accepted_planes = g.code >> 28;
    v3 pn = mk3(0.0f, 0.0f, 0.0f);
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        const DevPlane& p = sc.planes[k];
        const bool acc = (accepted_planes >> k) & 1u;
        pn.x = acc ? p.nx : pn.x; pn.y = acc ? p.ny : pn.y; pn.z = acc ? p.nz : pn.z;
    }
    const bool win_plane = accepted_planes != 0u;
    v3 c = mk3(0.0f, 0.0f, 0.0f);
    if (!win_plane && best != kNoSphere) {
        const float4 s = gather32(sc.spheres, best);                    // per-lane gather
        c = mk3(s.x, s.y, s.z);
    }
    v3 hp = ray.o + dist * ray.d;
    v3 sn = norm3(hp - c);
    return mk3(win_plane ? pn.x : sn.x, win_plane ? pn.y : sn.y, win_plane ? pn.z : sn.z);
}

// Media (dev_media.h): the material whose Medium the layered material of the hit carries — the nearest sphere's when it
// writes one, then the accepted planes' patches in order.
RPT_DEV uint32_t hit_medium_index(const SceneLarge& sc, const GeomHit& g)
{
    const uint32_t best = g.code & kNoSphere;
    const uint32_t accepted_planes = g.code >> 28;
    uint32_t idx = kNoMediumIdx;
    if (best != kNoSphere) {
        const uint32_t mi = gather32(sc.sphere_material, best);
        if (sc.materials[mi].mask & RPT_MAT_MEDIUM) idx = mi;
    }
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        const uint32_t mi = sc.planes[k].material;
        if (((cuint_p)sc.materials)[29u * mi] & RPT_MAT_MEDIUM) idx = ((accepted_planes >> k) & 1u) ? mi : idx;
    }
    return idx;
}
RPT_DEV DevMedium medium_at(const SceneLarge& sc, uint32_t index) { return medium_of(sc.materials[index]); }   // per-lane gather

RPT_DEV v3 hit_normal(const SceneLarge& sc, const RayD& ray, float dist, const GeomHit& g) { return normal_large(sc, ray, dist, g); }
RPT_DEV void hit_material(const SceneLarge& sc, const RayD& ray, const GeomHit& g, Mat& mat) { material_large(sc, ray, g.code, mat); }
// any_hit after the sphere part (`occluded` = its answer, however it was obtained)
RPT_DEV bool any_hit_finish(const SceneLarge& sc, const RayD& ray, float max_dist, bool occluded)
{
    bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        float t;
        bool h = hit_plane(ray, sc.planes[k], t);
        occluded = occluded || (h && (!use_max || t < max_dist));
    }
    return occluded;
}

RPT_DEV bool any_hit(const SceneLarge& sc, const RayD& ray, float max_dist)
{
    bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;
    bool occluded = sc.use_accel ? grid_any_sphere(sc, ray, use_max, max_dist) : brute_any_sphere(sc, ray, use_max, max_dist);
    for (uint32_t k = 0; k < sc.n_planes; ++k) {
        float t;
        bool h = hit_plane(ray, sc.planes[k], t);
        occluded = occluded || (h && (!use_max || t < max_dist));
    }
    return occluded;
}

}  // namespace RPT_NS
#endif  // this pass
