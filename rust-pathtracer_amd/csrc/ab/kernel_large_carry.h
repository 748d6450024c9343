// A/B only (-DRPT_AB_KERNELS; RPT_LARGE_MEGA=carry): measured SLOWER than the shipped large-scene megakernel — 10 k spheres, 2048^2 x 32
// spp: 1 217 Msamples/s (leave-for-the-block thresholds 32-56 lanes, refill votes 6-24: 1 021-1 166) against 1 862; every frame
// bit-identical (profiles/r3/experiments/large_carry_walk.txt).  The schedule does what it was built for — a cell iteration runs with
// 46-54 % of the lanes instead of 20 %, a sample takes 26-31 wave iterations instead of 69, the block runs 3.7 times per sample with
// 70 % of the lanes — but an iteration costs 2.4x as much: the texture path (TA 75 % busy, TD 92 %, profiles/r3/c5_mem_counters/)
// serves lanes, not waves, a fuller wave waits for the longest of more lists, and the state the block keeps alive costs 99 spilled
// VGPRs whose traffic goes down the same path (+45 % L1 accesses per launch).
// ab/kernel_large_carry.h — large scenes with a grid: the megakernel whose grid walks are a scheduling state of the lane.
// Included by kernels.hip.
//
// render_large_regen_kernel walks inside closest_hit / any_hit: the wave leaves a walk when its LONGEST lane does, and walk
// lengths are roughly exponential, so a cell iteration runs with a fifth of the lanes (profiles/NOTES.md 4b).  Here a lane's
// walk survives the loop that runs it:
//   * WALK phase: ONE loop steps every walking lane one cell per iteration, shadow rays and path rays alike (walk_cell2: the
//     specialised loops' cell body with the ray kind as a lane flag).  A bounce's two rays are walked back to back — the parked
//     shadow ray of next-event estimation one bounce late, then the path ray (DeferredQuery, the wavefront form's own) — and the lanes
//     whose shadow walk has ended set up their path walks together: when `walk_refill_at` of them have gathered or fewer than
//     `carry_walk_min` lanes still walk.
//   * The wave leaves the loop as soon as `carry_wait_at` / 64 of its live lanes are through both walks:
//     the stragglers PARK their walk in LDS (cell, the three exit parameters, t_end, nearest hit so far, guard: 8 dwords; the
//     DDA increments are recomputed from the ray, three correctly rounded divides) and resume in the next WALK phase next to the
//     fresh rays.  Nothing of a walk is live in registers across the block.
//   * BLOCK (one, as in render_sdf_march2_kernel) for the lanes that wait: add the parked light sample if its shadow ray got
//     through — before anything else touches the radiance, so the additions and their order are the reference's —, finish
//     closest_hit, then miss / emitter -> blend and the pixel's next camera path, or surface -> material, light sample (parked),
//     BSDF, next ray; sphere 0 / oversize spheres of the new path ray.
// Per pixel the arithmetic and its order are the megakernel's (same device functions), so the image is bit-identical.
#ifndef RPT_LARGE_CARRY_WAVES_PER_SIMD
#define RPT_LARGE_CARRY_WAVES_PER_SIMD 5
#endif
#ifndef RPT_CARRY_BATCH
#define RPT_CARRY_BATCH 2          // list entries per trip
#endif

// One cell of grid_closest_sphere / grid_any_sphere (dev_scene_large.h) for the ray a lane is walking; true: the walk is over.
RPT_DEV bool walk_cell2(const SceneLarge& sc, const RayD& ray, GridWalk& g, uint32_t& k0, uint32_t& k1, uint32_t& guard, bool shadow, bool use_max,
                        float max_dist, float& dist, uint32_t& best, bool& occluded)
{
    const float t_exit = grid_cell_exit(g);
    grid_step(sc, g);                                               // g is the NEXT cell from here on
    uint32_t n0 = 0, n1 = 0;
    if (g.alive) cell_bounds(sc, grid_cell_index(sc, g), n0, n1);
    bool done = false;
    float c_tca = 0.0f, c_rd = 0.0f;
    uint32_t c_k = 0u;
    bool cand_parked = false;
    auto resolve = [&](float tca, float rd, uint32_t kk) {          // hit_sphere's second half + acceptance
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
    for (uint32_t k = k0; k < k1; k += RPT_CARRY_BATCH) {
        float4 sp[RPT_CARRY_BATCH];
#pragma unroll
        for (uint32_t i = 0; i < RPT_CARRY_BATCH; ++i) {
            sp[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (k + i < k1) sp[i] = sc.cell_spheres[k + i];
        }
#pragma unroll
        for (uint32_t i = 0; i < RPT_CARRY_BATCH; ++i) {
            const v3 l = mk3(sp[i].x, sp[i].y, sp[i].z) - ray.o;    // hit_sphere's first half
            const float tca = dot3(l, ray.d);
            const float d2 = dot3(l, l) - tca * tca;
            const float radius2 = sp[i].w * sp[i].w;
            if ((k + i < k1) && !(d2 > radius2)) {
                if (!cand_parked) { c_tca = tca; c_rd = radius2 - d2; c_k = k + i; cand_parked = true; }
                else resolve(tca, radius2 - d2, k + i);
            }
        }
    }
    if (cand_parked) resolve(c_tca, c_rd, c_k);
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

template <class S>
RPT_DEV void render_large_carry_body(const S& sc, const RenderParams& launch)
{
    RPT_PROF_INIT();
    __shared__ FrameKey s_fkey[kMaxSppPerLaunch];
    __shared__ float s_weight[kMaxSppPerLaunch];
    __shared__ float4 s_acc[256];
    __shared__ float4 s_pix[256];
    __shared__ float4 s_sho[256], s_shd[256], s_gain[256];          // the parked shadow ray (o.w: max_dist; d.w: a parked walk's t_end) and light
                                                                    // sample (w: a parked walk's guard) of each lane
    __shared__ float4 s_walk[256];                                  // a parked walk: cell (8 bits per axis), t at which the ray leaves it along x, y, z
    __shared__ float2 s_near[256];                                  // the path ray's nearest sphere so far: dist, best
    const uint32_t tid = threadIdx.x;
    RenderParams rp;                                                // this workgroup's unit of the launch
    if (!lane_setup(LaneTables{s_fkey, s_weight, s_acc, s_pix}, sc.max_depth, launch, rp)) return;
    const bool use_max = (sc.flags & RPT_SCENE_ANYHIT_USES_MAX_DIST) != 0;

    uint32_t s = 0;
    PathRegs p;
    bool live = true;                                               // the pixel still has samples to go
    bool pending = false;                                           // a light sample is parked, its shadow ray not answered yet
    bool occluded = false;                                          // ... answered: blocked
    bool ending = false;                                            // the path is over once the parked sample is resolved
    bool parked = false;                                            // a walk is parked in LDS
    bool shadow = false;                                            // the walk in flight / parked is the shadow ray's
    bool has_path = false;                                          // the path ray needs a walk (closest_before_walk)
    uint32_t next;                                                  // the ray to set up next: 0 shadow, 1 path, 2 none (with no walk in flight: waiting for the block)
    {
        const float4 c = s_pix[tid];
        path_begin<true>(sc, p, c.x, c.y, s_fkey[0], rpt_f2u(c.z), rpt_f2u(c.w));
        float dist;
        uint32_t best;
        has_path = closest_before_walk(sc, p.ray, dist, best);
        s_near[tid] = make_float2(dist, rpt_u2f(best));
        next = has_path ? 1u : 2u;
    }

    for (;;) {
        RPT_PROF(PB_PASS);
        float dist = 0.0f;
        uint32_t best = 0u;
        {   // ---- WALK phase
            bool has = false;
            bool touched = parked || next < 2u;
            RayD ray{mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 0.0f)};
            float sh_max = 0.0f;
            uint32_t k0 = 0, k1 = 0, guard = 0;
            GridWalk g;
            g.alive = false;
            if (touched) { const float2 nb = s_near[tid]; dist = nb.x; best = rpt_f2u(nb.y); }
            if (parked) {                                           // resume: everything grid_begin derives from the ray is derived again
                const float4 w = s_walk[tid];
                const float4 sd = s_shd[tid];
                if (shadow) {
                    const float4 so = s_sho[tid];
                    ray.o = mk3(so.x, so.y, so.z); ray.d = mk3(sd.x, sd.y, sd.z); sh_max = so.w;
                } else {
                    ray = p.ray;
                }
                const uint32_t cell = rpt_f2u(w.x);
                g.ix = (int)(cell & 255u); g.iy = (int)((cell >> 8) & 255u); g.iz = (int)(cell >> 16);
                g.tmx = w.y; g.tmy = w.z; g.tmz = w.w;
                g.t_end = sd.w;
                guard = rpt_f2u(s_gain[tid].w);
                const float d[3] = {ray.d.x, ray.d.y, ray.d.z};
                int step[3];
                float tdel[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    if (d[a] > 0.0f) { step[a] = 1; tdel[a] = fdiv(sc.cell_size[a], d[a]); }
                    else if (d[a] < 0.0f) { step[a] = -1; tdel[a] = fdiv(-sc.cell_size[a], d[a]); }
                    else { step[a] = 0; tdel[a] = 3.40282347e+38f; }
                }
                g.sx = step[0]; g.sy = step[1]; g.sz = step[2];
                g.tdx = tdel[0]; g.tdy = tdel[1]; g.tdz = tdel[2];
                g.alive = true;
                g.coff = grid_tier(sc, ray);
                cell_bounds(sc, grid_cell_index(sc, g), k0, k1);
                has = true;
                parked = false;
            }
            bool first = true;
            for (;;) {
                const uint32_t n_has = (uint32_t)__popcll(__ballot(has));
                const uint32_t n_more = (uint32_t)__popcll(__ballot(!has && next < 2u));
                // lanes between two walks set up the next one together: when enough of them have gathered, or the loop runs low
                if (n_more != 0u && (first || n_more >= rp.walk_refill_at || n_has < rp.carry_walk_min)) {
                    RPT_PROF(PB_GRID_BEGIN);
                    if (!has && next < 2u) {
                        shadow = next == 0u;
                        if (shadow) {
                            const float4 so = s_sho[tid], sd = s_shd[tid];
                            ray.o = mk3(so.x, so.y, so.z); ray.d = mk3(sd.x, sd.y, sd.z); sh_max = so.w;
                        } else {
                            ray = p.ray;
                        }
                        next = (shadow && has_path) ? 1u : 2u;
                        g = grid_begin(sc, ray);
                        if (g.alive) {
                            has = true;
                            cell_bounds(sc, grid_cell_index(sc, g), k0, k1);
                            guard = sc.gn[0] + sc.gn[1] + sc.gn[2] + 3u;
                        }                                           // else: the ray misses the grid, nothing to walk
                    }
                    first = false;
                    continue;
                }
                first = false;
                if (n_has == 0u) break;                             // (nobody is left to set up either: that would have been done above)
                // leave for the block when enough of the wave's live lanes wait for it
                const uint32_t n_wait = (uint32_t)__popcll(__ballot(live && !has && next == 2u));
                const uint32_t n_live = (uint32_t)__popcll(__ballot(live));
                if (n_wait * 64u >= rp.carry_wait_at * n_live) break;
                if (has) {
                    RPT_PROF(PB_GRID_CELL);
                    if (walk_cell2(sc, ray, g, k0, k1, guard, shadow, use_max, sh_max, dist, best, occluded)) has = false;
                }
            }
            if (has) {                                              // a straggler: park the walk (g is the cell whose list is [k0, k1))
                s_walk[tid] = make_float4(rpt_u2f((uint32_t)g.ix | ((uint32_t)g.iy << 8) | ((uint32_t)g.iz << 16)), g.tmx, g.tmy, g.tmz);
                s_shd[tid].w = g.t_end;
                s_gain[tid].w = rpt_u2f(guard);
                parked = true;
            }
            if (touched) s_near[tid] = make_float2(dist, rpt_u2f(best));
        }
        // ---- BLOCK for the lanes that are through their walks
        const bool mine = live && !parked && next == 2u;
        if (__ballot(live) == 0ull) break;
        if (mine) {
            RPT_PROF(PB_SHADE);
            if (pending) {                                          // last bounce's light sample: visible unless its walk found an occluder
                if (!occluded) { const float4 gn = s_gain[tid]; p.radiance = p.radiance + mk3(gn.x, gn.y, gn.z); }
                pending = false;
            }
            occluded = false;
            bool over = ending;
            ending = false;
            if (!over) {
                const float2 nb = s_near[tid];
                GeomHit gh;
                gh.code = 0u;
                const WaveQuery q{nb.x, rpt_f2u(nb.y)};
                const uint32_t what = path_trace_geom_split(sc, q, p, gh);
                if (what == 0u) { p.radiance = p.radiance + background(sc, p.ray) * p.throughput; over = true; }
                else if (what == 1u) over = true;
                else {
                    ShadowReq sr;
                    over = path_shade_deferred(sc, p, gh, sr);
                    if (sr.pending) {
                        s_sho[tid] = make_float4(sr.ray.o.x, sr.ray.o.y, sr.ray.o.z, sr.max_dist);
                        s_shd[tid] = make_float4(sr.ray.d.x, sr.ray.d.y, sr.ray.d.z, 0.0f);
                        s_gain[tid] = make_float4(sr.c_lit.x, sr.c_lit.y, sr.c_lit.z, 0.0f);
                        pending = true;
                    }
                }
            }
            bool new_ray = !over;
            ending = pending && over;
            if (over && !pending) {                                 // blend, next sample of the pixel (or retire)
                RPT_PROF(PB_FINISH);
                float4 acc = s_acc[tid];
                blend(acc, p.radiance, s_weight[s]);
                s_acc[tid] = acc;
                s += 1;
                if (s >= rp.spp) {
                    live = false;
                } else {
                    const float4 c = s_pix[tid];
                    path_begin<true>(sc, p, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w));
                    new_ray = true;
                }
            }
            has_path = false;
            if (new_ray) {                                          // sphere 0, the oversize spheres (or the whole brute-force loop: then no walk)
                float nd;
                uint32_t nb2;
                has_path = closest_before_walk(sc, p.ray, nd, nb2);
                s_near[tid] = make_float2(nd, rpt_u2f(nb2));
            }
            next = pending ? 0u : (has_path ? 1u : 2u);
        }
    }
    RPT_PROF_FLUSH();
    lane_finish(rp, s_acc[tid]);
}

__global__ __launch_bounds__(256, RPT_LARGE_CARRY_WAVES_PER_SIMD) void RPT_K(render_large_carry_kernel)(const SceneLarge sc, const RenderParams rp) { render_large_carry_body(sc, rp); }
