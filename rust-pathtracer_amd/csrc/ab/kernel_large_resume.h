// A/B only (-DRPT_AB_KERNELS; RPT_LARGE_WALK_CAP=<cells>): measured SLOWER than the shipped large-scene megakernel — 10 k spheres, 2048^2 x 32
// spp: 1 851 / 1 995 / 2 085 / 2 177 / 2 229 / 2 295 Msamples/s at 2 / 3 / 4 / 6 / 8 / 12 cells per pass against 2 876 (shading thresholds
// 32 and 44 instead of 56: 1 947-2 153); every frame bit-identical (profiles/r4/experiments/large_resume_walk.txt).  Two costs, neither
// bought back by the fuller cell iterations: with the walk's state live beside the path's across the pass the kernel spills 52 dwords
// per lane instead of 18 (a cap that is never reached, 12, still loses 20 %), and every cut of a walk is one more pass through the
// state machine — set-up again, the list's first entries requested again, the votes — for rays whose mean walk is 3.8 cells.
// ab/kernel_large_resume.h — included by kernels.hip.
//
// Large scenes with a grid: render_regen_body_tf whose closest-hit walk is a scheduling state of the lane.  Walk lengths are roughly
// exponential (mean 3.8 cells, a wave's longest 12), and inside closest_hit a wave leaves the walk with its LONGEST lane: a cell
// iteration runs with a fifth of the lanes (profiles/r4/block_profile_c5.txt).  Here TRACE walks at most `walk_cap` cells per
// pass; a lane whose walk is not over parks it in LDS — cell, the three exit parameters, t_end, nearest hit so far, guard: 8 dwords;
// the DDA's increments and the first list are derived / requested again — and goes on in the next pass, next to the rays that the
// finishing and shading blocks have set up in the meantime.  Per pixel the arithmetic and its order are the megakernel's (same
// device functions: closest_before_walk, closest_walk_cell, closest_geom_finish), so the image is bit-identical.
RPT_DEV void render_large_resume_body(const SceneLarge& sc, const RenderParams& launch)
{
    RPT_PROF_INIT();
    __shared__ FrameKey s_fkey[kMaxSppPerLaunch];
    __shared__ float s_weight[kMaxSppPerLaunch];
    __shared__ float4 s_acc[256];
    __shared__ float4 s_pix[256];
    __shared__ float4 s_walk[256];                                  // a parked walk: cell (10 bits per axis, + 1), t at which the ray leaves it along x, y, z
    __shared__ float4 s_near[256];                                  // ... t_end, nearest sphere so far (dist, best), guard
    const uint32_t tid = threadIdx.x;
    RenderParams rp;                                                // this workgroup's unit of the launch
    if (!lane_setup(LaneTables{s_fkey, s_weight, s_acc, s_pix}, sc.max_depth, launch, rp)) return;
    constexpr uint32_t ST_WALK = 5u;
    static_assert(ST_WALK != ST_TRACE && ST_WALK != ST_SHADE && ST_WALK != ST_FINISH && ST_WALK != ST_DONE, "a state of its own");

    uint32_t s = 0;
    uint32_t state = ST_TRACE;
    PathRegs p;
    GeomHit g;
    g.code = 0u;
    {
        const float4 c = s_pix[tid];
        path_begin<true>(sc, p, c.x, c.y, s_fkey[0], rpt_f2u(c.z), rpt_f2u(c.w));
    }

    for (;;) {
        RPT_PROF(PB_PASS);
        if (state == ST_FINISH) {
            RPT_PROF(PB_FINISH);
            float4 acc = s_acc[tid];
            blend(acc, p.radiance, s_weight[s]);
            s_acc[tid] = acc;
            s += 1;
            if (s >= rp.spp) {
                state = ST_DONE;
            } else {
                const float4 c = s_pix[tid];
                path_begin<true>(sc, p, c.x, c.y, s_fkey[s], rpt_f2u(c.z), rpt_f2u(c.w));
                state = ST_TRACE;
            }
        }
        if (state == ST_TRACE || state == ST_WALK) {
            RPT_PROF(PB_TRACE);
            float dist;
            uint32_t best, guard;
            ClosestWalk w;
            uint32_t walking;                                       // (a word: see closest_walk_cell)
            if (state == ST_TRACE) {                                // sphere 0, the oversize spheres, the walk's set-up
                walking = closest_before_walk(sc, p.ray, dist, best) ? 1u : 0u;
                if (walking) {
                    { RPT_PROF(PB_GRID_BEGIN); w.g = grid_begin(sc, p.ray); }
                    walking = w.g.alive ? 1u : 0u;
                }
                guard = grid_walk_guard(sc);
                if (walking) closest_walk_fetch(sc, w, grid_cell_index(sc, w.g));
            } else {                                                // resume: what grid_begin derives from the ray alone is derived again
                const float4 a = s_walk[tid], b = s_near[tid];
                const uint32_t cell = rpt_f2u(a.x);
                w.g.ix = (int)(cell & 1023u) - 1; w.g.iy = (int)((cell >> 10) & 1023u) - 1; w.g.iz = (int)(cell >> 20) - 1;
                w.g.tmx = a.y; w.g.tmy = a.z; w.g.tmz = a.w;
                w.g.t_end = b.x; dist = b.y; best = rpt_f2u(b.z); guard = rpt_f2u(b.w);
                grid_increments(sc, p.ray, w.g);
                closest_walk_fetch(sc, w, grid_cell_index_clamped(sc, w.g));
                walking = 1u;
            }
            uint32_t hit_w = best != 0xFFFFFFFFu ? 1u : 0u;
            for (uint32_t it = 0; it < rp.large_walk_cap; ++it) {
                if (__ballot(walking != 0u) == 0ull) break;
                if (walking) {
                    guard -= 1u;
                    if (closest_walk_cell(sc, p.ray, w, dist, best, hit_w) || guard == 0u) walking = 0u;
                }
            }
            if (walking) {                                          // park (w.g is the cell whose list comes next)
                const int nx = (int)sc.gn[0], ny = (int)sc.gn[1], nz = (int)sc.gn[2];
                const int cx = w.g.ix < -1 ? -1 : (w.g.ix > nx ? nx : w.g.ix), cy = w.g.iy < -1 ? -1 : (w.g.iy > ny ? ny : w.g.iy),
                          cz = w.g.iz < -1 ? -1 : (w.g.iz > nz ? nz : w.g.iz);         // (a cell outside the grid is answered with any cell: grid_advance)
                s_walk[tid] = make_float4(rpt_u2f((uint32_t)(cx + 1) | ((uint32_t)(cy + 1) << 10) | ((uint32_t)(cz + 1) << 20)), w.g.tmx, w.g.tmy, w.g.tmz);
                s_near[tid] = make_float4(w.g.t_end, dist, rpt_u2f(best), rpt_u2f(guard));
                state = ST_WALK;
            } else {
                state = path_trace_geom(sc, WaveQuery{dist, best}, p, g) ? ST_SHADE : ST_FINISH;
            }
        }
        const uint64_t m_shade = __ballot(state == ST_SHADE);
        const uint64_t m_go = __ballot(state == ST_TRACE || state == ST_FINISH || state == ST_WALK);
        if ((m_shade | m_go) == 0ull) break;
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

__global__ __launch_bounds__(256, RPT_LARGE_WAVES_PER_SIMD) void RPT_K(render_large_resume_kernel)(const SceneLarge sc, const RenderParams rp) { render_large_resume_body(sc, rp); }
