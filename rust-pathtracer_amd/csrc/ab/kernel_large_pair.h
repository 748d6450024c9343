// A/B only (-DRPT_AB_KERNELS; RPT_LARGE_MEGA=pair): measured SLOWER than the shipped large-scene megakernel — 10 k spheres, 2048^2 x 32
// spp: 1 257-1 316 Msamples/s at refill thresholds 8-48 (4 / 5 / 6 waves per SIMD: 1 153 / 1 316 / 1 294) against 1 675
// (profiles/r3/experiments/large_pair_walk.txt); every frame bit-identical.  Fusing the two walks of a bounce does raise the lane
// utilisation of the cell iterations, but it needs the walk kernel's generic one-cell-per-iteration loop (four-entry batches,
// refill votes, a shadow / path switch per lane), which costs ~2x the instructions per cell of the megakernel's two specialised loops.
// Included by kernels.hip.
// Large scenes with a grid: the megakernel with the two grid walks of a bounce in one loop (dev_wavefront.h, grid_walk_pair).
// The shadow ray of next-event estimation is walked one bounce LATE, right before the next path ray: SHADE computes the light
// sample's contribution as if the light were visible and parks it (DeferredQuery, the wavefront form's own); TRACE walks the
// parked shadow ray and the path ray back to back, adds the parked contribution if the shadow ray got through — before anything
// else touches the radiance, so the additions and their order are the reference's — and finishes closest_hit.  A path that ends
// with a light sample parked goes through TRACE once more (shadow walk only) before its sample is blended.
#ifndef RPT_LARGE_PAIR_WAVES_PER_SIMD
#define RPT_LARGE_PAIR_WAVES_PER_SIMD 5
#endif
template <class S>
RPT_DEV void render_large_pair_body(const S& sc, const RenderParams& launch)
{
    RPT_PROF_INIT();
    __shared__ FrameKey s_fkey[kMaxSppPerLaunch];
    __shared__ float s_weight[kMaxSppPerLaunch];
    __shared__ float4 s_acc[256];
    __shared__ float4 s_pix[256];
    __shared__ float4 s_sho[256], s_shd[256], s_gain[256];          // the parked shadow ray (o.w: max_dist) and light sample of each lane
    const uint32_t tid = threadIdx.x;
    RenderParams rp;                                                // this workgroup's unit of the launch
    if (!lane_setup(LaneTables{s_fkey, s_weight, s_acc, s_pix}, sc.max_depth, launch, rp)) return;

    uint32_t s = 0;
    uint32_t state = ST_TRACE;
    PathRegs p;
    GeomHit g;
    g.code = 0u;
    bool pending = false;                                           // a light sample is parked, its shadow ray not walked yet
    bool ending = false;                                            // the path is over once the parked sample is resolved
    {
        const float4 c = s_pix[tid];
        path_begin<true>(sc, p, c.x, c.y, s_fkey[0], rpt_f2u(c.z), rpt_f2u(c.w));
    }

    for (;;) {
        RPT_PROF(PB_PASS);
        if (__ballot(state == ST_TRACE) != 0ull) {
            RPT_PROF(PB_TRACE);
            const bool tracing = state == ST_TRACE;
            float dist = 3.40282347e+38f;
            uint32_t best = 0xFFFFFFFFu;
            bool walk_path = false;
            if (tracing && !ending) walk_path = closest_before_walk(sc, p.ray, dist, best);
            const bool walk_shadow = tracing && pending;
            RayD sh{mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 0.0f)};
            float sh_max = 0.0f;
            if (walk_shadow) {
                const float4 so = s_sho[tid], sd = s_shd[tid];
                sh.o = mk3(so.x, so.y, so.z); sh.d = mk3(sd.x, sd.y, sd.z); sh_max = so.w;
            }
            bool occluded = false;
            grid_walk_pair(sc, walk_shadow, sh, sh_max, occluded, walk_path, p.ray, dist, best, rp.walk_refill_at);
            if (tracing) {
                if (pending) {                                      // last bounce's light sample: visible unless its walk found an occluder
                    if (!occluded) { const float4 gn = s_gain[tid]; p.radiance = p.radiance + mk3(gn.x, gn.y, gn.z); }
                    pending = false;
                }
                if (ending) {
                    ending = false;
                    state = ST_FINISH;
                } else {
                    const WaveQuery q{dist, best};
                    const uint32_t what = path_trace_geom_split(sc, q, p, g);
                    state = (what == 2u) ? ST_SHADE : ((what == 0u) ? ST_MISS : ST_FINISH);
                }
            }
        }
        const uint32_t n_shade = (uint32_t)__popcll(__ballot(state == ST_SHADE));
        const uint32_t n_fin = (uint32_t)__popcll(__ballot(state >= ST_FINISH));
        if ((n_shade | n_fin) == 0u) break;
        if (n_shade >= rp.shade_threshold || (n_fin < rp.finish_threshold && n_shade >= n_fin)) {
            if (state == ST_SHADE) {
                RPT_PROF(PB_SHADE);
                ShadowReq sr;
                const bool over = path_shade_deferred(sc, p, g, sr);
                if (sr.pending) {
                    s_sho[tid] = make_float4(sr.ray.o.x, sr.ray.o.y, sr.ray.o.z, sr.max_dist);
                    s_shd[tid] = make_float4(sr.ray.d.x, sr.ray.d.y, sr.ray.d.z, 0.0f);
                    s_gain[tid] = make_float4(sr.c_lit.x, sr.c_lit.y, sr.c_lit.z, 0.0f);
                    pending = true;
                }
                ending = over && pending;
                state = (over && !pending) ? ST_FINISH : ST_TRACE;
            }
        } else if (state >= ST_FINISH) {
            if (state == ST_MISS) {
                RPT_PROF(PB_BACKGROUND);
                p.radiance = p.radiance + background(sc, p.ray) * p.throughput;
            }
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
    }
    RPT_PROF_FLUSH();
    lane_finish(rp, s_acc[tid]);
}

__global__ __launch_bounds__(256, RPT_LARGE_PAIR_WAVES_PER_SIMD) void RPT_K(render_large_pair_kernel)(const SceneLarge sc, const RenderParams rp) { render_large_pair_body(sc, rp); }

