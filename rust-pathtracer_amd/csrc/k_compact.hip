// k_compact.hip — small scenes, launches of ONE sample per pixel (the reference's own usage: one render() per redraw): the workgroup's
// 256 paths in LDS, re-dealt to its threads before every stage.  Built with the range tests next to every operation (kernel_common.h).
#include "kernel_common.h"

// A workgroup's paths between stages (19 dwords each).
struct WfRecords {
    float f[14][256];          // ray o, d; throughput; radiance; hit_dist; scatter pdf
    uint32_t u[5][256];        // rng state, increment; bounce; GeomHit; sample << 8 | flags | status
};

// PathRegs.bounce and .medium share a dword wherever a path is stored (bounce <= 4096, medium < 2^16); kernels without media
// never look at the upper half.
template <bool MEDIA>
RPT_DEV uint32_t pack_bounce(const PathRegs& p) { return MEDIA ? (p.bounce | (p.medium << 16)) : p.bounce; }
template <bool MEDIA>
RPT_DEV void unpack_bounce(uint32_t v, PathRegs& p)
{
    p.bounce = MEDIA ? (v & 0xFFFFu) : v;
    p.medium = MEDIA ? (v >> 16) : 0u;
}

template <bool MEDIA = false>
RPT_DEV void wf_rec_put(WfRecords& r, uint32_t i, const PathRegs& p, uint32_t gcode, uint32_t ctl)
{
    r.f[0][i] = p.ray.o.x; r.f[1][i] = p.ray.o.y; r.f[2][i] = p.ray.o.z;
    r.f[3][i] = p.ray.d.x; r.f[4][i] = p.ray.d.y; r.f[5][i] = p.ray.d.z;
    r.f[6][i] = p.throughput.x; r.f[7][i] = p.throughput.y; r.f[8][i] = p.throughput.z;
    r.f[9][i] = p.radiance.x; r.f[10][i] = p.radiance.y; r.f[11][i] = p.radiance.z;
    r.f[12][i] = p.ps.hit_dist; r.f[13][i] = p.ps.scatter_pdf;
    r.u[0][i] = p.rng.state; r.u[1][i] = p.rng.inc; r.u[2][i] = pack_bounce<MEDIA>(p); r.u[3][i] = gcode; r.u[4][i] = ctl;
}

template <bool MEDIA = false>
RPT_DEV void wf_rec_get(const WfRecords& r, uint32_t i, PathRegs& p, uint32_t& gcode, uint32_t& ctl)
{
    p.ray.o = mk3(r.f[0][i], r.f[1][i], r.f[2][i]);
    p.ray.d = mk3(r.f[3][i], r.f[4][i], r.f[5][i]);
    p.throughput = mk3(r.f[6][i], r.f[7][i], r.f[8][i]);
    p.radiance = mk3(r.f[9][i], r.f[10][i], r.f[11][i]);
    p.ps.hit_dist = r.f[12][i]; p.ps.scatter_pdf = r.f[13][i];
    p.rng.state = r.u[0][i]; p.rng.inc = r.u[1][i]; unpack_bounce<MEDIA>(r.u[2][i], p); gcode = r.u[3][i]; ctl = r.u[4][i];
}

// append `value` to a workgroup list in LDS for the lanes that `want` (one LDS atomic per wave)
template <class T>
RPT_DEV void wf_list_add(T* list, uint32_t* count, bool want, uint32_t value)
{
    const uint64_t m = __ballot(want);
    if (m == 0ull) return;
    const uint32_t lane = __lane_id();
    const uint32_t leader = (uint32_t)__ffsll((unsigned long long)m) - 1u;
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, (int)leader);
    if (want) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (T)value;
}

// Small scenes, FEW samples per launch (the reference's own usage: one render() per redraw).  With nothing to regenerate a
// wave of the megakernel drains: its lanes end one by one and the wave runs on for its longest path.  Here the workgroup's 256
// paths live in LDS and are re-dealt to the threads before every stage, so TRACE and SHADE always run on full waves (the last one
// of a list excepted) and waves that get nothing issue nothing:
//   TRACE  thread t < |T|: entry t of the trace list: closest_hit; miss / emitter -> blend, the pixel's next sample (if any) -> next T
//                                                                   surface -> S
//   SHADE  thread t < |S|: entry t of the shade list: material, light sample, BSDF; path over -> blend, next sample -> next T;
//                                                                   otherwise -> next T
// Same device functions, same per-pixel order of samples: bit-identical to the other kernels.  Built in the RPT_PEROP_BUILD object: its
// paths change lanes between stages, range trackers would have to travel with them, and its stages are bound by their barriers, not by
// instruction issue (1080p x 1 spp 0.277 ms with trackers carried in the path records, 0.2755 with the tests per operation; 800x600
// 0.0852 against 0.0808).
#ifndef RPT_COMPACT_WAVES_PER_SIMD
#define RPT_COMPACT_WAVES_PER_SIMD 5
#endif
template <class S>
RPT_DEV bool compact_finish(const S& sc, const RenderParams& rp, uint32_t i, PathRegs& p, uint32_t& s)
{
    // tracer.rs:105-117 straight on the pixel in HBM (32 B per sample; keeping the 256 running means in LDS would cost the
    // kernel two of its eight resident workgroups per CU)
    const PixelSetup ps = pixel_setup(rp, i);
    float4* pixel = reinterpret_cast<float4*>(rp.pixels) + ps.pix_offset;
    float4 acc = *pixel;
    const uint64_t frames = rp.frames_done + s;
    blend(acc, p.radiance, 1.0f / (float)(frames + 1));
    *pixel = acc;
    s += 1u;
    if (s >= rp.spp) return false;
    path_begin(sc, p, ps.px, ps.py, frame_key_hd(rp.seed, rp.frames_done + s), ps.pixel_index);
    return true;
}

// Per pass two barriers:  TRACE for the trace list (closest_hit only: surface -> S, miss / emitter -> F)  |  SHADE for the
// entries of S from thread 0 up and, at the same time, FINISH (background for a miss, blend, the pixel's next camera path)
// for the entries of F from thread 255 down — |S| + |F| <= 256, so at most one wave has both kinds.
template <class MS = MaterialPerHit, class S>
RPT_DEV void render_compact_body(const S& sc, const RenderParams& launch, const MS& materials = MS{})
{
    constexpr bool M = S::kMedia;
    RenderParams rp = launch;
    rp.n_chunks = 0u;                                               // (no units: one workgroup per tile, all samples)
    __shared__ WfRecords rec;                                       // u[4] = sample index << 1 | "the ray left the scene"
    __shared__ uint8_t l_trace[2][256], l_shade[256], l_fin[256];   // path = pixel of the tile = thread that started it
    __shared__ uint32_t n_trace[2], n_shade[2], n_fin[2];
    const uint32_t tid = threadIdx.x;
    const PixelSetup ps = pixel_setup(rp);
    if (sc.max_depth == 0) {                                        // no bounce loop: every sample's radiance is zero
        if (ps.valid) {
            float4* pixel = reinterpret_cast<float4*>(rp.pixels) + ps.pix_offset;
            float4 acc = *pixel;
            for (uint32_t k = 0; k < rp.spp; ++k) blend(acc, mk3(0.0f, 0.0f, 0.0f), 1.0f / (float)(rp.frames_done + k + 1));
            *pixel = acc;
        }
        return;
    }
    if (tid < 2u) { n_trace[tid] = 0u; n_shade[tid] = 0u; n_fin[tid] = 0u; }
    const uint32_t t0 = cost_clock();
    __syncthreads();
    if (ps.valid) {
        PathRegs p;
        path_begin(sc, p, ps.px, ps.py, frame_key_hd(rp.seed, rp.frames_done), ps.pixel_index);
        wf_rec_put<M>(rec, tid, p, 0u, 0u);
    }
    wf_list_add(l_trace[0], &n_trace[0], ps.valid, tid);
    __syncthreads();

    for (uint32_t cur = 0u;; cur ^= 1u) {
        const uint32_t n_t = n_trace[cur];
        if (n_t == 0u) break;                                       // (the same value in every thread: read behind a barrier)
        {
            bool to_shade = false, to_fin = false;
            uint32_t i = 0u;
            if (tid < n_t) {
                i = l_trace[cur][tid];
                PathRegs p;
                uint32_t gcode, ctl;
                wf_rec_get<M>(rec, i, p, gcode, ctl);
                GeomHit g;
                g.code = 0u;
                const uint32_t what = path_trace_geom_split(sc, DirectQuery{}, p, g);
                to_shade = what == 2u;
                to_fin = !to_shade;
                wf_rec_put<M>(rec, i, p, g.code, (ctl & ~1u) | (what == 0u ? 1u : 0u));
            }
            wf_list_add(l_shade, &n_shade[cur], to_shade, i);
            wf_list_add(l_fin, &n_fin[cur], to_fin, i);
        }
        __syncthreads();
        if (tid == 0u) { n_trace[cur] = 0u; n_shade[cur ^ 1u] = 0u; n_fin[cur ^ 1u] = 0u; }   // last read before this barrier, next written after the next
        {
            const uint32_t n_s = n_shade[cur], n_f = n_fin[cur];
            bool to_trace = false;
            uint32_t i = 0u;
            if (tid < n_s) {
                i = l_shade[tid];
                PathRegs p;
                uint32_t gcode, ctl;
                wf_rec_get<M>(rec, i, p, gcode, ctl);
                GeomHit g;
                g.code = gcode;
                uint32_t s = ctl >> 1;
                if (path_shade_full(sc, DirectQuery{}, p, g, nullptr, nullptr, materials)) to_trace = compact_finish(sc, rp, i, p, s);
                else to_trace = true;
                if (to_trace) wf_rec_put<M>(rec, i, p, 0u, s << 1);
            } else if (255u - tid < n_f) {
                i = l_fin[255u - tid];
                PathRegs p;
                uint32_t gcode, ctl;
                wf_rec_get<M>(rec, i, p, gcode, ctl);
                uint32_t s = ctl >> 1;
                if (ctl & 1u) p.radiance = p.radiance + background(sc, p.ray) * p.throughput;     // tracer.rs:64-68
                to_trace = compact_finish(sc, rp, i, p, s);
                if (to_trace) wf_rec_put<M>(rec, i, p, 0u, s << 1);
            }
            wf_list_add(l_trace[cur ^ 1u], &n_trace[cur ^ 1u], to_trace, i);
        }
        __syncthreads();
    }
    // the workgroup's time, for the dispatch order of the next launch (its waves end together: one figure for all four)
    if (rp.tile_cost && (tid & 63u) == 0u) rp.tile_cost[block_tile(rp) * 4u + (tid >> 6)] = cost_clock() - t0;
}

__global__ __launch_bounds__(256, RPT_COMPACT_WAVES_PER_SIMD) void RPT_K(render_small_compact_kernel)(const SceneSmall sc, const RenderParams rp) { render_compact_body(kernarg_scene(sc), rp); }
// Frames of a few thousand workgroups (the reference's 800x600 window: 1 875) are a question of how many ROUNDS of workgroups the
// chip needs: six resident per CU make that 1.2 instead of 1.5 rounds.  The price is 80 VGPRs, 35 of the kernel's ~100 live values
// in scratch (116 B per lane, L1/L2-resident at this launch size) — and it is worth it: 800x600 x 1 spp 0.0809 ms against 0.0914
// with five per CU and no spill to speak of (round 4, tools/compact_time.py); from 1080p up the five-per-CU build is 1 % faster.
__global__ __launch_bounds__(256, 6) void RPT_K(render_small_compact_dense_kernel)(const SceneSmall sc, const RenderParams rp) { render_compact_body(kernarg_scene(sc), rp); }
__global__ __launch_bounds__(256, RPT_COMPACT_WAVES_PER_SIMD) void RPT_K(render_small_compact_sized_kernel)(const SceneSmall sc, const RenderParams rp)
{
    render_compact_body(sized_scene<RPT_REFERENCE_SIZES>(kernarg_scene(sc)), rp);
}
__global__ __launch_bounds__(256, 6) void RPT_K(render_small_compact_dense_sized_kernel)(const SceneSmall sc, const RenderParams rp)
{
    render_compact_body(sized_scene<RPT_REFERENCE_SIZES>(kernarg_scene(sc)), rp);
}
// ... with the material table (dev_integrator.h, MaterialTable)
__global__ __launch_bounds__(256, RPT_COMPACT_WAVES_PER_SIMD) void RPT_K(render_small_compact_sized_table_kernel)(const SceneSmall sc, const RenderParams rp)
{
    constexpr uint32_t sizes[3] = {RPT_REFERENCE_SIZES};
    __shared__ float4 s_rows[kMatTableRows * kMatRowFloat4s];
    const SceneSmall& s = sized_scene<RPT_REFERENCE_SIZES>(kernarg_scene(sc));
    render_compact_body(s, rp, material_table_build<false>(s, sizes[0], sizes[1], s_rows));
}
__global__ __launch_bounds__(256, 6) void RPT_K(render_small_compact_dense_sized_table_kernel)(const SceneSmall sc, const RenderParams rp)
{
    constexpr uint32_t sizes[3] = {RPT_REFERENCE_SIZES};
    __shared__ float4 s_rows[kMatTableRows * kMatRowFloat4s];
    const SceneSmall& s = sized_scene<RPT_REFERENCE_SIZES>(kernarg_scene(sc));
    render_compact_body(s, rp, material_table_build<false>(s, sizes[0], sizes[1], s_rows));
}
// ... for any scene of at most three primitives: the table's shape is data
__global__ __launch_bounds__(256, RPT_COMPACT_WAVES_PER_SIMD) void RPT_K(render_small_compact_table_kernel)(const SceneSmall sc, const RenderParams rp)
{
    __shared__ float4 s_rows[kMatTableRows * kMatRowFloat4s];
    const SceneSmall& s = kernarg_scene(sc);
    render_compact_body(s, rp, material_table_build<false>(s, uniform_here(s.n_spheres), uniform_here(s.n_planes), s_rows));
}
__global__ __launch_bounds__(256, 6) void RPT_K(render_small_compact_dense_table_kernel)(const SceneSmall sc, const RenderParams rp)
{
    __shared__ float4 s_rows[kMatTableRows * kMatRowFloat4s];
    const SceneSmall& s = kernarg_scene(sc);
    render_compact_body(s, rp, material_table_build<false>(s, uniform_here(s.n_spheres), uniform_here(s.n_planes), s_rows));
}
#ifndef RPT_RELAXED_BUILD                                           // (media have no relaxed form)
__global__ __launch_bounds__(256, RPT_COMPACT_WAVES_PER_SIMD) void RPT_K(render_small_compact_media_kernel)(const WithMedia<SceneSmall> sc, const RenderParams rp) { render_compact_body(kernarg_scene(sc), rp); }
#endif

namespace RPT_LAUNCH_NS {

hipError_t render_compact(const SceneSmall& sc, bool media, const RenderParams& rp, uint32_t nblocks, hipStream_t st, const KernelChoice& kc)
{
    const dim3 tiles(nblocks), wg(256);
    (void)hipGetLastError();
    const bool dense = nblocks <= 3072u;                             // (six workgroups per CU: see render_small_compact_dense_kernel)
    if (media) {
#ifdef RPT_RELAXED_BUILD
        return hipErrorNotSupported;
#else
        hipLaunchKernelGGL(RPT_K(render_small_compact_media_kernel), tiles, wg, 0, st, WithMedia<SceneSmall>(sc), rp);
#endif
    } else if (dense && kc.sized && kc.material_table) hipLaunchKernelGGL(RPT_K(render_small_compact_dense_sized_table_kernel), tiles, wg, 0, st, sc, rp);
    else if (kc.sized && kc.material_table) hipLaunchKernelGGL(RPT_K(render_small_compact_sized_table_kernel), tiles, wg, 0, st, sc, rp);
    else if (dense && kc.sized) hipLaunchKernelGGL(RPT_K(render_small_compact_dense_sized_kernel), tiles, wg, 0, st, sc, rp);
    else if (kc.sized) hipLaunchKernelGGL(RPT_K(render_small_compact_sized_kernel), tiles, wg, 0, st, sc, rp);
    else if (dense && kc.material_table) hipLaunchKernelGGL(RPT_K(render_small_compact_dense_table_kernel), tiles, wg, 0, st, sc, rp);
    else if (kc.material_table) hipLaunchKernelGGL(RPT_K(render_small_compact_table_kernel), tiles, wg, 0, st, sc, rp);
    else if (dense) hipLaunchKernelGGL(RPT_K(render_small_compact_dense_kernel), tiles, wg, 0, st, sc, rp);
    else hipLaunchKernelGGL(RPT_K(render_small_compact_kernel), tiles, wg, 0, st, sc, rp);
    return hipGetLastError();
}

}  // namespace RPT_LAUNCH_NS
