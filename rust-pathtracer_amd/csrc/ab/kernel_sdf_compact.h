// ab/kernel_sdf_compact.h — A/B only (-DRPT_AB_KERNELS, include/rpt.h RPT_RENDER_SDF_COMPACT): measured slower than the wave
// march kernel (1.69 vs 2.38 Gsamples/s on BASELINE configs[3]).  Included by kernels.hip after the compacting
// kernel of small scenes, whose helpers it uses.
#pragma once

// SDF scenes with the workgroup's 256 paths in LDS, re-dealt to the threads before every pass (the compacting kernel's idea
// with the march as a stage: a march step is uniform code, lanes only differ in WHEN their march ends, so dense lists fill
// the waves that the per-lane state machine below leaves 45 % full).  Per pass, one barrier; every path is in exactly one list:
//   M  marching (path ray or shadow ray): up to `sdf_compact_steps` iterations of sdf_march -> R / S when over, else M again
//   R  closest_hit's acceptance with the march's outcome: miss / emitter -> F; surface: normal, hit point, the shadow ray's
//      analytic part -> M (shadow march) or S
//   S  material, next-event estimation with the shadow march's outcome, BSDF sample -> F (path over) or M (next bounce)
//   F  background of a miss, blend into the pixel in HBM, the pixel's next camera path -> M
// The lists' entries are dealt to the threads one list after the other (each starting on a wave boundary, wrapping round), so a
// wave runs one stage's code on 64 paths.  Same device functions as the other SDF kernels: bit-identical.
struct SdfRecords {
    float f[25][256];          // ray o, d; throughput; radiance; hit_dist; scatter pdf | normal; hit point | march d, t, t_useful
    uint32_t u[7][256];        // rng key, counter; bounce; GeomHit; sample << 2 | miss << 1 | shadow march; march steps; accepted | hit << 31
};
enum : uint32_t { SC_M = 0u, SC_R = 1u, SC_S = 2u, SC_F = 3u };

RPT_DEV void sc_put_path(SdfRecords& r, uint32_t i, const PathRegs& p)
{
    r.f[0][i] = p.ray.o.x; r.f[1][i] = p.ray.o.y; r.f[2][i] = p.ray.o.z;
    r.f[3][i] = p.ray.d.x; r.f[4][i] = p.ray.d.y; r.f[5][i] = p.ray.d.z;
    r.f[6][i] = p.throughput.x; r.f[7][i] = p.throughput.y; r.f[8][i] = p.throughput.z;
    r.f[9][i] = p.radiance.x; r.f[10][i] = p.radiance.y; r.f[11][i] = p.radiance.z;
    r.f[12][i] = p.ps.hit_dist; r.f[13][i] = p.ps.scatter_pdf;
    r.u[0][i] = p.rng.state; r.u[1][i] = p.rng.inc; r.u[2][i] = p.bounce;
}
RPT_DEV void sc_get_path(const SdfRecords& r, uint32_t i, PathRegs& p)
{
    p.ray.o = mk3(r.f[0][i], r.f[1][i], r.f[2][i]);
    p.ray.d = mk3(r.f[3][i], r.f[4][i], r.f[5][i]);
    p.throughput = mk3(r.f[6][i], r.f[7][i], r.f[8][i]);
    p.radiance = mk3(r.f[9][i], r.f[10][i], r.f[11][i]);
    p.ps.hit_dist = r.f[12][i]; p.ps.scatter_pdf = r.f[13][i];
    p.rng.state = r.u[0][i]; p.rng.inc = r.u[1][i]; p.bounce = r.u[2][i];
}
RPT_DEV void sc_put_march(SdfRecords& r, uint32_t i, const MarchRegs& m)
{
    r.f[20][i] = m.d.x; r.f[21][i] = m.d.y; r.f[22][i] = m.d.z; r.f[23][i] = m.t; r.f[24][i] = m.t_useful;
    r.u[5][i] = m.steps; r.u[6][i] = m.accepted | (m.hit ? 0x80000000u : 0u);
}
RPT_DEV void sc_get_march(const SdfRecords& r, uint32_t i, MarchRegs& m)
{
    m.d = mk3(r.f[20][i], r.f[21][i], r.f[22][i]); m.t = r.f[23][i]; m.t_useful = r.f[24][i];
    m.steps = r.u[5][i]; m.accepted = r.u[6][i] & 0x7FFFFFFFu; m.hit = (r.u[6][i] >> 31) != 0u;
}

// (The scene comes through a pointer into device memory, read with scalar loads like the kernarg copy the other kernels use:
// with this much inlined code behind one by-value argument the compiler keeps its prologue copy of the 2 KB struct in scratch.)
__global__ __launch_bounds__(256, 4) void RPT_K(render_sdf_compact_kernel)(const SceneSmallSdf* __restrict__ scene, const RenderParams rp)
{
    const SceneSmallSdf& sc = *(const SceneSmallSdf*)(const RPT_CONST_AS SceneSmallSdf*)scene;
    __shared__ SdfRecords rec;
    __shared__ uint8_t lists[2][4][256];
    __shared__ uint32_t counts[3][4];
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
    if (tid < 12u) counts[tid >> 2][tid & 3u] = 0u;
    __syncthreads();
    if (ps.valid) {
        PathRegs p;
        MarchRegs m;
        path_begin(sc, p, ps.px, ps.py, frame_key_hd(rp.seed, rp.frames_done), ps.pixel_index);
        march_begin_primary(sc, p, m);
        sc_put_path(rec, tid, p);
        sc_put_march(rec, tid, m);
        rec.u[3][tid] = 0u; rec.u[4][tid] = 0u;
    }
    wf_list_add(lists[0][SC_M], &counts[0][SC_M], ps.valid, tid);
    __syncthreads();

    const uint32_t chunk = rp.sdf_compact_steps ? rp.sdf_compact_steps : 8u;
    for (uint32_t pass = 0u;; ++pass) {
        const uint32_t cb = pass % 3u, nb = (pass + 1u) % 3u, zb = (pass + 2u) % 3u;       // counters read / written / cleared this pass
        const uint32_t cl = pass & 1u, nl = cl ^ 1u;
        const uint32_t n_m = counts[cb][SC_M], n_r = counts[cb][SC_R], n_s = counts[cb][SC_S], n_f = counts[cb][SC_F];
        if ((n_m | n_r | n_s | n_f) == 0u) break;                   // (the same in every thread: read behind a barrier)
        if (tid < 4u) counts[zb][tid] = 0u;                         // last read before the previous barrier, next written after this pass's
        const uint32_t b_r = (n_m + 63u) & ~63u, b_s = b_r + ((n_r + 63u) & ~63u), b_f = b_s + ((n_s + 63u) & ~63u);
        bool to_m = false, to_r = false, to_s = false, to_f = false;
        uint32_t dest = 0u;                                         // the path this thread forwards (a thread may serve two lists: the last one wins below, so forward at once)
        // ---- M
        if (tid < n_m) {
            const uint32_t i = lists[cl][SC_M][tid];
            MarchRegs m;
            sc_get_march(rec, i, m);
            const v3 o = mk3(rec.f[0][i], rec.f[1][i], rec.f[2][i]);
            bool over = false;
            for (uint32_t k = 0; k < chunk && !over; ++k) over = march_step(sc.sdf, o, m);
            rec.f[23][i] = m.t; rec.u[5][i] = m.steps; rec.u[6][i] = m.accepted | (m.hit ? 0x80000000u : 0u);
            const bool shadow = (rec.u[4][i] & 1u) != 0u;
            to_m = !over; to_r = over && !shadow; to_s = over && shadow;
            dest = i;
        }
        wf_list_add(lists[nl][SC_M], &counts[nb][SC_M], to_m, dest);
        wf_list_add(lists[nl][SC_R], &counts[nb][SC_R], to_r, dest);
        wf_list_add(lists[nl][SC_S], &counts[nb][SC_S], to_s, dest);
        to_m = to_r = to_s = false;
        // ---- R
        if (((tid - b_r) & 255u) < n_r) {
            const uint32_t i = lists[cl][SC_R][(tid - b_r) & 255u];
            PathRegs p;
            MarchRegs m;
            sc_get_path(rec, i, p);
            sc_get_march(rec, i, m);
            GeomHit g;
            g.code = 0u;
            const SdfInjectedQuery q{{m.hit, m.t}, march_analytic(m)};
            const uint32_t what = path_trace_geom_split(sc, q, p, g);
            uint32_t ctl = rec.u[4][i] & ~3u;
            if (what == 2u) {
                const v3 normal = hit_normal(sc, p.ray, p.ps.hit_dist, g);
                const bool front = (dot3(normal, p.ray.d) <= 0.0f);                     // State::finalize, globals.rs:53-57
                const v3 ffnormal = mk3(front ? normal.x : -normal.x, front ? normal.y : -normal.y, front ? normal.z : -normal.z);
                const v3 fhp = p.ray.o + p.ps.hit_dist * p.ray.d;
                rec.f[14][i] = normal.x; rec.f[15][i] = normal.y; rec.f[16][i] = normal.z;
                rec.f[17][i] = fhp.x; rec.f[18][i] = fhp.y; rec.f[19][i] = fhp.z;
                if (march_begin_shadow(sc, p, fhp, ffnormal, m)) { to_m = true; ctl |= 1u; }
                else to_s = true;
                sc_put_march(rec, i, m);
            } else {
                to_f = true;
                ctl |= (what == 0u) ? 2u : 0u;
            }
            sc_put_path(rec, i, p);
            rec.u[3][i] = g.code; rec.u[4][i] = ctl;
            dest = i;
        }
        wf_list_add(lists[nl][SC_M], &counts[nb][SC_M], to_m, dest);
        wf_list_add(lists[nl][SC_S], &counts[nb][SC_S], to_s, dest);
        wf_list_add(lists[nl][SC_F], &counts[nb][SC_F], to_f, dest);
        to_m = to_s = to_f = false;
        // ---- S
        if (((tid - b_s) & 255u) < n_s) {
            const uint32_t i = lists[cl][SC_S][(tid - b_s) & 255u];
            PathRegs p;
            MarchRegs m;
            sc_get_path(rec, i, p);
            sc_get_march(rec, i, m);
            GeomHit g;
            g.code = rec.u[3][i];
            const v3 normal = mk3(rec.f[14][i], rec.f[15][i], rec.f[16][i]);
            const float4 hitp = make_float4(rec.f[17][i], rec.f[18][i], rec.f[19][i], 0.0f);
            const SdfInjectedQuery q{{m.hit, m.t}, {0.0f, 0u}};
            if (path_shade_full(sc, q, p, g, &normal, &hitp)) {
                to_f = true;
            } else {
                march_begin_primary(sc, p, m);
                sc_put_march(rec, i, m);
                to_m = true;
            }
            sc_put_path(rec, i, p);
            rec.u[4][i] &= ~3u;
            dest = i;
        }
        wf_list_add(lists[nl][SC_M], &counts[nb][SC_M], to_m, dest);
        wf_list_add(lists[nl][SC_F], &counts[nb][SC_F], to_f, dest);
        to_m = to_f = false;
        // ---- F
        if (((tid - b_f) & 255u) < n_f) {
            const uint32_t i = lists[cl][SC_F][(tid - b_f) & 255u];
            PathRegs p;
            sc_get_path(rec, i, p);
            const uint32_t ctl = rec.u[4][i];
            uint32_t s = ctl >> 2;
            if (ctl & 2u) p.radiance = p.radiance + background(sc, p.ray) * p.throughput;     // tracer.rs:64-68
            if (compact_finish(sc, rp, i, p, s)) {
                MarchRegs m;
                march_begin_primary(sc, p, m);
                sc_put_path(rec, i, p);
                sc_put_march(rec, i, m);
                rec.u[4][i] = s << 2;
                to_m = true;
            }
            dest = i;
        }
        wf_list_add(lists[nl][SC_M], &counts[nb][SC_M], to_m, dest);
        __syncthreads();
    }
}

