// dev_probes.h — the bodies of the test probes (include/rpt.h: rpt_probe_math, rpt_probe_fn, rpt_probe_rays): one library function per
// record, the same device functions the megakernels inline.  Re-includable like the headers it uses (dev_math.h, "two passes"): the probe kernels
// (k_probes.hip) run the normal pass and, for a record whose operands left the range of the short divide / square root (dev_math.h, range
// trackers), the plain pass — as the render kernels do per sample.
#ifndef RPT_NS                        // (the namespace of this pass: dev_math.h, "two passes")
#define RPT_NS rptdev
#endif
#if (defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_PROBES_H_PLAIN)) || (!defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_PROBES_H_NORMAL))
#ifdef RPT_PLAIN_PASS
#define RPT_DEV_PROBES_H_PLAIN
#else
#define RPT_DEV_PROBES_H_NORMAL
#endif

#include "dev_scene_large.h"

namespace RPT_NS {
using namespace rptscene;

RPT_DEV float probe_math_body(uint32_t fn, float a, float b, uint64_t i)
{
    float r = 0.0f;
    switch (fn) {
    case RPT_PROBE_SIN: r = rpt_sinf(a); break;
    case RPT_PROBE_COS: r = rpt_cosf(a); break;
    case RPT_PROBE_LOG2: r = rpt_log2f(a); break;
    case RPT_PROBE_POW: r = rpt_powf(a, b); break;
    case RPT_PROBE_DIV: r = fdiv(a, b); break;                 // the library's divide (dev_math.h), not hipcc's
    case RPT_PROBE_DIV3: {                                            // three quotients by one denominator (divs3; normalize's own form
                                                                      // takes only len3 of its numerators: the integrator probes cover it)
        const v3 q = (i & 4u) ? divs3(mk3(a, 0.5f * b, 0.0f), b) : divs3(mk3(a, -b, 0.75f * a), b);
        r = (i % 3u == 0u) ? q.x : ((i % 3u == 1u) ? q.y : q.z);
        break;
    }
    case RPT_PROBE_SQRT: r = fsqrt(a); break;                     // the library's square root (dev_math.h)
    case RPT_PROBE_EXP: r = rpt_expf(a); break;
    case RPT_PROBE_LOG: r = rpt_logf(a); break;
    case RPT_PROBE_RNG: {                                            // a = seed bits, b = frame bits, i = pixel; first draw
        Rng rng;
        rng.init(frame_key_hd((uint64_t)rpt_f2u(a), (uint64_t)rpt_f2u(b)), (uint32_t)i);
        r = rng.gen();
        break;
    }
    default: break;
    }
    return r;
}

// One integrator function per record (include/rpt.h, rpt_probe_fn): the same device functions the megakernel inlines.
struct ProbeLightScene {                                             // what sample_light reads of a scene
    float n_lights_f;
    uint32_t flags;
};

RPT_DEV void probe_material(const float* r, Mat& m)
{
    m.rgb = mk3(r[0], r[1], r[2]); m.emission = mk3(r[3], r[4], r[5]);
    m.anisotropic = r[6]; m.metallic = r[7]; m.roughness = r[8]; m.subsurface = r[9]; m.specular_tint = r[10];
    m.sheen = r[11]; m.sheen_tint = r[12]; m.clearcoat = r[13]; m.clearcoat_gloss = r[14]; m.spec_trans = r[15]; m.ior = r[16];
    m.clearcoat_roughness = 0.0f; m.ax = 0.0f; m.ay = 0.0f;
    mat_finalize(m);
}

RPT_DEV void probe_fn_body(uint32_t fn, const DevCamera& cam, const float* __restrict__ r, float* __restrict__ o)
{
    for (int k = 0; k < RPT_PROBE_OUT_STRIDE; ++k) o[k] = 0.0f;
    switch (fn) {
    case RPT_PROBE_FN_GEN_RAY: {
        const RayD ray = camera_ray(cam, r[0], r[1], r[2], r[3]);
        o[0] = ray.o.x; o[1] = ray.o.y; o[2] = ray.o.z; o[3] = ray.d.x; o[4] = ray.d.y; o[5] = ray.d.z;
        break;
    }
    case RPT_PROBE_FN_HIT_SPHERE: {
        float t = 0.0f;
        const bool h = hit_sphere(RayD{mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5])}, mk3(r[6], r[7], r[8]), r[9], t);
        o[0] = h ? 1.0f : 0.0f; o[1] = h ? t : 0.0f;
        break;
    }
    case RPT_PROBE_FN_HIT_PLANE: {
        float t = 0.0f;
        const DevPlane p{r[6], r[7], r[8], r[9], r[10], r[11], r[12], 0u, r[13]};
        const bool h = hit_plane(RayD{mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5])}, p, t);
        o[0] = h ? 1.0f : 0.0f; o[1] = h ? t : 0.0f;
        break;
    }
    case RPT_PROBE_FN_SAMPLE_LIGHT: {
        const DevLight L{rpt_f2u(r[0]), r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11], r[12], r[13], r[14]};
        const ProbeLightScene sc{r[18], rpt_f2u(r[19])};
        Rng rng;
        rng.state = rpt_f2u(r[20]);
        rng.inc = rpt_f2u(r[21]) | 1u;
        const Rng rng0 = rng;
        LightSample ls;
        sample_light(sc, L, mk3(r[15], r[16], r[17]), ls, rng);
        o[0] = ls.normal.x; o[1] = ls.normal.y; o[2] = ls.normal.z;
        o[3] = ls.emission.x; o[4] = ls.emission.y; o[5] = ls.emission.z;
        o[6] = ls.direction.x; o[7] = ls.direction.y; o[8] = ls.direction.z;
        o[9] = ls.dist; o[10] = ls.pdf; o[11] = (float)rng_draws_between(rng0, rng);
        break;
    }
    case RPT_PROBE_FN_DISNEY_EVAL: {
        Mat m;
        probe_material(r, m);
        const float eta = r[17];
        const v3 v = mk3(r[18], r[19], r[20]), nn = mk3(r[21], r[22], r[23]), l = mk3(r[24], r[25], r[26]);
        const ShadeFrame fr = make_frame(m, eta, v, nn);
        float pdf;
        const v3 f = disney_eval(m, eta, fr, nn, l, pdf);
        o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = pdf;
        break;
    }
    case RPT_PROBE_FN_DISNEY_SAMPLE: {
        Mat m;
        probe_material(r, m);
        const float eta = r[17];
        const v3 v = mk3(r[18], r[19], r[20]), nn = mk3(r[21], r[22], r[23]);
        v3 l = mk3(r[24], r[25], r[26]);
        Rng rng;
        rng.state = rpt_f2u(r[27]);
        rng.inc = rpt_f2u(r[28]) | 1u;
        const Rng rng0 = rng;
        const ShadeFrame fr = make_frame(m, eta, v, nn);
        float pdf;
        const v3 f = disney_sample(m, eta, fr, nn, l, pdf, rng);
        o[0] = f.x; o[1] = f.y; o[2] = f.z; o[3] = l.x; o[4] = l.y; o[5] = l.z; o[6] = pdf; o[7] = (float)rng_draws_between(rng0, rng);
        break;
    }
    default: break;
    }
}

RPT_DEV void probe_rays_body(const SceneLarge& sc, const float* __restrict__ r, uint32_t* __restrict__ out, uint64_t i)
{
    RayD ray{mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5])};
    float dist = 3.40282347e+38f;
    uint32_t best = 0xFFFFFFFFu;
    bool hit = false;
    bool any;
    if (sc.use_accel) {
        grid_closest_sphere(sc, ray, dist, best, hit);
        any = grid_any_sphere(sc, ray, true, r[6]);
    } else {
        any = false;
        for (uint32_t k = 0; k < sc.n_spheres; ++k) {
            const float4 s = sphere_uniform(sc, k);
            float t;
            bool h = hit_sphere(ray, mk3(s.x, s.y, s.z), s.w, t);
            if (h && (k == 0 || t < dist)) { dist = t; best = k; hit = true; }
            any = any || (h && t < r[6]);
        }
    }
    out[i * 3 + 0] = rpt_f2u(dist);
    out[i * 3 + 1] = best;
    out[i * 3 + 2] = any ? 1u : 0u;
}

}  // namespace RPT_NS
#endif  // this pass
