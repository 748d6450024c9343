// dev_bsdf.h — Disney principled BSDF of rust-pathtracer/src/tracer.rs:223-626 and
// spherical-light sampling (tracer.rs:173-220), device side.  Operation order is
// the reference's; see dev_math.h for why.
#ifndef RPT_NS                        // (the namespace of this pass: dev_math.h, "two passes")
#define RPT_NS rptdev
#endif
#if (defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_BSDF_H_PLAIN)) || (!defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_BSDF_H_NORMAL))
#ifdef RPT_PLAIN_PASS
#define RPT_DEV_BSDF_H_PLAIN
#else
#define RPT_DEV_BSDF_H_NORMAL
#endif

#include "dev_math.h"

namespace RPT_NS {
using namespace rptscene;

// Material after Material::new() + patches + finalize() (material.rs:82-131).
struct Mat {
    v3 rgb, emission;
    float anisotropic, metallic, roughness, subsurface, specular_tint, sheen, sheen_tint;
    float clearcoat, clearcoat_gloss, clearcoat_roughness, spec_trans, ior, ax, ay;
};

RPT_DEV void mat_defaults(Mat& m)                                   // material.rs:82-114
{
    m.rgb = mk3(1.5f, 1.5f, 1.5f);
    m.emission = mk3(0.0f, 0.0f, 0.0f);
    m.anisotropic = 0.0f; m.metallic = 0.0f; m.roughness = 0.5f; m.subsurface = 0.0f;
    m.specular_tint = 0.0f; m.sheen = 0.0f; m.sheen_tint = 0.0f;
    m.clearcoat = 0.0f; m.clearcoat_gloss = 0.0f; m.clearcoat_roughness = 0.0f;
    m.spec_trans = 0.0f; m.ior = 1.45f; m.ax = 0.0f; m.ay = 0.0f;
}

RPT_DEV void mat_finalize(Mat& m)                                   // material.rs:117-131
{
    m.roughness = rmax(m.roughness, 0.01f);
    m.clearcoat_roughness = mixf(0.1f, 0.001f, m.clearcoat_gloss);
    float aspect = fsqrt(1.0f - m.anisotropic * 0.9f);
    m.ax = rmax(fdiv(m.roughness, aspect), 0.001f);
    m.ay = rmax(m.roughness * aspect, 0.001f);
}

RPT_DEV float power_heuristic(float a, float b)                     // tracer.rs:223
{
    float t = a * a;
    return fdiv(t, b * b + t);
}

RPT_DEV float gtr1(float ndoth, float a)                            // tracer.rs:233 (log2, sic)
{
    if (a >= 1.0f) return kInvPi;
    float a2 = a * a;
    float t = 1.0f + (a2 - 1.0f) * ndoth * ndoth;
    return fdiv(a2 - 1.0f, kPi * rpt_log2f(a2) * t);
}

RPT_DEV v3 sample_gtr1(float rgh, float r1)                         // tracer.rs:242 (r2 is unused there)
{
    float a = rmax(0.001f, rgh);
    float a2 = a * a;
    float phi = r1 * kTwoPi;
    float cos_theta = fsqrt(fdiv(1.0f - rpt_powf(a2, 1.0f - r1), 1.0f - a2));
    float sin_theta = clamp01(fsqrt(1.0f - (cos_theta * cos_theta)));
    float sin_phi, cos_phi;
    rpt_sincosf(phi, &sin_phi, &cos_phi);
    return mk3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
}

RPT_DEV v3 sample_ggxvndf(v3 v, float ax, float ay, float r1, float r2)   // tracer.rs:256
{
    v3 vh = norm3(mk3(ax * v.x, ay * v.y, v.z));
    float lensq = vh.x * vh.x + vh.y * vh.y;
    v3 t_1 = mk3(1.0f, 0.0f, 0.0f);
    if (lensq > 0.0f) t_1 = scale3(mk3(-vh.y, vh.x, 0.0f), fdiv(1.0f, fsqrt(lensq)));
    v3 t_2 = cross3(vh, t_1);
    float r = fsqrt(r1);
    float phi = (2.0f * kPi) * r2;
    float sn, cs;
    rpt_sincosf(phi, &sn, &cs);
    float t1 = r * cs;
    float t2 = r * sn;
    float s = 0.5f * (1.0f + vh.z);
    t2 = (1.0f - s) * fsqrt(1.0f - t1 * t1) + s * t2;
    v3 nh = t1 * t_1 + t2 * t_2 + fsqrt(rmax(0.0f, 1.0f - t1 * t1 - t2 * t2)) * vh;
    return norm3(mk3(ax * nh.x, ay * nh.y, rmax(0.0f, nh.z)));
}

RPT_DEV float smithg(float ndotv, float alphag)                     // tracer.rs:276
{
    float a = alphag * alphag;
    float b = ndotv * ndotv;
    return fdiv(2.0f * ndotv, ndotv + fsqrt(a + b - a * b));
}

RPT_DEV float luminance(v3 c)                                       // tracer.rs:284
{
    return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z;
}

RPT_DEV float schlick_fresnel(float u)                              // tracer.rs:288
{
    float m = clamp01(1.0f - u);
    float m2 = m * m;
    return m2 * m2 * m;
}

RPT_DEV float gtr2aniso(float ndoth, float hdotx, float hdoty, float ax, float ay)   // tracer.rs:294
{
    float a = fdiv(hdotx, ax);
    float b = fdiv(hdoty, ay);
    float c = a * a + b * b + ndoth * ndoth;
    return fdiv(1.0f, kPi * ax * ay * c * c);
}

RPT_DEV float smithganiso(float ndotv, float vdotx, float vdoty, float ax, float ay)   // tracer.rs:301
{
    float a = vdotx * ax;
    float b = vdoty * ay;
    float c = ndotv;
    return fdiv(2.0f * ndotv, ndotv + fsqrt(a * a + b * b + c * c));
}

RPT_DEV float dielectric_fresnel(float cos_theta_i, float eta)      // tracer.rs:308
{
    float sin_theta_tsq = eta * eta * (1.0f - cos_theta_i * cos_theta_i);
    if (sin_theta_tsq > 1.0f) return 1.0f;
    float cos_theta_t = fsqrt(rmax(1.0f - sin_theta_tsq, 0.0f));
    float rs = fdiv(eta * cos_theta_t - cos_theta_i, eta * cos_theta_t + cos_theta_i);
    float rp = fdiv(eta * cos_theta_i - cos_theta_t, eta * cos_theta_i + cos_theta_t);
    return 0.5f * (rs * rs + rp * rp);
}

RPT_DEV v3 cosine_sample_hemisphere(float r1, float r2)             // tracer.rs:324
{
    float r = fsqrt(r1);
    float phi = kTwoPi * r2;
    float sn, cs;
    rpt_sincosf(phi, &sn, &cs);
    v3 dir;
    dir.x = r * cs;
    dir.y = r * sn;
    dir.z = fsqrt(rmax(0.0f, 1.0f - dir.x * dir.x - dir.y * dir.y));
    return dir;
}

// What a row of a material table holds beyond the finalized material (dev_integrator.h, MaterialTable): the values the code below
// computes from the material ALONE at every hit — twice where disney_eval and disney_sample both do.  The functions that need one ask
// for it through mat_*(m): computed on the spot for a Mat, read for a MatRow; the operations that produced the value are the same.
struct MatRowValues {
    float lum;                        // luminance(rgb)
    float w_diffuse, w_clearcoat;     // get_lobe_probabilities: the diffuse and clearcoat weights before they are normalised
    float one_m_metallic;             // 1 - metallic
    float dm;                         // eval_diffuse: (1 - metallic) * (1 - spec_trans)
    float gtr1_a2m1, gtr1_k;          // gtr1 of a = clearcoat_roughness: a^2 - 1, pi * log2(a^2)
    float cc_a2;                      // sample_gtr1 of a = max(0.001, clearcoat_roughness): a^2
    double cc_log2_a2;                // ... and rpt_log2_core(a^2), the first half of powf(a^2, 1 - r1)
};
// A MatRow is the row's address: every value — the finalized material's fields too — stays in LDS until the code that needs it reads
// it (one ds_read off the one address register).  Kept in registers from the row's fetch on, the derived values cost more in spills
// than they saved (+0.3 %); read where they are used +2.4 %, the two colours likewise +1.1 %, the material's own fields +-0.
constexpr int kMatRowMore = 6;                                       // row layout (dev_integrator.h, material_table_row), in float4s:
constexpr int kMatRowSpecCol = 2 - kMatRowMore, kMatRowSheenCol = 3 - kMatRowMore;   // relative to MatRow::more
struct MatRow {
    const float4* more;               // {lum, w_diffuse, w_clearcoat, one_m_metallic}, {dm, gtr1_a2m1, gtr1_k, cc_a2}
};
// m.field for either kind of material
#define RPT_MAT_FIELD(name, at)                                                                          \
    RPT_DEV float mat_##name(const Mat& m) { return m.name; }                                            \
    RPT_DEV float mat_##name(const MatRow& m) { return ((const float*)(m.more - kMatRowMore))[at]; }
RPT_MAT_FIELD(metallic, 3)
RPT_MAT_FIELD(roughness, 7)
RPT_MAT_FIELD(subsurface, 11)
RPT_MAT_FIELD(sheen, 15)
RPT_MAT_FIELD(clearcoat, 16)
RPT_MAT_FIELD(clearcoat_roughness, 17)
RPT_MAT_FIELD(spec_trans, 18)
RPT_MAT_FIELD(ax, 20)
RPT_MAT_FIELD(ay, 21)
#undef RPT_MAT_FIELD
RPT_DEV v3 mat_rgb(const Mat& m) { return m.rgb; }
RPT_DEV v3 mat_rgb(const MatRow& m) { const float4 c = m.more[-kMatRowMore]; return mk3(c.x, c.y, c.z); }
RPT_DEV float mat_lum(const Mat& m) { return luminance(m.rgb); }
RPT_DEV float mat_lum(const MatRow& m) { return m.more[0].x; }
RPT_DEV float mat_one_m_metallic(const Mat& m) { return 1.0f - m.metallic; }
RPT_DEV float mat_one_m_metallic(const MatRow& m) { return m.more[0].w; }
RPT_DEV float mat_w_diffuse(const Mat& m, float lum) { return lum * (1.0f - m.metallic) * (1.0f - m.spec_trans); }
RPT_DEV float mat_w_diffuse(const MatRow& m, float) { return m.more[0].y; }
RPT_DEV float mat_w_clearcoat(const Mat& m) { return 0.25f * m.clearcoat * (1.0f - m.metallic); }
RPT_DEV float mat_w_clearcoat(const MatRow& m) { return m.more[0].z; }
RPT_DEV float mat_dm(const Mat& m) { return (1.0f - m.metallic) * (1.0f - m.spec_trans); }
RPT_DEV float mat_dm(const MatRow& m) { return m.more[1].x; }
RPT_DEV float mat_gtr1(const Mat& m, float ndoth) { return gtr1(ndoth, m.clearcoat_roughness); }
RPT_DEV float mat_gtr1(const MatRow& m, float ndoth)                // gtr1, tracer.rs:233
{
    if (mat_clearcoat_roughness(m) >= 1.0f) return kInvPi;
    const float4 r = m.more[1];
    float t = 1.0f + r.y * ndoth * ndoth;
    return fdiv(r.y, r.z * t);
}
// cos(theta) of sample_gtr1 (tracer.rs:242-245) for the clearcoat's roughness
RPT_DEV float mat_cc_cos_theta(const Mat& m, float r1)
{
    float a = rmax(0.001f, m.clearcoat_roughness);
    float a2 = a * a;
    return fsqrt(fdiv(1.0f - rpt_powf(a2, 1.0f - r1), 1.0f - a2));
}
RPT_DEV float mat_cc_cos_theta(const MatRow& m, float r1)
{
    const float a2 = m.more[1].w;
    const float4 r = m.more[-1];
    const double lx = rpt_u2d((uint64_t)rpt_f2u(r.z) | ((uint64_t)rpt_f2u(r.w) << 32));
    return fsqrt(fdiv(1.0f - rpt_powf_log2x(a2, lx, 1.0f - r1), 1.0f - a2));
}
RPT_DEV void mat_row_derive(const Mat& m0, MatRowValues& m)
{
    m.lum = luminance(m0.rgb);
    m.w_diffuse = m.lum * (1.0f - m0.metallic) * (1.0f - m0.spec_trans);
    m.w_clearcoat = 0.25f * m0.clearcoat * (1.0f - m0.metallic);
    m.one_m_metallic = 1.0f - m0.metallic;
    m.dm = (1.0f - m0.metallic) * (1.0f - m0.spec_trans);
    {
        float a = m0.clearcoat_roughness;
        float a2 = a * a;
        m.gtr1_a2m1 = a2 - 1.0f;
        m.gtr1_k = kPi * rpt_log2f(a2);
    }
    {
        float a = rmax(0.001f, m0.clearcoat_roughness);
        float a2 = a * a;
        m.cc_a2 = a2;
        m.cc_log2_a2 = (rpt_f2u(a2) - 1u < 0x7f7fffffu) ? rpt_log2_core(a2) : 0.0;     // (rpt_powf_log2x reads it for such an a2 only)
    }
}

RPT_DEV void get_spec_color(const Mat& m, float eta, v3& spec_col, v3& sheen_col)   // tracer.rs:335
{
    float lum = luminance(m.rgb);
    v3 ctint = mk3(1.0f, 1.0f, 1.0f);
    if (lum > 0.0f) ctint = divs3(m.rgb, lum);
    float f0 = fdiv(1.0f - eta, 1.0f + eta);
    spec_col = mix3((f0 * f0) * mix3(mk3(1.0f, 1.0f, 1.0f), ctint, m.specular_tint), m.rgb, m.metallic);
    sheen_col = mix3(mk3(1.0f, 1.0f, 1.0f), ctint, m.sheen_tint);
}

template <class MT>
RPT_DEV float disney_fresnel(const MT& m, float eta, float ldoth, float vdoth)   // tracer.rs:435
{
    float metallic_fresnel = schlick_fresnel(ldoth);
    float dielectric = dielectric_fresnel(__builtin_fabsf(vdoth), eta);
    return mixf(dielectric, metallic_fresnel, mat_metallic(m));
}

template <class MT>
RPT_DEV v3 eval_diffuse(const MT& m, v3 c_sheen, v3 v, v3 l, v3 h, float& pdf)   // tracer.rs:343
{
    pdf = 0.0f;
    if (l.z <= 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    float ldh = dot3(l, h);
    float fl = schlick_fresnel(l.z);
    float fv = schlick_fresnel(v.z);
    float fh = schlick_fresnel(ldh);
    float fd90 = 0.5f + 2.0f * ldh * ldh * mat_roughness(m);
    float fd = mixf(1.0f, fd90, fl) * mixf(1.0f, fd90, fv);
    float fss90 = ldh * ldh * mat_roughness(m);
    float fss = mixf(1.0f, fss90, fl) * mixf(1.0f, fss90, fv);
    float ss = 1.25f * (fss * (fdiv(1.0f, l.z + v.z) - 0.5f) + 0.5f);
    v3 fsheen = (fh * mat_sheen(m)) * c_sheen;
    pdf = l.z * kInvPi;
    return mat_dm(m) * ((kInvPi * mixf(fd, ss, mat_subsurface(m))) * mat_rgb(m) + fsheen);
}

template <class MT>
RPT_DEV v3 eval_spec_reflection(const MT& m, float eta, v3 spec_col, v3 v, v3 l, v3 h, float& pdf, float fm)   // tracer.rs:368
{
    // fm = disney_fresnel(m, eta, dot3(l, h), dot3(v, h)) (tracer.rs:372): disney_eval, the only caller, has just computed it (tracer.rs:572)
    pdf = 0.0f;
    if (l.z <= 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    v3 f = mix3(spec_col, mk3(1.0f, 1.0f, 1.0f), fm);
    float d = gtr2aniso(h.z, h.x, h.y, mat_ax(m), mat_ay(m));
    float g1 = smithganiso(__builtin_fabsf(v.z), v.x, v.y, mat_ax(m), mat_ay(m));
    float g2 = g1 * smithganiso(__builtin_fabsf(l.z), l.x, l.y, mat_ax(m), mat_ay(m));
    pdf = fdiv(g1 * d, 4.0f * v.z);
    return divs3((d * g2) * f, 4.0f * l.z * v.z);
}

template <class MT>
RPT_DEV v3 eval_spec_refraction(const MT& m, float eta, v3 v, v3 l, v3 h, float& pdf)   // tracer.rs:384
{
    pdf = 0.0f;
    if (l.z >= 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    float vdh = dot3(v, h), ldh = dot3(l, h);
    float f = dielectric_fresnel(__builtin_fabsf(vdh), eta);
    float d = gtr2aniso(h.z, h.x, h.y, mat_ax(m), mat_ay(m));
    float g1 = smithganiso(__builtin_fabsf(v.z), v.x, v.y, mat_ax(m), mat_ay(m));
    float g2 = g1 * smithganiso(__builtin_fabsf(l.z), l.x, l.y, mat_ax(m), mat_ay(m));
    float denom = ldh + vdh * eta;
    denom *= denom;
    float eta2 = eta * eta;
    float jacobian = fdiv(__builtin_fabsf(ldh), denom);
    pdf = fdiv(g1 * rmax(0.0f, vdh) * d * jacobian, v.z);
    float s = fdiv((1.0f - mat_metallic(m)) * mat_spec_trans(m) * (1.0f - f) * d * g2 * __builtin_fabsf(vdh) * jacobian * eta2,
                   __builtin_fabsf(l.z * v.z));
    return s * mk3(rpt_powf(mat_rgb(m).x, 0.5f), rpt_powf(mat_rgb(m).y, 0.5f), rpt_powf(mat_rgb(m).z, 0.5f));
}

template <class MT>
RPT_DEV v3 eval_clearcoat(const MT& m, v3 v, v3 l, v3 h, float& pdf)   // tracer.rs:404
{
    pdf = 0.0f;
    if (l.z <= 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    float vdh = dot3(v, h);
    float fh = dielectric_fresnel(vdh, 1.0f / 1.5f);
    float f = mixf(0.04f, 1.0f, fh);
    float d = mat_gtr1(m, h.z);
    float g = smithg(l.z, 0.25f) * smithg(v.z, 0.25f);
    float jacobian = fdiv(1.0f, 4.0f * vdh);
    pdf = d * h.z * jacobian;
    return fdiv(mat_clearcoat(m) * f * d * g, 4.0f * l.z * v.z) * mk3(0.25f, 0.25f, 0.25f);
}

struct LobeWeights {
    float diffuse, spec_reflect, spec_refract, clearcoat;
};

template <class MT>
RPT_DEV LobeWeights get_lobe_probabilities(const MT& m, v3 spec_col, float approx_fresnel)   // tracer.rs:421
{
    LobeWeights w;
    float lum = mat_lum(m);
    w.diffuse = mat_w_diffuse(m, lum);
    w.spec_reflect = luminance(mix3(spec_col, mk3(1.0f, 1.0f, 1.0f), approx_fresnel));
    w.spec_refract = (1.0f - approx_fresnel) * mat_one_m_metallic(m) * mat_spec_trans(m) * lum;
    w.clearcoat = mat_w_clearcoat(m);
    float total = w.diffuse + w.spec_reflect + w.spec_refract + w.clearcoat;
    const v3 w3 = divs3(mk3(w.diffuse, w.spec_reflect, w.spec_refract), total);      // (four quotients by one total)
    w.diffuse = w3.x;
    w.spec_reflect = w3.y;
    w.spec_refract = w3.z;
    w.clearcoat = fdiv(w.clearcoat, total);
    return w;
}

// tracer.rs:184-189 / 449-454 / 559-564
RPT_DEV void onb(v3 n, v3& t, v3& b)
{
    v3 up = (__builtin_fabsf(n.z) < 0.999f) ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
    t = norm3(cross3(up, n));
    b = cross3(n, t);
}
RPT_DEV v3 to_local(v3 x, v3 y, v3 z, v3 v) { return mk3(dot3(v, x), dot3(v, y), dot3(v, z)); }   // tracer.rs:456
RPT_DEV v3 to_world(v3 x, v3 y, v3 z, v3 v) { return v.x * x + v.y * y + v.z * z; }               // tracer.rs:460
RPT_DEV v3 reflect3(v3 i, v3 n) { return i - (mk3(2.0f, 2.0f, 2.0f) * n) * splat3(dot3(n, i)); }  // tracer.rs:464
RPT_DEV v3 refract3(v3 i, v3 n, float eta)                                                        // tracer.rs:468
{
    float ndi = dot3(n, i);
    float k = 1.0f - eta * eta * (1.0f - ndi * ndi);
    if (k < 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    return eta * i - (eta * ndi + fsqrt(k)) * n;
}

// What disney_sample (tracer.rs:449-486) and disney_eval (tracer.rs:559-591) both
// compute first from the same inputs: the tangent frame of the shading normal, the view
// vector in it, and the specular / sheen colours.  Built once per bounce and shared.
struct ShadeFrame {
    v3 t, b;                   // onb(n)
    v3 v;                      // to_local(v_world)
    v3 spec_col, sheen_col;    // get_spec_color
};

RPT_DEV ShadeFrame make_frame(const Mat& m, float eta, v3 v_world, v3 n)
{
    ShadeFrame fr;
    onb(n, fr.t, fr.b);
    fr.v = to_local(fr.t, fr.b, n, v_world);
    get_spec_color(m, eta, fr.spec_col, fr.sheen_col);
    return fr;
}

// The two colours of get_spec_color: in the frame for a Mat; for a MatRow they are part of the row and stay there until read.
RPT_DEV v3 mat_spec_col(const Mat&, const ShadeFrame& fr) { return fr.spec_col; }
RPT_DEV v3 mat_spec_col(const MatRow& m, const ShadeFrame&) { const float4 c = m.more[kMatRowSpecCol]; return mk3(c.x, c.y, c.z); }
RPT_DEV v3 mat_sheen_col(const Mat&, const ShadeFrame& fr) { return fr.sheen_col; }
RPT_DEV v3 mat_sheen_col(const MatRow& m, const ShadeFrame&) { const float4 c = m.more[kMatRowSheenCol]; return mk3(c.x, c.y, c.z); }

// tracer.rs:441-553.  l_io: in = the previous bounce's world-space direction (zeros
// on the first bounce) which the specular branch reads before overwriting it
// (tracer.rs:531); out = the sampled world-space direction.
//
// The reference is an if / else-if / else over the three lobes.  On a 64-wide wave the three arms run one
// after the other, each with the lanes that chose it (measured on BASELINE configs[1]: 33 % / 13 % / 48 % of the
// wave), so every operation the arms have in common is hoisted out of them and runs once with all lanes:
//   * the renormalisation of r1 — (r1 - lo) / (hi - lo) with (lo, hi) = (0, cdf0), (cdf0, cdf1), (cdf1, 1); for the
//     diffuse arm r1 - 0 and cdf0 - 0 are exact, so this is the reference's r1 / cdf0;
//   * sin/cos of 2*pi*r2 (clearcoat: of 2*pi*r1, tracer.rs:247) and sqrt(r1) (diffuse and specular);
//   * the normalize every arm ends its sampling with — h = normalize(l + v) in the diffuse arm, l =
//     normalize(reflect(-v, h)) in the other two;
//   * for the clearcoat and the specular-reflection arm, which are both F*D*G / (4 l.z v.z) microfacet terms: the
//     dielectric Fresnel term, the two Smith terms' common tail 2n / (n + sqrt(E)), the pdf's division and the
//     first division of the value.
// Every lane still executes exactly the reference's operations on its own values, in the reference's order.
template <class MT>
RPT_DEV v3 disney_sample(const MT& m, float eta, const ShadeFrame& fr, v3 n, v3& l_io, float& pdf, Rng& rng)
{
    pdf = 0.0f;
    v3 f = mk3(0.0f, 0.0f, 0.0f);
    float r1 = rng.gen();
    float r2 = rng.gen();

    const v3 t = fr.t, b = fr.b, v = fr.v;
    float approx_fresnel = disney_fresnel(m, eta, v.z, v.z);
    LobeWeights w = get_lobe_probabilities(m, mat_spec_col(m, fr), approx_fresnel);

    float cdf0 = w.diffuse;
    float cdf1 = cdf0 + w.clearcoat;

    const bool is_d = r1 < cdf0;                                    // tracer.rs:501
    const bool is_c = !is_d && (r1 < cdf1);                         // tracer.rs:510
    const bool is_s = !is_d && !is_c;                               // tracer.rs:520

    const float lo = is_d ? 0.0f : (is_c ? cdf0 : cdf1);
    const float hi = is_d ? cdf0 : (is_c ? cdf1 : 1.0f);
    r1 = fdiv(r1 - lo, hi - lo);                                    // tracer.rs:502, 511, 521
    const float phi = kTwoPi * (is_c ? r1 : r2);                    // tracer.rs:329, 247, 267
    float sn, cs;
    rpt_sincosf(phi, &sn, &cs);
    const float rs = fsqrt(r1);                           // tracer.rs:327, 266

    v3 pre;                    // what the arm normalises: l + v (diffuse), reflect / refract(-v, h) (the others)
    v3 other;                  // the arm's other vector: l (diffuse), h (the others)
    bool reflected = true;
    float ff = 1.0f;
    // the dielectric Fresnel term of the arm's half vector: the specular arm needs it to choose between reflection and refraction
    // (tracer.rs:527), eval_spec_reflection needs the same value again (tracer.rs:372), eval_clearcoat its own (tracer.rs:409): computed
    // once, for both arms at once, together with what else they have in common: h's flip into the upper hemisphere and the reflection
    float dfr = 0.0f;
    v3 hs = mk3(0.0f, 0.0f, 1.0f);                                  // the sampled half vector of the clearcoat / specular arm
    if (is_d) {
        RPT_PROF(PB_LOBE_DIFFUSE);
        v3 l;                                                       // cosine_sample_hemisphere, tracer.rs:324
        l.x = rs * cs;
        l.y = rs * sn;
        l.z = fsqrt(rmax(0.0f, 1.0f - l.x * l.x - l.y * l.y));
        pre = l + v;
        other = l;
    } else if (is_c) {
        RPT_PROF(PB_LOBE_CLEARCOAT);
        float cos_theta = mat_cc_cos_theta(m, r1);                  // sample_gtr1, tracer.rs:242 (r2 is unused there)
        float sin_theta = clamp01(fsqrt(1.0f - (cos_theta * cos_theta)));
        hs = mk3(sin_theta * cs, sin_theta * sn, cos_theta);
    } else {
        RPT_PROF(PB_LOBE_SPEC);
        v3 vh = norm3(mk3(mat_ax(m) * v.x, mat_ay(m) * v.y, v.z));            // sample_ggxvndf, tracer.rs:256
        float lensq = vh.x * vh.x + vh.y * vh.y;
        v3 t_1 = mk3(1.0f, 0.0f, 0.0f);
        if (lensq > 0.0f) t_1 = scale3(mk3(-vh.y, vh.x, 0.0f), fdiv(1.0f, fsqrt(lensq)));
        v3 t_2 = cross3(vh, t_1);
        float t1 = rs * cs;
        float t2 = rs * sn;
        float s = 0.5f * (1.0f + vh.z);
        t2 = (1.0f - s) * fsqrt(1.0f - t1 * t1) + s * t2;
        v3 nh = t1 * t_1 + t2 * t_2 + fsqrt(rmax(0.0f, 1.0f - t1 * t1 - t2 * t2)) * vh;
        hs = norm3(mk3(mat_ax(m) * nh.x, mat_ay(m) * nh.y, rmax(0.0f, nh.z)));
    }
    if (!is_d) {
        v3 h = hs;
        if (h.z < 0.0f) h = -h;                                     // tracer.rs:513, 524
        const float vdh = dot3(v, h);
        dfr = dielectric_fresnel(is_c ? vdh : __builtin_fabsf(vdh), is_c ? (1.0f / 1.5f) : eta);
        if (is_s) {
            float fresnel = mixf(dfr, schlick_fresnel(dot3(l_io, h)), mat_metallic(m));        // disney_fresnel, tracer.rs:435
            ff = 1.0f - ((1.0f - fresnel) * mat_spec_trans(m) * (1.0f - mat_metallic(m)));
            float rnd = rng.gen();
            reflected = rnd < ff;
        }
        if (reflected) pre = reflect3(-v, h);
        else pre = refract3(-v, h, eta);
        other = h;
    }
    const v3 nrm = norm3(pre);                                      // tracer.rs:504, 515, 537 / 542
    const v3 l = mk3(is_d ? other.x : nrm.x, is_d ? other.y : nrm.y, is_d ? other.z : nrm.z);
    const v3 h = mk3(is_d ? nrm.x : other.x, is_d ? nrm.y : other.y, is_d ? nrm.z : other.z);

    if (is_d) {
        RPT_PROF(PB_LOBE_DIFFUSE);
        f = eval_diffuse(m, mat_sheen_col(m, fr), v, l, h, pdf);
        pdf *= w.diffuse;
    } else if (!reflected) {
        f = eval_spec_refraction(m, eta, v, l, h, pdf);
        pdf *= 1.0f - ff;
        pdf *= w.spec_reflect + w.spec_refract;
    } else {
        // eval_clearcoat (tracer.rs:404-419) and eval_spec_reflection (tracer.rs:368-382) side by side
        RPT_PROF(PB_LOBE_SPEC);
        if (!(l.z <= 0.0f)) {                                       // tracer.rs:370, 406
            const float vdh = dot3(v, h);                           // (dfr: dielectric_fresnel(is_c ? vdh : |vdh|, is_c ? 1 / 1.5 : eta), from the arm)
            float d;                                                // the normal-distribution term
            float e_v, e_l;                                         // what each Smith term takes the root of
            v3 fcol;
            if (is_c) {
                d = mat_gtr1(m, h.z);
                const float a = 0.25f * 0.25f;                      // smithg, tracer.rs:276
                const float bv = v.z * v.z, bl = l.z * l.z;
                e_v = a + bv - a * bv;
                e_l = a + bl - a * bl;
                const float fc = mixf(0.04f, 1.0f, dfr);
                fcol = mk3(fc, fc, fc);
            } else {
                d = gtr2aniso(h.z, h.x, h.y, mat_ax(m), mat_ay(m));
                const float av = v.x * mat_ax(m), bv = v.y * mat_ay(m), cv = __builtin_fabsf(v.z);   // smithganiso, tracer.rs:301
                const float al = l.x * mat_ax(m), bl = l.y * mat_ay(m), cl = __builtin_fabsf(l.z);
                e_v = av * av + bv * bv + cv * cv;
                e_l = al * al + bl * bl + cl * cl;
                const float fm = mixf(dfr, schlick_fresnel(dot3(l, h)), mat_metallic(m));       // disney_fresnel, tracer.rs:435
                fcol = mix3(mat_spec_col(m, fr), mk3(1.0f, 1.0f, 1.0f), fm);
            }
            const float n_v = is_c ? v.z : __builtin_fabsf(v.z);
            const float n_l = is_c ? l.z : __builtin_fabsf(l.z);
            const float g_v = fdiv(2.0f * n_v, n_v + fsqrt(e_v));
            const float g_l = fdiv(2.0f * n_l, n_l + fsqrt(e_l));
            const float g = g_v * g_l;                              // clearcoat: smithg(l) * smithg(v); specular: g1 * smithganiso(l)
            // pdf: clearcoat d * h.z * (1 / (4 vdh)); specular (g1 * d) / (4 v.z)
            const float q = fdiv(is_c ? 1.0f : g_v * d, 4.0f * (is_c ? vdh : v.z));
            pdf = is_c ? (d * h.z * q) : q;
            const float den = 4.0f * l.z * v.z;
            // value: clearcoat (clearcoat * F * d * g / den) * 0.25; specular ((d * g) * F) / den per channel
            const float dg = d * g;
            const float numx = is_c ? (mat_clearcoat(m) * fcol.x * d * g) : (dg * fcol.x);
            const v3 q3 = divs3(mk3(numx, dg * fcol.y, dg * fcol.z), den);          // (the clearcoat arm uses the first quotient only)
            if (is_c) {
                const float c = q3.x * 0.25f;
                f = mk3(c, c, c);
            } else {
                f = q3;
            }
        }
        if (is_c) {
            pdf *= w.clearcoat;
        } else {
            pdf *= ff;
            pdf *= w.spec_reflect + w.spec_refract;
        }
    }
    l_io = to_world(t, b, n, l);
    return __builtin_fabsf(dot3(n, l_io)) * f;
}

// tracer.rs:555-626
template <class MT>
RPT_DEV v3 disney_eval(const MT& m, float eta, const ShadeFrame& fr, v3 n, v3 l_world, float& bsdf_pdf)
{
    bsdf_pdf = 0.0f;
    v3 f = mk3(0.0f, 0.0f, 0.0f);
    const v3 t = fr.t, b = fr.b, v = fr.v;
    v3 l = to_local(t, b, n, l_world);

    // tracer.rs:566-570: h = normalize(l + v) above the surface, normalize(l + eta * v) below.  One normalize for both (1 * v is v):
    // as an if / else a wave with lanes on either side runs two.
    v3 h = norm3(l + ((l.z > 0.0f) ? 1.0f : eta) * v);
    if (h.z < 0.0f) h = -h;

    float fresnel = disney_fresnel(m, eta, dot3(l, h), dot3(v, h));
    LobeWeights w = get_lobe_probabilities(m, mat_spec_col(m, fr), fresnel);

    float pdf;
    if (w.diffuse > 0.0f && l.z > 0.0f) {
        f = f + eval_diffuse(m, mat_sheen_col(m, fr), v, l, h, pdf);
        bsdf_pdf += pdf * w.diffuse;
    }
    if (w.spec_reflect > 0.0f && l.z > 0.0f && v.z > 0.0f) {
        f = f + eval_spec_reflection(m, eta, mat_spec_col(m, fr), v, l, h, pdf, fresnel);
        bsdf_pdf += pdf * w.spec_reflect;
    }
    if (w.spec_refract > 0.0f && l.z < 0.0f) {
        f = f + eval_spec_refraction(m, eta, v, l, h, pdf);
        bsdf_pdf += pdf * w.spec_refract;
    }
    if (w.clearcoat > 0.0f && l.z > 0.0f && v.z > 0.0f) {
        f = f + eval_clearcoat(m, v, l, h, pdf);
        bsdf_pdf += pdf * w.clearcoat;
    }
    return __builtin_fabsf(l.z) * f;
}

}  // namespace RPT_NS
#endif  // this pass
