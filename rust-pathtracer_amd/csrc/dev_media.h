// dev_media.h — participating media, device side: the leaf functions.
//
// PROJECT-DEFINED (include/rpt.h, "participating media"): the reference declares Medium (material.rs:8-34, globals.rs:19)
// and never reads it, so there is no reference behaviour to match.  The specification is the arithmetic of
// oracle/rpt_oracle.hpp (Tracer::phase_hg, sample_hg, medium_transmittance, the media branches of sample_pixel and
// direct_light); these functions restate it operation for operation and are compared with it bit for bit.
// Only kernels instantiated for WithMedia<Scene> (dev_scene.h) contain any of this.
#ifndef RPT_NS                        // (the namespace of this pass: dev_math.h, "two passes")
#define RPT_NS rptdev
#endif
#if (defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_MEDIA_H_PLAIN)) || (!defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_MEDIA_H_NORMAL))
#ifdef RPT_PLAIN_PASS
#define RPT_DEV_MEDIA_H_PLAIN
#else
#define RPT_DEV_MEDIA_H_NORMAL
#endif

#include "dev_bsdf.h"
#include "dev_scene.h"

namespace RPT_NS {
using namespace rptscene;

struct DevMedium {
    uint32_t type;             // RPT_MEDIUM_*
    float density;
    v3 color;
    float anisotropy;          // already clamped to [-0.9, 0.9] (Material::finalize, material.rs:126)
};

// PathRegs.medium: 0 = not in a medium, otherwise 1 + index of the material whose Medium the path is in;
// kMediumScatterNow: the bounce being processed is a scatter event inside that medium (set by TRACE, consumed by SHADE).
constexpr uint32_t kMediumScatterNow = 0x8000u;
constexpr uint32_t kMediumIndexMask = 0x7FFFu;
constexpr uint32_t kNoMediumIdx = 0xFFFFu;                          // "the layered material carries no Medium"
constexpr uint32_t kMaxMediaMaterials = 0x7FFEu;                    // material indices a path can remember (rpt_upload_scene checks)
constexpr float kInv4Pi = 0.0795774715459476679f;

// f32::min: a NaN operand yields the other one.
RPT_DEV float rmin(float self, float other)
{
    float r = (self < other) ? self : other;
    r = (other != other) ? self : r;
    r = (self != self) ? other : r;
    return r;
}

RPT_DEV float phase_hg(float cos_theta, float g)
{
    const float denom = 1.0f + g * g + 2.0f * g * cos_theta;
    return fdiv(kInv4Pi * (1.0f - g * g), denom * fsqrt(denom));
}

RPT_DEV v3 sample_hg(v3 v, float g, float r1, float r2)
{
    float cos_theta;
    if (__builtin_fabsf(g) < 0.001f) cos_theta = 1.0f - 2.0f * r2;
    else {
        const float sqr_term = fdiv(1.0f - g * g, 1.0f + g - 2.0f * g * r2);
        cos_theta = fdiv(-(1.0f + g * g - sqr_term * sqr_term), 2.0f * g);
    }
    const float phi = r1 * kTwoPi;
    const float sin_theta = clamp01(fsqrt(1.0f - (cos_theta * cos_theta)));
    float sin_phi, cos_phi;
    rpt_sincosf(phi, &sin_phi, &cos_phi);
    v3 t, b;
    onb(v, t, b);
    return (sin_theta * cos_phi) * t + (sin_theta * sin_phi) * b + cos_theta * v;
}

// what is left of a light's radiance after `dist` inside the medium
RPT_DEV v3 medium_transmittance(const DevMedium& md, float dist)
{
    if (md.type == RPT_MEDIUM_ABSORB)
        return mk3(rpt_expf(-(((1.0f - md.color.x) * dist) * md.density)), rpt_expf(-(((1.0f - md.color.y) * dist) * md.density)),
                   rpt_expf(-(((1.0f - md.color.z) * dist) * md.density)));
    if (md.type == RPT_MEDIUM_SCATTER) {
        const float e = rpt_expf(-(dist * md.density));
        return mk3(e, e, e);
    }
    return mk3(1.0f, 1.0f, 1.0f);
}

RPT_DEV DevMedium medium_of(const DevMaterial& m)
{
    return DevMedium{m.medium_type, m.medium_density, mk3(m.medium_color[0], m.medium_color[1], m.medium_color[2]),
                     clampf(m.medium_anisotropy, -0.9f, 0.9f)};
}

}  // namespace RPT_NS
#endif  // this pass
