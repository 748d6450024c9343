// dev_math.h — f32 vector arithmetic and RNG of the HIP integrator (gfx950).
//
// Every expression keeps the reference's operation order, because results are
// compared bit for bit with a CPU restatement: the whole library is compiled with
// -ffp-contract=off, f32 divide and sqrt are the correctly rounded expansions
// (hipcc default, -fhip-fp32-correctly-rounded-divide-sqrt), and transcendental
// functions come from include/rpt_strict_math.h.
#ifndef RPT_NS                        // (the namespace of this pass: dev_math.h, "two passes")
#define RPT_NS rptdev
#endif
#if (defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_MATH_H_PLAIN)) || (!defined(RPT_PLAIN_PASS) && !defined(RPT_DEV_MATH_H_NORMAL))
#ifdef RPT_PLAIN_PASS
#define RPT_DEV_MATH_H_PLAIN
#else
#define RPT_DEV_MATH_H_NORMAL
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rpt_strict_math.h"
#include "dev_prof.h"
#include "dev_scene.h"

#define RPT_DEV __device__ __forceinline__

namespace RPT_NS {
using namespace rptscene;

// lib.rs:8-10
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 1.0f / 3.14159265358979323846f;
constexpr float kTwoPi = 3.14159265358979323846f * 2.0f;

struct v3 {
    float x, y, z;
};

RPT_DEV v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
RPT_DEV v3 splat3(float s) { return v3{s, s, s}; }                       // F3::new_x, fx.rs:233
RPT_DEV v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }   // fx.rs:437
RPT_DEV v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }   // fx.rs:453
RPT_DEV v3 operator*(v3 a, v3 b) { return v3{a.x * b.x, a.y * b.y, a.z * b.z}; }   // fx.rs:469
RPT_DEV float fdiv(float n, float d);
RPT_DEV v3 operator/(v3 a, v3 b) { return v3{fdiv(a.x, b.x), fdiv(a.y, b.y), fdiv(a.z, b.z)}; }   // fx.rs:485
RPT_DEV v3 operator*(float s, v3 a) { return v3{s * a.x, s * a.y, s * a.z}; }      // fx.rs:477 (f32 * F3)
RPT_DEV v3 operator-(v3 a) { return v3{-a.x, -a.y, -a.z}; }                        // fx.rs:509
RPT_DEV v3 scale3(v3 a, float f) { return v3{a.x * f, a.y * f, a.z * f}; }         // F3::mult_f, fx.rs:346
// ---- division and square root -----------------------------------------------------------------------------------
// The reference divides (fx.rs:307-313: normalize is THREE divides by the length; F3 / F3 is component-wise), and results are
// compared bit for bit, so every quotient and every root here must be the correctly rounded one.  hipcc's expansion of `a / b` is a
// chain of ten dependent instructions (v_div_scale x2, v_rcp, five fma, v_div_fmas, v_div_fixup), its sqrtf ~14 (scaling, v_sqrt,
// both neighbours tested with an fma each, fix-ups), and the megakernels are bound by exactly these chains (profiles/NOTES.md).
//
// The SHORT sequences
//     r0 = v_rcp_f32(d);  r = fma(fma(-d, r0, 1), r0, r0);  q0 = n * r;  q = fma(fma(-d, q0, n), r, q0);  v_div_fixup(q, d, n)
//     y = v_rsq_f32(x);  s0 = x * y;  s = fma(fma(-s0, s0, x), 0.5 * y, s0)
// — one Newton step on the reciprocal, shared by every numerator of one denominator, ONE Markstein correction of the quotient; one
// correction of the root by its residual — return the correctly rounded result for EVERY significand: tools/proofs/div_exhaustive.hip
// compares the quotient with hipcc's on all 2^23 x 2^23 pairs (7.0e13 quotients, 0 mismatches), tools/proofs/sqrt_exhaustive.hip the
// root with hipcc's sqrtf on all 121 * 2^23 floats of [2^-60, 2^61) (0 mismatches, also for round 3's form with v_sqrt_f32 AND v_rsq_f32 —
// transcendental instructions issue at a quarter of the rate, so the one saved is worth three others —; the variant with v_rcp_f32(s0)
// has 60), both on gfx950 (profiles/r3/proofs/, profiles/r4/proofs/).  Every step commutes exactly with scaling by powers of two and with the operands' signs as long as no
// intermediate leaves the normal range: |d|, |n| in [2^-61, 2^60) keeps r, q0, the residual (>= 2^-46 |n|) and q normal.  Zero and
// NaN numerators, and NaN denominators, are answered by v_div_fixup without looking at q, exactly as at the end of hipcc's sequence
// (it supplies IEEE's sign of a zero quotient too).  Outside the range the sequences are WRONG, so somebody has to look:
//
//   RPT_MATH_MODE 2 (the shipped strict kernels): the range tests are TRACKERS.  Every operation folds its operands into two words per
//     lane in LDS with no-return DS min / max — no VALU result to wait for, no vote, no branch: the basic block goes on — and the kernel
//     looks at them ONCE PER SAMPLE (guard_sample_ok); a sample that saw an operand outside the range is recomputed from its camera
//     ray with the plain operations (namespace rptplain, "two passes" below; kernel_common.h sample_guard).  Per operation that is 2 VALU + 2 DS
//     where the vote below costs 6 VALU, 3 scalar instructions and a branch that ends the scheduler's block: configs[1] 11.7 -> 12.6
//     Gsamples/s (round 4; without any test at all: 13.3).
//       lo_e  min of v_frexp_exp_i32_f32 over every numerator and every root's argument: floor(log2 |x|) + 1 for finite non-zero x
//             (denormals included), 0 for zero, infinity and NaN, which therefore pass;           the sample is good if lo_e >= -59
//       hi_u  max, as unsigned integers, of the bit patterns of max(|n|, |d|, |1/d|) of every divide (v_max3_f32 ignores a NaN
//             operand; d = 0 and |d| <= 2^-60 show as a huge 1/d) and of every root's argument (a negative one, an infinity or a
//             NaN reads as a huge integer);                                                          good if hi_u < bits(2^60)
//     so inside a good sample every n is 0, NaN or in [2^-60, 2^60), every d NaN or in (2^-60, 2^60), every root's argument +0 or in
//     [2^-60, 2^60): inside what the proofs cover.  (+0 under the root: 0 x rsq(0) would be 0 x inf; with rsq clamped it is 0, and so
//     is the correction.)
//   RPT_MATH_MODE 1 (-DRPT_GUARD_PER_OP: k_compact.hip, k_sdf.hip, k_large.hip — kernels whose walks, marches and barriers wait at every step): the test next
//     to the operation, operands outside the range take hipcc's sequence in the lanes concerned behind a wave vote (rounds 3-4).
//   RPT_MATH_MODE 0 (namespace rptplain; the relaxed build, where `/` and sqrtf are hipcc's fast ones): the plain operations.
//
// TWO PASSES.  dev_math.h, dev_bsdf.h, dev_media.h, dev_integrator.h, dev_scene_large.h (and dev_probes.h) hold FUNCTIONS (and the types
// only they use) and can be included twice by one translation unit: normally — namespace rptdev, the mode above that the build asks
// for — and once more under `#define RPT_PLAIN_PASS` + `#define RPT_NS rptplain` — the same functions in RPT_MATH_MODE 0, what a
// kernel recomputes a flagged sample with (kernel_common.h does the second inclusion).  Their include guards are per pass.  The types both
// passes and the host share are in dev_scene.h, namespace rptscene, which holds no function over them: argument-dependent lookup
// cannot mix the passes.
#undef RPT_MATH_MODE
#if defined(RPT_PLAIN_PASS) || defined(RPT_RELAXED_BUILD) || defined(RPT_PLAIN_MATH)
#define RPT_MATH_MODE 0
#elif defined(RPT_GUARD_PER_OP)
#define RPT_MATH_MODE 1
#else
#define RPT_MATH_MODE 2
#endif

RPT_DEV int imin3(int a, int b, int c) { int m = a < b ? a : b; return m < c ? m : c; }
RPT_DEV int imax3(int a, int b, int c) { int m = a > b ? a : b; return m > c ? m : c; }
RPT_DEV int div_exp(float x) { return __builtin_amdgcn_frexp_expf(x); }
RPT_DEV float div_refine(float d, float r0) { return __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0); }   // 1 / d to within what the correction needs
RPT_DEV float div_with_rcp(float n, float d, float r)
{
    const float q0 = n * r;
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-d, q0, n), r, q0), d, n);
}

#if RPT_MATH_MODE == 0
RPT_DEV float fdiv(float n, float d) { return n / d; }
RPT_DEV v3 divs3(v3 a, float d) { return v3{a.x / d, a.y / d, a.z / d}; }
RPT_DEV v3 divs3_norm(v3 a, float len) { return divs3(a, len); }
RPT_DEV float fsqrt(float x) { return __builtin_sqrtf(x); }
#elif RPT_MATH_MODE == 1
RPT_DEV float fmax3abs(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c)); }
// The range tests next to the operation, in the operands the short sequence has anyway: everything below 2^60 is one v_max3_f32 over
// |n|, |d| and |1 / d| (a tiny d is a huge reciprocal; a NaN is ignored unless all three are: then the compare fails and hipcc's divide
// answers), the numerators' lower end is v_frexp_exp >= -59 (zero, infinity and NaN read 0 and pass).  4 VALU for a lone quotient where
// the exponents of both operands took 6, 2 for a root where the exponent field took 3 (round 4: configs[3] +2.6 %, configs[4] +1.3 %,
// one-sample launches +2 %).
RPT_DEV float fdiv(float n, float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    float q = div_with_rcp(n, d, div_refine(d, r0));
    const bool ok = (fmax3abs(n, d, r0) < 0x1p60f) && (div_exp(n) >= -59);
    if (__builtin_expect(__ballot(!ok) != 0ull, 0)) { if (!ok) q = n / d; }
    return q;
}
RPT_DEV v3 divs3(v3 a, float d)                                     // a / F3::new_x(d): three quotients, one reciprocal
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float r = div_refine(d, r0);
    v3 q = v3{div_with_rcp(a.x, d, r), div_with_rcp(a.y, d, r), div_with_rcp(a.z, d, r)};
    const bool ok = (fmax3abs(fmax3abs(a.x, a.y, a.z), d, r0) < 0x1p60f) && (imin3(div_exp(a.x), div_exp(a.y), div_exp(a.z)) >= -59);
    if (__builtin_expect(__ballot(!ok) != 0ull, 0)) { if (!ok) q = v3{a.x / d, a.y / d, a.z / d}; }
    return q;
}
// normalize (`len` = len3(a)): |a.i| >= 2^60 makes the sum of squares >= 2^120 and its root >= 2^60: the numerators need no upper test
RPT_DEV v3 divs3_norm(v3 a, float len)
{
    const float r0 = __builtin_amdgcn_rcpf(len);
    const float r = div_refine(len, r0);
    v3 q = v3{div_with_rcp(a.x, len, r), div_with_rcp(a.y, len, r), div_with_rcp(a.z, len, r)};
    const bool ok = (__builtin_fmaxf(__builtin_fabsf(len), __builtin_fabsf(r0)) < 0x1p60f) && (imin3(div_exp(a.x), div_exp(a.y), div_exp(a.z)) >= -59);
    if (__builtin_expect(__ballot(!ok) != 0ull, 0)) { if (!ok) q = v3{a.x / len, a.y / len, a.z / len}; }
    return q;
}
RPT_DEV float fsqrt(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y;
    float s = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * y, s0);
    const bool ok = (rpt_f2u(x) - 0x21800000u) < (0x5E000000u - 0x21800000u);      // positive, 2^-60 <= x < 2^61
    if (__builtin_expect(__ballot(!ok) != 0ull, 0)) { if (!ok) s = __builtin_sqrtf(x); }
    return s;
}
#else
struct GuardCell {
    int lo_e;
    uint32_t hi_u;
    uint32_t pad[2];                                                // 16 bytes per lane, like the kernels' other per-lane LDS slots: one address register serves them all (+0.3 %)
};
__shared__ GuardCell g_guard[256];                                  // (4 KB of the workgroup's LDS)
constexpr int kGuardLoE = -59;
constexpr uint32_t kGuardHiU = 0x5D800000u;                         // bits(2^60)
RPT_DEV void guard_note_lo(int e) { (void)__hip_atomic_fetch_min(&g_guard[threadIdx.x].lo_e, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RPT_DEV void guard_note_hi(float hi) { (void)__hip_atomic_fetch_max(&g_guard[threadIdx.x].hi_u, rpt_f2u(hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
RPT_DEV void guard_note(int e, float hi)
{
    guard_note_lo(e);
    guard_note_hi(hi);
}
RPT_DEV GuardCell guard_clean() { GuardCell c; c.lo_e = 0; c.hi_u = 0u; return c; }
RPT_DEV void guard_reset() { g_guard[threadIdx.x].lo_e = 0; g_guard[threadIdx.x].hi_u = 0u; }
RPT_DEV bool guard_cell_ok(GuardCell c) { return c.lo_e >= kGuardLoE && c.hi_u < kGuardHiU; }
RPT_DEV bool guard_sample_ok() { GuardCell c; c.lo_e = g_guard[threadIdx.x].lo_e; c.hi_u = g_guard[threadIdx.x].hi_u; return guard_cell_ok(c); }
// a path that changes lanes (the compacting kernel) carries its trackers along: fold the lane's into `c`, leave the lane's clean
RPT_DEV void guard_take(GuardCell& c)
{
    const GuardCell l = g_guard[threadIdx.x];
    c.lo_e = l.lo_e < c.lo_e ? l.lo_e : c.lo_e;
    c.hi_u = l.hi_u > c.hi_u ? l.hi_u : c.hi_u;
    guard_reset();
}
RPT_DEV float fmax3abs(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c)); }
RPT_DEV float fdiv(float n, float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    // a literal numerator inside the range (1 / x, 2 / x ...) needs no test of its own (+0.3 %)
    if (__builtin_constant_p(n) && ((__builtin_fabsf(n) >= 0x1p-60f && __builtin_fabsf(n) < 0x1p60f) || n == 0.0f)) {
        guard_note_hi(__builtin_fmaxf(__builtin_fabsf(d), __builtin_fabsf(r0)));
        return div_with_rcp(n, d, div_refine(d, r0));
    }
    guard_note(div_exp(n), fmax3abs(n, d, r0));
    return div_with_rcp(n, d, div_refine(d, r0));
}
RPT_DEV v3 divs3(v3 a, float d)                                     // a / F3::new_x(d): three quotients, one reciprocal
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    guard_note(imin3(div_exp(a.x), div_exp(a.y), div_exp(a.z)), fmax3abs(fmax3abs(a.x, a.y, a.z), d, r0));
    const float r = div_refine(d, r0);
    return v3{div_with_rcp(a.x, d, r), div_with_rcp(a.y, d, r), div_with_rcp(a.z, d, r)};
}
// normalize, and nothing else: `len` MUST be len3(a) — the root of an argument fsqrt has just tracked, so either the sample is flagged
// already or len is in [2^-30, 2^30] and no numerator exceeds it: only the numerators' lower end is left to test (+0.9 %)
RPT_DEV v3 divs3_norm(v3 a, float len)
{
    const float r0 = __builtin_amdgcn_rcpf(len);
    guard_note_lo(imin3(div_exp(a.x), div_exp(a.y), div_exp(a.z)));
    const float r = div_refine(len, r0);
    return v3{div_with_rcp(a.x, len, r), div_with_rcp(a.y, len, r), div_with_rcp(a.z, len, r)};
}
RPT_DEV float fsqrt(float x)
{
    const float y = __builtin_fminf(__builtin_amdgcn_rsqf(x), 1.0e30f);   // (x = +0: 0 x 1e30 = 0 where 0 x inf would be a NaN; no root of the range has rsq above 2^30)
    const float s0 = x * y;
    guard_note(div_exp(x), x);
    return __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * y, s0);
}
#endif

RPT_DEV float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       // fx.rs:335
RPT_DEV v3 cross3(v3 a, v3 b)                                                      // fx.rs:339
{
    return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RPT_DEV float len3(v3 a) { return fsqrt(a.x * a.x + a.y * a.y + a.z * a.z); }   // fx.rs:331
RPT_DEV v3 norm3(v3 a) { return divs3_norm(a, len3(a)); }                                    // fx.rs:307 (three divides by the length)
RPT_DEV v3 mix3(v3 a, v3 b, float v)                                               // math.rs:34
{
    return v3{(1.0f - v) * a.x + b.x * v, (1.0f - v) * a.y + b.y * v, (1.0f - v) * a.z + b.z * v};
}
RPT_DEV float mixf(float a, float b, float v) { return (1.0f - v) * a + b * v; }   // tracer.rs:229

// f32::max as the oracle restates it: a NaN operand yields the other one; equal operands (+0 and -0 too) yield `other`.
// (self > other) ? self : other already answers a NaN `self` (the compare fails: other); a NaN `other` needs the second select — unless
// `self` is a literal, where (other >= self) ? other : self does both in one compare: two instructions instead of six.
RPT_DEV float rmax(float self, float other)
{
    if (__builtin_constant_p(self) && self == self) return (other >= self) ? other : self;
    const float r = (self > other) ? self : other;
    return (other != other) ? self : r;
}
// f32::clamp(0, 1): NaN stays NaN.
RPT_DEV float clamp01(float x)
{
    float r = (x > 1.0f) ? 1.0f : x;
    return (x < 0.0f) ? 0.0f : r;
}
// f32::clamp(lo, hi): NaN stays NaN.
RPT_DEV float clampf(float x, float lo, float hi)
{
    float r = (x > hi) ? hi : x;
    return (x < lo) ? lo : r;
}
// a % 2.0 for Rust f32 (C fmodf): exact via trunc for every finite a; +-inf and NaN
// give NaN like fmodf.  (Sign of a zero result can differ from fmodf; callers only
// compare the result with 1.0.)
RPT_DEV float rem2(float a) { return a - 2.0f * __builtin_truncf(a * 0.5f); }

// ---- RNG --------------------------------------------------------------------
// Replacement for rand::thread_rng (tracer.rs:44), which cannot be seeded: every path owns a PCG stream — PCG-RXS-M-XS-32
// (O'Neill 2014): a 32-bit LCG whose increment selects the stream, with the RXS-M-XS output permutation — started from a
// 63-bit key (state, increment) hashed from (seed, frame, pixel); u32 -> f32 as rand 0.8.5 Standard: (u >> 8) * 2^-24.
// Round 2 drew pcg_hash(key + counter) from a 32-bit key: paths with nearby keys read overlapping windows of ONE sequence,
// and (seed, frame) was folded to 32 bits per frame.  Now two paths share draws only if state AND increment coincide.  The cost
// per draw is the same two multiplies (pcg_hash IS one LCG step + the permutation).
RPT_DEV uint32_t pcg_hash(uint32_t v)
{
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
struct Rng {
    uint32_t state;
    uint32_t inc;              // odd: the stream
    // a = pcg_hash(pixel index), b = pcg_hash(a): computed once per pixel by the state-machine kernels
    RPT_DEV void init_hashed(FrameKey fk, uint32_t a, uint32_t b)
    {
        state = pcg_hash(a ^ fk.k0);
        inc = pcg_hash(b ^ fk.k1) | 1u;
    }
    RPT_DEV void init(FrameKey fk, uint32_t pixel_index)
    {
        const uint32_t a = pcg_hash(pixel_index);
        init_hashed(fk, a, pcg_hash(a));
    }
    RPT_DEV uint32_t next_u32()
    {
        state = state * 747796405u + inc;
        uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
        return (word >> 22u) ^ word;
    }
    RPT_DEV float gen() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
};

// test probes: how many draws lie between two states of one stream (at most `limit`)
RPT_DEV uint32_t rng_draws_between(Rng from, const Rng& to, uint32_t limit = 64u)
{
    uint32_t n = 0u;
    while (from.state != to.state && n < limit) { (void)from.next_u32(); ++n; }
    return n;
}

}  // namespace RPT_NS
#endif  // this pass
