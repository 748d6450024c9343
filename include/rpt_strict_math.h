/* rpt_strict_math.h — bit-reproducible elementary functions (host + gfx950 device).
 *
 * WHY THIS EXISTS
 * The reference computes its transcendental functions with Rust's
 * f32::{sin,cos,tan,powf,log2}, which defer to the platform libm
 * (call sites: rust-pathtracer/src/tracer.rs:181,239,248,250-251,266-267,
 * 329-330,401; rust-pathtracer/src/scene.rs:33; camera/pinhole.rs:43;
 * buffer.rs:44,59).  A platform libm is not bit-specified, so "the reference's
 * result" is only defined up to ~1 ulp per call.  To make a CPU oracle and a
 * GPU kernel comparable BIT FOR BIT, both use the functions below instead of a
 * libm: they are built only from IEEE-754 operations that are correctly rounded
 * on x86-64 and on gfx950 (add, mul, explicit fma, f32<->f64 conversion, integer
 * ops), so gcc on the host and hipcc on the device produce identical bits.
 *
 * Accuracy (tests/test_strict_math.py checks these against mpmath/glibc):
 *   rpt_sincosf : <= 1.5 ulp (abs error < 7.1e-8) on [0, 2*pi], the only range the path
 *                 uses; abs error < 3e-7 for |x| <= 3e4
 *   rpt_tanf    : <= 3 ulp on (0, 1.5) (host-only use: camera fov)
 *   rpt_log2f   : f64 core, rounded once to f32 -> correctly rounded in all but
 *                 ~1e-7 of inputs
 *   rpt_powf    : f64 core (log2 abs err 1.5e-12, exp2 rel err 1.4e-14), rounded
 *                 once to f32 -> <= 0.5001 ulp
 *   rpt_expf, rpt_logf : the same cores (2^(x log2 e), log2(x) ln 2), rounded once to f32
 * Coefficients come from tools/gen_strict_math_coeffs.py (mpmath chebyfit).
 *
 * Compile every translation unit that includes this with -ffp-contract=off:
 * the only fused operations are the explicit __builtin_fma[f] below.
 */
#ifndef RPT_STRICT_MATH_H
#define RPT_STRICT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RPT_HD __host__ __device__ inline
#else
#define RPT_HD static inline
#endif

RPT_HD uint32_t rpt_f2u(float x)  { return __builtin_bit_cast(uint32_t, x); }
RPT_HD float    rpt_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
RPT_HD uint64_t rpt_d2u(double x) { return __builtin_bit_cast(uint64_t, x); }
RPT_HD double   rpt_u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

/* ---- sin/cos -------------------------------------------------------------
 * Cody-Waite reduction by pi/2 in three f32 pieces, then minimax polynomials
 * on [-pi/4, pi/4]:  sin r = r + r^3 S(r^2),  cos r = 1 + r^2 C(r^2).        */
RPT_HD void rpt_sincosf(float x, float* sn, float* cs)
{
    const float TWO_OVER_PI = 0x1.45f306p-1f;
    const float MAGIC       = 12582912.0f;           /* 1.5 * 2^23: round-to-nearest-int trick */
    const float PIO2_HI     = 0x1.92p+0f;            /* 1.5703125 */
    const float PIO2_MID    = 0x1.fb4p-12f;
    const float PIO2_LO     = 0x1.4442d2p-24f;

    float t = __builtin_fmaf(x, TWO_OVER_PI, MAGIC);
    float j = t - MAGIC;
    uint32_t q = rpt_f2u(t);                          /* low 2 bits = quadrant (mod 4) */
    float r = __builtin_fmaf(-j, PIO2_HI, x);
    r = __builtin_fmaf(-j, PIO2_MID, r);
    r = __builtin_fmaf(-j, PIO2_LO, r);
    float r2 = r * r;

    float ps = 0x1.6da906p-19f;
    ps = __builtin_fmaf(ps, r2, -0x1.a01366p-13f);
    ps = __builtin_fmaf(ps, r2, 0x1.11110ep-7f);
    ps = __builtin_fmaf(ps, r2, -0x1.555556p-3f);
    float s = __builtin_fmaf(r * r2, ps, r);

    float pc = -0x1.24636p-22f;
    pc = __builtin_fmaf(pc, r2, 0x1.a0124cp-16f);
    pc = __builtin_fmaf(pc, r2, -0x1.6c16bap-10f);
    pc = __builtin_fmaf(pc, r2, 0x1.555556p-5f);
    pc = __builtin_fmaf(pc, r2, -0.5f);
    float c = __builtin_fmaf(r2, pc, 1.0f);

    float so = (q & 1u) ? c : s;
    float co = (q & 1u) ? s : c;
    so = (q & 2u) ? -so : so;
    co = ((q + 1u) & 2u) ? -co : co;
    *sn = so;
    *cs = co;
}

RPT_HD float rpt_sinf(float x) { float s, c; rpt_sincosf(x, &s, &c); return s; }
RPT_HD float rpt_cosf(float x) { float s, c; rpt_sincosf(x, &s, &c); return c; }
RPT_HD float rpt_tanf(float x) { float s, c; rpt_sincosf(x, &s, &c); return s / c; }

/* One Horner step p * f + c with a literal coefficient.  On the device the coefficient is asked for in a SCALAR register pair: left to
 * itself the compiler takes v_fmac_f64, whose addend is the destination, and fills that with two v_mov_b32 per coefficient — vector
 * instructions that do no arithmetic (two per f64 step of every log2 / exp2; the render kernels are bound by vector instruction issue
 * and the scalar unit has room).  The same fused multiply-add on the same values either way. */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RPT_STRICT_MATH_PLAIN_HORNER)
static __device__ __forceinline__ double rpt_horner(double p, double f, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(f), "s"(c));
    return r;
}
#else
#define rpt_horner(p, f, c) __builtin_fma((p), (f), (c))
#endif

/* ---- log2 core in f64 ------------------------------------------------------
 * x > 0 finite.  x = m * 2^e with m in [sqrt(1/2), sqrt(2)), f = m - 1,
 * log2(x) = e + f * P(f), P of degree 13 (abs error 1.5e-12).               */
RPT_HD double rpt_log2_core(float x)
{
    uint32_t ix = rpt_f2u(x);
    int32_t eadj = 0;
    if (ix < 0x00800000u) {                            /* f32 subnormal: scale exactly */
        ix = rpt_f2u(x * 8388608.0f);
        eadj = -23;
    }
    uint32_t tmp = ix - 0x3f3504f3u;                   /* sqrt(1/2) */
    int32_t e = ((int32_t)tmp) >> 23;
    float m = rpt_u2f(ix - ((uint32_t)e << 23));
    double f = (double)m - 1.0;
    double p = -0x1.16c9192143833p-4;
    p = rpt_horner(p, f, 0x1.0c91c8ec04459p-3);
    p = rpt_horner(p, f, -0x1.140c67e94fcdep-3);
    p = rpt_horner(p, f, 0x1.0b872a83448f7p-3);
    p = rpt_horner(p, f, -0x1.23e34ff2dfed7p-3);
    p = rpt_horner(p, f, 0x1.480da32d9077ap-3);
    p = rpt_horner(p, f, -0x1.7186b110dbb40p-3);
    p = rpt_horner(p, f, 0x1.a61c50cd4bc62p-3);
    p = rpt_horner(p, f, -0x1.ec6f47f38905dp-3);
    p = rpt_horner(p, f, 0x1.2776b48f57946p-2);
    p = rpt_horner(p, f, -0x1.7154784e5d08ep-2);
    p = rpt_horner(p, f, 0x1.ec709de7df48dp-2);
    p = rpt_horner(p, f, -0x1.71547651d9376p-1);
    p = rpt_horner(p, f, 0x1.71547652b5270p+0);
    return __builtin_fma(p, f, (double)(e + eadj));
}

/* ---- 2^t in f64, t clamped to [-200, 200], returned as f32 ----------------- *
 * t is not a NaN (the callers see to it): min / max are then the two selects they replace, in one instruction each on the device. */
RPT_HD float rpt_exp2_core(double t)
{
    t = __builtin_fmin(t, 200.0);
    t = __builtin_fmax(t, -200.0);
    double kd = __builtin_rint(t);                     /* round to nearest, ties to even (default rounding mode): one instruction on both sides */
    int32_t n = (int32_t)kd;                            /* |kd| <= 200: exact */
    double r = t - kd;                                  /* r in [-0.5, 0.5] */
    double p = 0x1.b6571de2f2351p-24;
    p = rpt_horner(p, r, 0x1.63ef969a64d3cp-20);
    p = rpt_horner(p, r, 0x1.ffcb76789860fp-17);
    p = rpt_horner(p, r, 0x1.43088e257f341p-13);
    p = rpt_horner(p, r, 0x1.5d87fe908f88ap-10);
    p = rpt_horner(p, r, 0x1.3b2ab72b175eep-7);
    p = rpt_horner(p, r, 0x1.c6b08d7044119p-5);
    p = rpt_horner(p, r, 0x1.ebfbdff8149f2p-3);
    p = rpt_horner(p, r, 0x1.62e42fefa39f7p-1);
    p = rpt_horner(p, r, 0x1.000000000003dp+0);
    uint64_t bits = rpt_d2u(p) + ((uint64_t)(int64_t)n << 52);   /* exact scaling by 2^n */
    return (float)rpt_u2d(bits);                        /* single rounding to f32 (inf / subnormal handled by the conversion) */
}

/* log2f: Rust f32::log2 semantics for the special cases. */
RPT_HD float rpt_log2f(float x)
{
    uint32_t ix = rpt_f2u(x);
    if (ix - 1u < 0x7f7fffffu)                          /* 0 < x < inf */
        return (float)rpt_log2_core(x);
    if ((ix << 1) == 0u) return -__builtin_inff();      /* +-0 -> -inf */
    if (ix == 0x7f800000u) return x;                    /* +inf */
    return __builtin_nanf("");                          /* negative or NaN */
}

/* powf: Rust f32::powf (C99 powf) semantics for the special cases. */
RPT_HD float rpt_powf(float x, float y)
{
    uint32_t ix = rpt_f2u(x), iy = rpt_f2u(y);
    /* fast path: 0 < x < inf, y finite */
    if ((ix - 1u < 0x7f7fffffu) && ((iy & 0x7fffffffu) < 0x7f800000u)) {
        if (ix == 0x3f800000u) return 1.0f;
        return rpt_exp2_core((double)y * rpt_log2_core(x));
    }
    uint32_t ay = iy & 0x7fffffffu, ax = ix & 0x7fffffffu;
    if (ay == 0u) return 1.0f;                           /* pow(x, +-0) = 1, even for NaN */
    if (ix == 0x3f800000u) return 1.0f;                  /* pow(1, y) = 1, even for NaN */
    if (ax > 0x7f800000u || ay > 0x7f800000u) return __builtin_nanf("");
    /* is y an odd integer / an integer? */
    int yint = 0;                                        /* 0: non-integer, 1: odd, 2: even */
    if (ay >= 0x4b800000u) yint = 2;                     /* |y| >= 2^24: even integer */
    else if (ay >= 0x3f800000u) {
        int k = 150 - (int)(ay >> 23);                   /* fractional mantissa bits, 0..23 */
        uint32_t mant = (ay & 0x007fffffu) | 0x00800000u;
        if ((mant & ((1u << k) - 1u)) == 0u) yint = ((mant >> k) & 1u) ? 1 : 2;
    }
    int xneg = (int)(ix >> 31), yneg = (int)(iy >> 31);
    if (ay == 0x7f800000u) {                             /* y = +-inf */
        if (ax == 0x3f800000u) return 1.0f;              /* pow(-1, +-inf) = 1 */
        int big = ax > 0x3f800000u;
        return (big != yneg) ? __builtin_inff() : 0.0f;
    }
    if (ax == 0u || ax == 0x7f800000u) {                 /* x = +-0 or +-inf */
        int to_inf = (ax == 0u) ? yneg : !yneg;
        float r = to_inf ? __builtin_inff() : 0.0f;
        return (xneg && yint == 1) ? -r : r;
    }
    /* x < 0, finite, y finite non-zero */
    if (yint == 0) return __builtin_nanf("");
    float r = rpt_exp2_core((double)y * rpt_log2_core(rpt_u2f(ax)));
    return (yint == 1) ? -r : r;
}

/* rpt_powf(x, y) for a caller that already holds lx = rpt_log2_core(x) of an x in (0, inf) (for any other x, lx is not read): the same
 * operations on the same values.  (The library's material tables keep the logarithm of a roughness that many samples raise to a power.) */
RPT_HD float rpt_powf_log2x(float x, double lx, float y)
{
    uint32_t ix = rpt_f2u(x), iy = rpt_f2u(y);
    if ((ix - 1u < 0x7f7fffffu) && ((iy & 0x7fffffffu) < 0x7f800000u)) {
        if (ix == 0x3f800000u) return 1.0f;
        return rpt_exp2_core((double)y * lx);
    }
    return rpt_powf(x, y);
}

/* expf / logf (natural): Rust f32::exp / f32::ln semantics for the special cases; the same f64 cores, so <= 0.5001 ulp.
 * Only the project-defined participating media use them (include/rpt.h): the reference's tracer calls neither. */
RPT_HD float rpt_expf(float x)
{
    if (x != x) return x;                               /* NaN (the clamp below would not propagate it) */
    return rpt_exp2_core((double)x * 0x1.71547652b82fep+0);   /* 2^(x * log2 e); +-inf -> the clamp -> inf / 0 */
}

RPT_HD float rpt_logf(float x)
{
    uint32_t ix = rpt_f2u(x);
    if (ix - 1u < 0x7f7fffffu)                          /* 0 < x < inf */
        return (float)(rpt_log2_core(x) * 0x1.62e42fefa39efp-1);   /* ln 2 */
    if ((ix << 1) == 0u) return -__builtin_inff();      /* +-0 -> -inf */
    if (ix == 0x7f800000u) return x;                    /* +inf */
    return __builtin_nanf("");                          /* negative or NaN */
}

#endif /* RPT_STRICT_MATH_H */
