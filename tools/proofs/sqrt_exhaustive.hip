// Exhaustive check of short square-root sequences against hipcc's correctly rounded sqrtf, over ALL finite positive inputs
// (2^31 bit patterns incl. denormals; the library only uses the short form inside a guarded exponent range).
//   A: s0 = v_sqrt_f32(x);  h = 0.5 * v_rsq_f32(x);  s = fma(fma(-s0, s0, x), h, s0)
//   B: s0 = v_sqrt_f32(x);  h = 0.5 * v_rcp_f32(s0); s = fma(fma(-s0, s0, x), h, s0)
//   C: s0 = v_sqrt_f32(x);  the neighbours s0 -/+ 1 ulp, picked by the signs of fma(-(s0 -/+ ulp), s0, x) (LLVM's own test, unscaled)
//   D: y = v_rsq_f32(x);  s0 = x * y;  s = fma(fma(-s0, s0, x), 0.5 * y, s0)              (ONE transcendental instruction instead of two)
//   E: D with a second correction: s1 = D's s;  s = fma(fma(-s1, s1, x), 0.5 * y, s1)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/proofs/sqrt_exhaustive.hip -o tools/proofs/sqrt_exhaustive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

__global__ __launch_bounds__(256) void check(unsigned long long* mm, uint32_t* ex)
{
    const uint32_t bits = blockIdx.x * 256u + threadIdx.x;                     // 2^31 threads: every non-negative float
    const float x = u2f(bits);
    if (!(x > 0.0f) || !(x < __builtin_inff())) return;
    const float ref = __builtin_sqrtf(x);
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float e = __builtin_fmaf(-s0, s0, x);
    const float a = __builtin_fmaf(e, 0.5f * __builtin_amdgcn_rsqf(x), s0);
    const float b = __builtin_fmaf(e, 0.5f * __builtin_amdgcn_rcpf(s0), s0);
    const float dn = u2f(f2u(s0) - 1u), up = u2f(f2u(s0) + 1u);
    float c = s0;
    if (__builtin_fmaf(-dn, s0, x) <= 0.0f) c = dn;
    if (__builtin_fmaf(-up, s0, x) > 0.0f) c = up;
    const float y = __builtin_amdgcn_rsqf(x);
    const float g0 = x * y;
    const float d = __builtin_fmaf(__builtin_fmaf(-g0, g0, x), 0.5f * y, g0);
    const float e2 = __builtin_fmaf(__builtin_fmaf(-d, d, x), 0.5f * y, d);
    const bool normal = ((bits >> 23) - 67u) <= 120u;                           // 2^-60 <= x < 2^61: the range the library guards
    const uint32_t base = normal ? 0u : 8u;                                     // counters 0-7: inputs in that range, 8-15: all others
    if (f2u(a) != f2u(ref)) { if (atomicAdd(mm + base + 0, 1ull) < 4 && normal) ex[0 + 0] = bits; }
    if (f2u(b) != f2u(ref)) { if (atomicAdd(mm + base + 1, 1ull) < 4 && normal) ex[4 + 0] = bits; }
    if (f2u(c) != f2u(ref)) { if (atomicAdd(mm + base + 2, 1ull) < 4 && normal) ex[8 + 0] = bits; }
    if (f2u(s0) != f2u(ref)) atomicAdd(mm + base + 3, 1ull);
    if (f2u(d) != f2u(ref)) { if (atomicAdd(mm + base + 4, 1ull) < 4 && normal) ex[12 + 0] = bits; }
    if (f2u(e2) != f2u(ref)) { if (atomicAdd(mm + base + 5, 1ull) < 4 && normal) ex[13 + 0] = bits; }
}

int main()
{
    unsigned long long* mm;
    uint32_t* ex;
    hipMalloc(&mm, 128);
    hipMalloc(&ex, 64);
    hipMemset(mm, 0, 128);
    hipMemset(ex, 0, 64);
    hipLaunchKernelGGL(check, dim3(1u << 23), dim3(256), 0, 0, mm, ex);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    unsigned long long h[16];
    uint32_t e[16];
    hipMemcpy(h, mm, 128, hipMemcpyDeviceToHost);
    hipMemcpy(e, ex, 64, hipMemcpyDeviceToHost);
    printf("inputs in [2^-60, 2^61) (121 * 2^23): mismatches  A (rsq) %llu   B (rcp) %llu   C (neighbours) %llu   raw v_sqrt_f32 %llu\n", h[0], h[1], h[2], h[3]);
    printf("                                      mismatches  D (x * rsq, one correction) %llu   E (two corrections) %llu\n", h[4], h[5]);
    printf("all other positive finite inputs:     mismatches  A %llu   B %llu   C %llu   raw %llu   D %llu   E %llu\n", h[8], h[9], h[10], h[11], h[12], h[13]);
    printf("examples (in range): A %08x  B %08x  C %08x  D %08x  E %08x\n", e[0], e[4], e[8], e[12], e[13]);
    return 0;
}
