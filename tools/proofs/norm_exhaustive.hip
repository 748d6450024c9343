// Exhaustive check of normalize's three quotients with the square root's v_rsq_f32 ALSO serving as the divide's first reciprocal:
//
//   y = v_rsq_f32(x);  s0 = x * y;  len = fma(fma(-s0, s0, x), 0.5 * y, s0)      (the short square root: sqrt_exhaustive.hip, form D)
//   r = fma(fma(-len, y, 1), y, y)                                                 (y ~ 1 / len to ~1.5 ulp instead of v_rcp_f32(len)'s 1)
//   q0 = n * r;  q = v_div_fixup(fma(fma(-len, q0, n), r, q0), len, n)             (the short divide: div_exhaustive.hip)
//
// RESULT (profiles/r4/proofs/norm_exhaustive.txt): 2 mismatches in 1.4e14 — the form is NOT used by the library.
//
// against n / len in hipcc's correctly rounded expansion, for EVERY x with a significand of 23 bits and either exponent parity
// (x in [1, 4): len in [1, 2)) and EVERY numerator significand: 2^24 x 2^23 = 1.4e14 quotients.  Scaling x by 4^k scales y, len and r
// by exact powers of two, scaling n by 2^k likewise, so the exponents need no enumeration (dev_math.h keeps everything normal).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/proofs/norm_exhaustive.hip -o tools/proofs/norm_exhaustive
//   tools/proofs/norm_exhaustive [first_block] [n_blocks]        (x bit patterns in blocks of 2^17; 128 blocks = all)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

__global__ __launch_bounds__(256) void check(uint32_t x_first, unsigned long long* mismatches, uint32_t* examples)
{
    const uint32_t xi = x_first + blockIdx.x * 256u + threadIdx.x;             // 24 bits: exponent parity + significand
    const float x = u2f(0x3F800000u + xi);                                     // [1, 4)
    const float y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y;
    const float len = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * y, s0);
    if (f2u(len) != f2u(__builtin_sqrtf(x))) atomicAdd(mismatches + 2, 1ull);   // (the root itself: must stay 0)
    const float r = __builtin_fmaf(__builtin_fmaf(-len, y, 1.0f), y, y);
    uint32_t bad = 0;
    for (uint32_t nm = 0; nm < (1u << 23); ++nm) {
        const float n = u2f(0x3F800000u | nm);
        const float q0 = n * r;
        const float q = __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-len, q0, n), r, q0), len, n);
        const float ref = n / len;                                             // hipcc's correctly rounded expansion
        if (f2u(q) != f2u(ref)) {
            if (bad == 0) {
                const unsigned long long k = atomicAdd(mismatches + 1, 1ull);  // distinct x with a mismatch
                if (k < 64) { examples[2 * k] = f2u(n); examples[2 * k + 1] = f2u(x); }
            }
            ++bad;
        }
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

int main(int argc, char** argv)
{
    const uint32_t first = argc > 1 ? (uint32_t)atoi(argv[1]) : 0u, count = argc > 2 ? (uint32_t)atoi(argv[2]) : 128u;
    unsigned long long* mm;
    uint32_t* ex;
    hipMalloc(&mm, 24);
    hipMalloc(&ex, 64 * 8);
    hipMemset(mm, 0, 24);
    hipMemset(ex, 0, 64 * 8);
    for (uint32_t b = first; b < first + count && b < 128u; ++b) {
        hipLaunchKernelGGL(check, dim3(512), dim3(256), 0, 0, b << 17, mm, ex);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        if ((b & 15u) == 15u) { printf("block %u of 128 done\n", b + 1); fflush(stdout); }
    }
    unsigned long long h[3];
    uint32_t e[128];
    hipMemcpy(h, mm, 24, hipMemcpyDeviceToHost);
    hipMemcpy(e, ex, 64 * 8, hipMemcpyDeviceToHost);
    printf("x blocks %u..%u of 128 (2^17 arguments each, 2^23 numerators per argument): %llu mismatching quotients, %llu arguments with one; "
           "roots that differ from sqrtf: %llu\n", first, first + count - 1, h[0], h[1], h[2]);
    for (unsigned long long k = 0; k < h[1] && k < 8; ++k) printf("  example: n = %08x  x = %08x\n", e[2 * k], e[2 * k + 1]);
    return h[0] == 0 && h[2] == 0 ? 0 : 1;
}
