// Exhaustive check of the short division sequence against the correctly rounded divide, over ALL pairs of significands.
//
//   r0 = v_rcp_f32(d);  r = fma(fma(-d, r0, 1), r0, r0);          (one Newton step on the reciprocal, shared by every numerator)
//   q0 = n * r;  q = v_div_fixup(fma(fma(-d, q0, n), r, q0), d, n)  (one Markstein correction of the quotient; the fix-up only acts on
//                                                                   zeros, infinities and NaNs — and must leave everything else alone)
//
// For operands whose exponents are far from the ends of the range (no intermediate over- or underflows; dev_math.h guards that)
// every step commutes exactly with scaling n and d by powers of two, and with their signs, so it is enough to compare the 2^23 x 2^23
// significand pairs n, d in [1, 2): 7.0e13 divisions, about a minute of one MI355X.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/proofs/div_exhaustive.hip -o tools/proofs/div_exhaustive
//   tools/proofs/div_exhaustive [first_d_block] [n_d_blocks]      (d significands in blocks of 2^17; 64 blocks = all)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

__global__ __launch_bounds__(256) void check(uint32_t d_first, unsigned long long* mismatches, uint32_t* examples)
{
    const uint32_t dm = d_first + blockIdx.x * 256u + threadIdx.x;             // d's 23 significand bits
    const float d = u2f(0x3F800000u | dm);
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    uint32_t bad = 0;
    for (uint32_t nm = 0; nm < (1u << 23); ++nm) {
        const float n = u2f(0x3F800000u | nm);
        const float q0 = n * r;
        const float q = __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-d, q0, n), r, q0), d, n);
        const float ref = n / d;                                               // hipcc's correctly rounded expansion
        if (f2u(q) != f2u(ref)) {
            if (bad == 0) {
                const unsigned long long k = atomicAdd(mismatches + 1, 1ull);  // distinct d with a mismatch
                if (k < 64) { examples[2 * k] = f2u(n); examples[2 * k + 1] = f2u(d); }
            }
            ++bad;
        }
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// The scaling / sign argument, spot-checked: 2^38 pseudo-random pairs with exponents anywhere in the guarded range [2^-61, 2^60)
// and both signs (and zero numerators every so often), the shared-reciprocal form with v_div_fixup as the library runs it.
__device__ __forceinline__ uint32_t mix32(uint32_t v)
{
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
__global__ __launch_bounds__(256) void check_scaled(uint32_t round, unsigned long long* mismatches)
{
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    uint32_t h = mix32(id ^ (round * 0x9E3779B9u));
    uint32_t bad = 0;
    for (uint32_t k = 0; k < 1024u; ++k) {
        h = mix32(h + k);
        const uint32_t hd = mix32(h ^ 0x85EBCA6Bu);
        const uint32_t ed = 67u + (hd >> 9) % 120u, en = 67u + (h >> 9) % 120u;       // biased exponents 67 .. 186
        const float d = u2f((hd & 0x80000000u) | (ed << 23) | (hd & 0x007FFFFFu));
        float n = u2f((h & 0x80000000u) | (en << 23) | (mix32(hd) & 0x007FFFFFu));
        if ((h & 0x3F0u) == 0u) n = (h & 1u) ? 0.0f : -0.0f;
        float r = __builtin_amdgcn_rcpf(d);
        r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
        const float q0 = n * r;
        const float q = __builtin_amdgcn_div_fixupf(__builtin_fmaf(__builtin_fmaf(-d, q0, n), r, q0), d, n);
        if (f2u(q) != f2u(n / d)) ++bad;
    }
    if (bad) atomicAdd(mismatches + 2, (unsigned long long)bad);
}

int main(int argc, char** argv)
{
    const uint32_t first = argc > 1 ? (uint32_t)atoi(argv[1]) : 0u, count = argc > 2 ? (uint32_t)atoi(argv[2]) : 64u;
    unsigned long long* mm;
    uint32_t* ex;
    hipMalloc(&mm, 24);
    hipMalloc(&ex, 64 * 8);
    hipMemset(mm, 0, 24);
    hipMemset(ex, 0, 64 * 8);
    for (uint32_t b = first; b < first + count && b < 64u; ++b) {
        hipLaunchKernelGGL(check, dim3((1u << 17) / 256u), dim3(256), 0, 0, b << 17, mm, ex);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        unsigned long long h[2];
        hipMemcpy(h, mm, 16, hipMemcpyDeviceToHost);
        printf("d block %2u / 64 done: %llu mismatching pairs so far, in %llu denominators\n", b + 1, h[0], h[1]);
        fflush(stdout);
    }
    for (uint32_t round = 0; round < 64u; ++round) hipLaunchKernelGGL(check_scaled, dim3(1u << 14), dim3(256), 0, 0, round, mm);   // 64 x 2^22 x 2^10 = 2^38 pairs
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    unsigned long long h[3];
    uint32_t e[128];
    hipMemcpy(h, mm, 24, hipMemcpyDeviceToHost);
    hipMemcpy(e, ex, sizeof(e), hipMemcpyDeviceToHost);
    printf("scaled / signed spot check: 2^38 pairs with exponents in [2^-61, 2^60), both signs, zero numerators: %llu mismatches\n", h[2]);
    for (unsigned long long k = 0; k < h[1] && k < 8; ++k) printf("  example: n = %08x  d = %08x\n", e[2 * k], e[2 * k + 1]);
    printf("RESULT %u d-blocks of 2^17 x 2^23 numerators: %llu mismatches\n", count, h[0]);
    return (h[0] || h[2]) ? 1 : 0;
}
