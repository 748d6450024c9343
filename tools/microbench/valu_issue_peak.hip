// How many VALU wave-instructions a gfx950 SIMD issues per cycle at best — the ceiling against which bench.py's
// `valu_issue.insts_per_simd_quad_cycle` of the render kernels is to be read.  Every lane runs CHAINS independent v_fma_f32 chains
// (no memory, no scalar work in the loop), launched so that each SIMD holds W waves (W = 1, 2, 4, 5, 8); the rate is
// (VALU instructions issued) / (SIMDs x seconds x 2.4 GHz), x 4 for "per quad-cycle" as rocprofv3's VALUBusy counts.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_issue_peak.hip -o tools/microbench/valu_issue_peak
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 16
#define ITERS 8192
__global__ __launch_bounds__(256) void fma_chains(float* out, float seed)
{
    float a[CHAINS];
    for (int i = 0; i < CHAINS; ++i) a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
    const float b = seed * 0.999f, c = seed * 1e-4f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) a[i] = __builtin_fmaf(a[i], b, c);
    }
    float s = 0.0f;
    for (int i = 0; i < CHAINS; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float* out;
    hipMalloc(&out, 8192 * 256 * 4);
    const double clock_hz = 2.4e9, simds = 1024.0;
    for (int waves_per_simd : {1, 2, 4, 5, 8}) {
        const int wgs = 256 * waves_per_simd;                                   // 256 CUs x 4 SIMDs: one 4-wave workgroup per CU per wave-per-SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(fma_chains, dim3(wgs), dim3(256), 0, 0, out, 1.0001f);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(fma_chains, dim3(wgs), dim3(256), 0, 0, out, 1.0001f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double insts = (double)wgs * 4.0 * ITERS * CHAINS;                // wave-level v_fma_f32
        const double per_cycle = insts / (simds * best * 1e-3 * clock_hz);
        printf("%d wave(s) per SIMD: %.3f ms -> %.3f VALU wave-instructions per SIMD per cycle = %.2f per quad-cycle\n", waves_per_simd, best, per_cycle, 4.0 * per_cycle);
    }
    return 0;
}
