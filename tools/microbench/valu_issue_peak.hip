// How many VALU wave-instructions a gfx950 SIMD issues per cycle at best — the ceiling against which bench.py's
// `valu_issue.insts_per_simd_quad_cycle` of the render kernels is to be read.  Round 5: the streams are written in inline assembly
// (round 4's C loop was SLP-packed by hipcc into v_pk_fma_f32 — half as many instructions as it was priced for), the instruction
// count comes from the program (checked against SQ_INSTS_VALU) and the clock from GRBM_GUI_ACTIVE of the same launch
// (tools/runs/r5_valu_peak.sh), not from an assumed 2.4 GHz.
//   stream 0: v_fma_f32, 16 independent chains          stream 1: v_pk_fma_f32, 8 independent chains (2 fmas per lane each)
//   stream 2: v_fma_f32 / v_mul_f32 / v_add_f32 / v_cndmask_b32 mixed, 16 chains (closer to what a path kernel issues)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_issue_peak.hip -o /tmp/valu_issue_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITERS 4096
#define UNROLL 16
template <int STREAM>
__global__ __launch_bounds__(256) void stream_kernel(float* out, float seed)
{
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
    float b = seed * 0.999f, c = seed * 1e-4f;
    typedef float float2_t __attribute__((ext_vector_type(2)));
    float2_t p[8];
    for (int i = 0; i < 8; ++i) p[i] = float2_t{a[2 * i], a[2 * i + 1]};
    float2_t pb = float2_t{b, b}, pc = float2_t{c, c};
    for (int it = 0; it < ITERS; ++it) {
        if (STREAM == 0) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (STREAM == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
        } else {
#pragma unroll
            for (int i = 0; i < UNROLL; i += 4) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i + 1]) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i + 2]) : "v"(c));
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i + 3]) : "v"(b));
            }
        }
    }
    float s = 0.0f;
    for (int i = 0; i < 16; ++i) s += a[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int STREAM>
static void run(float* out, const char* name)
{
    for (int waves_per_simd : {1, 2, 4, 5, 8}) {
        const int wgs = 256 * waves_per_simd;                                   // 256 CUs x 4 SIMDs: one 4-wave workgroup per CU per wave-per-SIMD
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(stream_kernel<STREAM>, dim3(wgs), dim3(256), 0, 0, out, 1.0001f);
        (void)hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(stream_kernel<STREAM>, dim3(wgs), dim3(256), 0, 0, out, 1.0001f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double insts = (double)wgs * 4.0 * ITERS * 16;                    // wave-level VALU instructions of the loop
        printf("%-10s %d wave(s)/SIMD: %.3f ms, %.4g VALU wave-instructions -> %.3f per SIMD per ns (x 4 / GHz = per quad-cycle; the clock: GRBM_GUI_ACTIVE / 8 / ns)\n",
               name, waves_per_simd, best, insts, insts / (1024.0 * best * 1e6));
    }
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 8192 * 256 * 4);
    run<0>(out, "v_fma_f32");
    run<1>(out, "v_pk_fma");
    run<2>(out, "mixed");
    return 0;
}
