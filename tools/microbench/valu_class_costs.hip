// What one instruction of each class costs a gfx950 SIMD to issue, relative to v_fma_f32 — the missing model behind "the headline
// kernel issues 92 % of what a pure-FMA stream issues": streams of 16 independent instructions of ONE class (inline assembly), 5 waves
// per SIMD on every SIMD, timed with HIP events; tools/runs/r5_class_costs.sh runs it under rocprofv3 for the counted clock.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_class_costs.hip -o /tmp/valu_class_costs
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 2048
#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

#define STREAM_KERNEL(NAME, ASM)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float* out, float seed)                           \
    {                                                                                             \
        float a[16];                                                                              \
        for (int i = 0; i < 16; ++i) a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;              \
        float b = seed * 0.999f, c = seed * 1e-4f;                                                \
        for (int it = 0; it < ITERS; ++it) { R16(ASM) }                                           \
        float s = 0.0f;                                                                           \
        for (int i = 0; i < 16; ++i) s += a[i];                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                  \
    }

#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define A_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define A_SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define A_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
#define A_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define A_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define A_CMPX(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(a[i]), "v"(b) : "s20", "s21");
#define A_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define A_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define A_RSQ(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
#define A_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_FREXP(i) asm volatile("v_frexp_exp_i32_f32 %0, %0" : "+v"(a[i]));
#define A_FIXUP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define A_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define A_FMA_S(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_mov_b32 s20, 0x3f800000" : "+v"(a[i]) : "v"(b), "v"(c) : "s20");
#define A_FMA_SS(i) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_mov_b32 s20, 0x3f800000\n s_mov_b32 s22, 0x40000000" : "+v"(a[i]) : "v"(b), "v"(c) : "s20", "s22");   // (s_mov: no SCC — an s_and here once clobbered the loop's compare and hung the box)

STREAM_KERNEL(k_fma, A_FMA)
STREAM_KERNEL(k_mul, A_MUL)
STREAM_KERNEL(k_add, A_ADD)
STREAM_KERNEL(k_sub, A_SUB)
STREAM_KERNEL(k_mov, A_MOV)
STREAM_KERNEL(k_cndmask, A_CND)
STREAM_KERNEL(k_cmp_vcc, A_CMP)
STREAM_KERNEL(k_cmp_sgpr, A_CMPX)
STREAM_KERNEL(k_max3, A_MAX3)
STREAM_KERNEL(k_rcp, A_RCP)
STREAM_KERNEL(k_rsq, A_RSQ)
STREAM_KERNEL(k_add_u32, A_ADDU)
STREAM_KERNEL(k_mul_lo_u32, A_MULLO)
STREAM_KERNEL(k_frexp_exp, A_FREXP)
STREAM_KERNEL(k_div_fixup, A_FIXUP)
STREAM_KERNEL(k_xor, A_XOR)
STREAM_KERNEL(k_fma_plus_1salu, A_FMA_S)
STREAM_KERNEL(k_fma_plus_2salu, A_FMA_SS)

// f64: 8 independent v_fma_f64 chains, twice per iteration (16 instructions)
__global__ __launch_bounds__(256) void k_fma_f64(float* out, float seed)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + (double)(threadIdx.x + i) * 1e-3;
    double b = seed * 0.999, c = seed * 1e-4;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

template <class K>
static void run(K kernel, float* out, const char* name, int per_iter)
{
    for (int waves_per_simd : {1, 5}) {
        const int wgs = 256 * waves_per_simd;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), 0, 0, out, 1.0001f);
        (void)hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), 0, 0, out, 1.0001f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double insts = (double)wgs * 4.0 * ITERS * per_iter;
        printf("%-18s %d wave(s)/SIMD: %8.3f ms  %.3f wave-instructions of the class per SIMD per ns\n", name, waves_per_simd, best, insts / (1024.0 * best * 1e6));
        fflush(stdout);
    }
}

int main()
{
    float* out;
    (void)hipMalloc(&out, 8192 * 256 * 4);
    run(k_fma, out, "v_fma_f32", 16);
    run(k_mul, out, "v_mul_f32", 16);
    run(k_add, out, "v_add_f32", 16);
    run(k_sub, out, "v_sub_f32", 16);
    run(k_mov, out, "v_mov_b32", 16);
    run(k_cndmask, out, "v_cndmask_b32", 16);
    run(k_cmp_vcc, out, "v_cmp (vcc)", 16);
    run(k_cmp_sgpr, out, "v_cmp (sgpr pair)", 16);
    run(k_max3, out, "v_max3_f32", 16);
    run(k_rcp, out, "v_rcp_f32", 16);
    run(k_rsq, out, "v_rsq_f32", 16);
    run(k_add_u32, out, "v_add_u32", 16);
    run(k_mul_lo_u32, out, "v_mul_lo_u32", 16);
    run(k_frexp_exp, out, "v_frexp_exp_i32", 16);
    run(k_div_fixup, out, "v_div_fixup_f32", 16);
    run(k_xor, out, "v_xor_b32", 16);
    run(k_fma_f64, out, "v_fma_f64", 16);
    run(k_fma_plus_1salu, out, "fma + 1 salu", 16);
    run(k_fma_plus_2salu, out, "fma + 2 salu", 16);
    return 0;
}
