// v_cndmask_b32 back to back issued at a SEVENTH of the v_fma_f32 rate in tools/microbench/valu_class_costs.hip.  Which shape is slow?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/cndmask_costs.hip -o /tmp/cndmask_costs
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
        asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_mov_b64 s[20:21], vcc" : : "v"(a[0]), "v"(b) : "vcc", "s20", "s21");   \
        for (int it = 0; it < ITERS; ++it) { R16(ASM) }                                           \
        float s = 0.0f;                                                                           \
        for (int i = 0; i < 16; ++i) s += a[i];                                                   \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                  \
    }
#define A_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define A_CND_VCC(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define A_CND_SGPR(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b));
#define A_CND_OUT(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(b), "v"(c));                 /* no read of the destination */
#define A_ALT(i) if ((i) & 1) { A_FMA(i) } else { A_CND_VCC(i) }
#define A_3AND1(i) if (((i) & 3) == 3) { A_FMA(i) } else { A_CND_VCC(i) }
#define A_1AND3(i) if (((i) & 3) == 0) { A_CND_VCC(i) } else { A_FMA(i) }
#define A_CND_INLINE(i) asm volatile("v_cndmask_b32 %0, %0, 1.0, vcc" : "+v"(a[i]));
#define A_CND_E64_VCC(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                /* the same operation in the 8-byte encoding */
#define A_CND_NOP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n s_nop 0" : "+v"(a[i]) : "v"(b));
#define A_CND_NOP3(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n s_nop 3" : "+v"(a[i]) : "v"(b));
#define A_CND_MIX(i) if ((i) & 1) { A_CND_E64_VCC(i) } else { A_CND_VCC(i) }
#define A_CND_MOV(i) if ((i) & 1) { asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b)); } else { A_CND_VCC(i) }
STREAM_KERNEL(k_fma, A_FMA)
STREAM_KERNEL(k_cnd_e64_vcc, A_CND_E64_VCC)
STREAM_KERNEL(k_cnd_nop, A_CND_NOP)
STREAM_KERNEL(k_cnd_nop3, A_CND_NOP3)
STREAM_KERNEL(k_cnd_mix, A_CND_MIX)
STREAM_KERNEL(k_cnd_mov, A_CND_MOV)
STREAM_KERNEL(k_cnd_vcc, A_CND_VCC)
STREAM_KERNEL(k_cnd_sgpr, A_CND_SGPR)
STREAM_KERNEL(k_cnd_out, A_CND_OUT)
STREAM_KERNEL(k_cnd_inline, A_CND_INLINE)
STREAM_KERNEL(k_alt, A_ALT)
STREAM_KERNEL(k_3and1, A_3AND1)
STREAM_KERNEL(k_1and3, A_1AND3)
template <class K>
static void run(K kernel, float* out, const char* name)
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
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-34s %d wave(s)/SIMD: %8.3f ms  %.3f wave-instructions per SIMD per ns\n", name, waves_per_simd, best, (double)wgs * 4.0 * ITERS * 16 / (1024.0 * best * 1e6));
        fflush(stdout);
    }
}
int main()
{
    float* out;
    (void)hipMalloc(&out, 8192 * 256 * 4);
    run(k_fma, out, "v_fma_f32");
    run(k_cnd_vcc, out, "v_cndmask vcc, dst = src0");
    run(k_cnd_sgpr, out, "v_cndmask s[20:21], dst = src0");
    run(k_cnd_out, out, "v_cndmask vcc, dst write-only");
    run(k_cnd_inline, out, "v_cndmask vcc, src1 inline 1.0");
    run(k_cnd_e64_vcc, out, "v_cndmask_b32_e64 ..., vcc");
    run(k_cnd_nop, out, "v_cndmask vcc + s_nop 0");
    run(k_cnd_nop3, out, "v_cndmask vcc + s_nop 3");
    run(k_cnd_mix, out, "cndmask e32 / e64 alternating");
    run(k_cnd_mov, out, "cndmask e32 / v_mov alternating");
    run(k_alt, out, "cndmask / fma alternating");
    run(k_3and1, out, "3 cndmask + 1 fma");
    run(k_1and3, out, "1 cndmask + 3 fma");
    return 0;
}
