// What a per-lane gather costs the texture path as a function of HOW MANY LANES are active — the question behind the large-scene
// kernel's grid walks (profiles/r6/c5_mem/: TD busy 96 % of the launch, 24 busy cycles per vector-memory wave-instruction, while the
// walks' loads run with 19-29 % of the lanes).  Each wave issues the same number of global_load_dwordx4 / x2 / x1 gathers from a 1 MB
// table (L2-resident, like the grid's lists) with 64, 32, 16, 8 or 4 lanes active; addresses are random per lane (distinct cache lines)
// or shared by groups of lanes.  If the time does not fall with the active lanes, a gather is priced per wave-instruction, and the
// walks' idle lanes are free to carry other list entries of the same rays.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_lane_cost.hip -o /tmp/gather_lane_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITERS 2048
template <int WIDTH>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ table, uint32_t mask_entries, uint32_t active, uint32_t spread, float* out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    float acc = 0.0f;
    if (lane < active) {
        for (int it = 0; it < ITERS; ++it) {
            idx = idx * 1664525u + 1013904223u;
            // `spread`: lanes lane / spread share an index (1: every lane its own line)
            uint32_t k = ((idx >> 8) ^ ((lane / spread) * 0x9E3779B9u)) & mask_entries;
            if (spread > 1) k = (uint32_t)__shfl((int)k, (int)((lane / spread) * spread)) + (lane % spread);
            k &= mask_entries;
            if (WIDTH == 4) { const float4 v = table[k]; acc += v.x + v.w; }
            else if (WIDTH == 2) { const float2 v = reinterpret_cast<const float2*>(table)[k]; acc += v.x + v.y; }
            else { acc += reinterpret_cast<const float*>(table)[k]; }
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

int main()
{
    const uint32_t entries = 1u << 16;                       // 64 Ki float4 = 1 MB
    std::vector<float> h(entries * 4, 1.0f);
    float4* table; float* out;
    (void)hipMalloc(&table, entries * 16); (void)hipMalloc(&out, 256 * 5 * 256 * 4);
    (void)hipMemcpy(table, h.data(), entries * 16, hipMemcpyHostToDevice);
    const int wgs = 256 * 5;                                 // 5 waves per SIMD, like the render kernels
    printf("%-6s %-7s %-7s %10s %14s\n", "width", "active", "spread", "ms", "ns per wave-gather per CU");
    auto run = [&](int width, uint32_t active, uint32_t spread) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            if (width == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(wgs), dim3(256), 0, 0, table, entries - 1, active, spread, out);
            else if (width == 2) hipLaunchKernelGGL(gather_kernel<2>, dim3(wgs), dim3(256), 0, 0, table, entries * 2 - 1, active, spread, out);
            else hipLaunchKernelGGL(gather_kernel<1>, dim3(wgs), dim3(256), 0, 0, table, entries * 4 - 1, active, spread, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        // per CU: 20 waves x ITERS gathers each
        printf("x%-5d %-7u %-7u %10.3f %14.2f\n", width, active, spread, best, best * 1e6 / (20.0 * ITERS));
    };
    for (int width : {4, 2, 1})
        for (uint32_t active : {64u, 32u, 16u, 8u, 4u})
            run(width, active, 1);
    for (uint32_t spread : {2u, 4u, 8u}) { run(4, 64, spread); run(4, 16, spread); }
    return 0;
}
