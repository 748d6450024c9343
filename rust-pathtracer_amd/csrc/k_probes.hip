// k_probes.hip — TEST BUILD ONLY (librpt_hip_test.so, include/rpt_test.h): one device function per record, so that tests can compare
// leaf functions and grid queries with the oracle bit for bit.  Built like small scenes' megakernel (range trackers + a second, plain
// computation of a flagged record).
#define RPT_WITH_PROBES
#include "kernel_common.h"

// The test probes (dev_probes.h holds the bodies).  RPT_MATH_MODE 2: like a sample of the render kernels, a record whose operands left
// the range of the short sequences is computed again with the plain operations.
RPT_DEV bool probe_begin()
{
#if RPT_MATH_MODE == 2
    guard_reset();
#endif
    return true;
}
RPT_DEV bool probe_redo()
{
#if RPT_MATH_MODE == 2
    return !guard_sample_ok();
#else
    return false;
#endif
}
#if RPT_MATH_MODE == 2
#define RPT_PROBE_PLAIN(call) rptplain::call
#else
#define RPT_PROBE_PLAIN(call) call
#endif

__global__ __launch_bounds__(256) void probe_math_kernel(uint32_t fn, const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    probe_begin();
    float r = probe_math_body(fn, a[i], b[i], i);
    if (probe_redo()) r = RPT_PROBE_PLAIN(probe_math_body(fn, a[i], b[i], i));
    out[i] = r;
}

__global__ __launch_bounds__(256) void probe_fn_kernel(uint32_t fn, const DevCamera cam, const float* __restrict__ in, float* __restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = in + i * RPT_PROBE_IN_STRIDE;
    float* o = out + i * RPT_PROBE_OUT_STRIDE;
    probe_begin();
    probe_fn_body(fn, cam, r, o);
    if (probe_redo()) RPT_PROBE_PLAIN(probe_fn_body(fn, cam, r, o));
}

__global__ __launch_bounds__(256) void probe_rays_kernel(const SceneLarge sc, const float* __restrict__ rays, uint32_t* __restrict__ out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    probe_begin();
    probe_rays_body(sc, rays + i * 7, out, i);
    if (probe_redo()) RPT_PROBE_PLAIN(probe_rays_body(sc, rays + i * 7, out, i));
}

namespace rptlaunch {

hipError_t probe_math(uint32_t fn, const float* a, const float* b, float* out, uint64_t n, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(probe_math_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, fn, a, b, out, n);
    return hipGetLastError();
}

hipError_t probe_fn(uint32_t fn, const DevCamera& cam, const float* in, float* out, uint64_t n, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(probe_fn_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, fn, cam, in, out, n);
    return hipGetLastError();
}
hipError_t probe_rays(const SceneLarge& sc, const float* rays, uint32_t* out, uint64_t n, hipStream_t st)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL(probe_rays_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, sc, rays, out, n);
    return hipGetLastError();
}

}  // namespace rptlaunch
