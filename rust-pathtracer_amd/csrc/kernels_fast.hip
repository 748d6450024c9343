// kernels_fast.hip — the SAME kernel sources compiled with relaxed arithmetic (build.py adds
// -fno-hip-fp32-correctly-rounded-divide-sqrt -ffp-contract=fast to this file only): v_rcp/v_rsq-based divide
// and sqrt (~2.5 ulp) and fused multiply-adds.  Selected by RPT_RENDER_FAST_MATH.  NOT bit-identical to the
// reference arithmetic: an ulp-level difference occasionally flips a branch and changes a sample by O(1), so
// this mode is validated statistically (tests/test_gpu_parity.py::test_fast_math_mode_is_statistically_equivalent)
// and is never what bench.py measures.
// Only the render kernels are built here: untile, the u8 conversions and the test probes have no relaxed form.
#define RPT_RENDER_KERNELS_ONLY
#define RPT_NO_MEDIA_KERNELS          // scenes with participating media have no relaxed form (RPT_ERR_UNSUPPORTED)
#define RPT_K(name) name##_fast
#define RPT_LAUNCH_NS rptlaunch_fast
#include "kernels.hip"
