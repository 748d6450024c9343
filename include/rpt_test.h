/* rpt_test.h — TEST HOOKS of the MI355X-native path-tracing integrator.  NOT part of the drop-in surface (include/rpt.h) and NOT
 * exported by the shipped library: librpt_hip_test.so — the same objects, linked with these entry points and the probe kernels
 * (python rust-pathtracer_amd/build.py builds both) — exports them, and rpt_build_has_test_hooks() tells which library is loaded.
 * The parity tests use them to localise a frame mismatch to one function. */
#ifndef RPT_TEST_H
#define RPT_TEST_H

#include "rpt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- probes ------------------------------------------------------------------
 * Evaluate one device function over arrays (device pointers), so tests can compare
 * leaf functions with the oracle bit for bit.  */
enum {
    RPT_PROBE_SIN = 0, RPT_PROBE_COS = 1, RPT_PROBE_LOG2 = 2, RPT_PROBE_POW = 3,
    RPT_PROBE_DIV = 4, RPT_PROBE_SQRT = 5, RPT_PROBE_RNG = 6, RPT_PROBE_EXP = 7, RPT_PROBE_LOG = 8,
    RPT_PROBE_DIV3 = 9                /* three quotients by one denominator, the library's shared-reciprocal form: see dev_math.h */
};
int rpt_probe_math(rpt_ctx* ctx, uint32_t fn, const float* a_dev, const float* b_dev,
                   float* out_dev, uint64_t n, void* stream);

/* One integrator function per record, for tests that localise a frame mismatch: records are RPT_PROBE_IN_STRIDE floats
 * in, RPT_PROBE_OUT_STRIDE floats out (u32 values as their bit patterns), device pointers.  Layouts (in -> out):
 *   GEN_RAY        {px, py, offx, offy}; camera = the uploaded scene's, params = {width, height} (host)
 *                  -> {origin[3], direction[3]}                                      camera/pinhole.rs:38-60
 *   HIT_SPHERE     {o[3], d[3], centre[3], radius} -> {hit, t}                       analytical.rs:166-190
 *   HIT_PLANE      {o[3], d[3], normal[3], point[3], min_denom, max_t} -> {hit, t}   analytical.rs:193-204
 *   SAMPLE_LIGHT   {type, position[3], emission[3], radius, area, u[3], v[3], scatter_pos[3], n_lights, scene flags,
 *                   rng state, rng increment, -} -> {normal[3], emission[3], direction[3], dist, pdf, draws}   tracer.rs:173-220
 *   DISNEY_EVAL    {material: rgb[3], emission[3], anisotropic, metallic, roughness, subsurface, specular_tint, sheen,
 *                   sheen_tint, clearcoat, clearcoat_gloss, spec_trans, ior (before finalize), eta, v[3], n[3], l[3]}
 *                  -> {f[3], pdf}                                                    tracer.rs:555-626
 *   DISNEY_SAMPLE  {material (17), eta, v[3], n[3], l_stale[3], rng state, rng increment, -}
 *                  -> {f[3], l[3], pdf, draws}                                       tracer.rs:441-553            */
enum {
    RPT_PROBE_FN_GEN_RAY = 0, RPT_PROBE_FN_HIT_SPHERE = 1, RPT_PROBE_FN_HIT_PLANE = 2, RPT_PROBE_FN_SAMPLE_LIGHT = 3,
    RPT_PROBE_FN_DISNEY_EVAL = 4, RPT_PROBE_FN_DISNEY_SAMPLE = 5, RPT_PROBE_FN_COUNT = 6
};
#define RPT_PROBE_IN_STRIDE 32
#define RPT_PROBE_OUT_STRIDE 16
int rpt_probe_fn(rpt_ctx* ctx, uint32_t fn, const float* in_dev, float* out_dev, uint64_t n, const float* params, void* stream);

/* Ray queries against the uploaded LARGE scene's spheres, for testing the acceleration structure:
 * rays_dev = n x {origin[3], direction[3], max_dist}; out_dev = n x {t (f32 bits), nearest sphere index
 * or 0xFFFFFFFF, any_hit (0/1) with max_dist honoured}.  use_grid = 0 forces the brute-force loops. */
int rpt_probe_rays(rpt_ctx* ctx, const float* rays_dev, uint32_t* out_dev, uint64_t n, uint32_t use_grid, void* stream);

/* Multi-device contexts, after rpt_render / rpt_resident_render: the time in ms from the moment device index `b` (position in
 * rpt_create_multi's list) BEGAN its part of the last render to the moment device index `a` ENDED its part (HIP events on their
 * streams).  Positive for a != b means the two overlapped: what the fan-out inside render() promises (tracer.rs:29-32).  Events
 * of two different physical devices cannot be compared (RPT_ERR_UNSUPPORTED): the probe is for virtual ranks, i.e. repeated
 * device ids.  Waits for both events. */
int rpt_debug_render_overlap_ms(rpt_ctx* ctx, int a, int b, float* ms);

/* What the last launch on the context's first device left for the next one's dispatch (rpt_set_dispatch): per tile of that launch
 * (16x16 pixels, row-major over the device's rows) out[tile * 4 + wave] = the time, in 10 ns, wave `wave` of the tile's last unit
 * held its slot, then from out[4 * n] the dispatch order (position -> tile: a permutation of 0 .. n - 1) and 5 * n more words of
 * development data (tools/dispatch_timeline.py).  `out` holds 10 * capacity_tiles dwords; *n_tiles = n.  Waits for the device. */
int rpt_debug_sched_read(rpt_ctx* ctx, uint32_t* out, uint32_t capacity_tiles, uint32_t* n_tiles);

/* Which instantiation of its kernel class the context's last render launch took on its first device (csrc/launch.h, KernelChoice):
 * bit 0 the table sizes known at compile time, bit 1 the material table (at most 3 primitives), bit 2 its 64-row form (4 primitives),
 * bit 3 the table by class of accepted set (5-12 primitives), bits 8-15 the number of classes then, bits 16-19 the SDF object's
 * compile-time primitive count.  For tests that must know that the kernel they aim at is the one that ran. */
int rpt_debug_kernel_choice(rpt_ctx* ctx, uint32_t* out);

/* Read the environment's knobs (csrc/knobs.h: the library reads them ONCE per process) again: for tests that change one between two
 * scenes or contexts of one process. */
int rpt_debug_reload_knobs(void);

#ifdef __cplusplus
}
#endif
#endif /* RPT_TEST_H */
