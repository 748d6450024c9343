// dev_pass.h — which compilation of the device functions this is.  No include guard: read at the top of every re-includable header.
//
// dev_math.h, dev_bsdf.h, dev_media.h, dev_integrator.h and dev_scene_large.h hold FUNCTIONS (and the types only they use) and can be
// included twice by one translation unit:
//   * the normal pass: namespace rptdev.  f32 divide and square root are the short correctly rounded sequences of dev_math.h, whose
//     range tests are tracked per lane and looked at once per sample (dev_math.h, "range tests as trackers");
//   * #define RPT_PLAIN_PASS, include again: namespace rptplain.  The same functions over hipcc's own correctly rounded divide and
//     sqrtf, valid for every operand: what a kernel recomputes a sample with when its trackers say an operand left the range
//     (kernels.hip, sample_guard).  Practically never executed; it is there so that "bit-identical" has no exceptions.
// The types both passes and the host share are in dev_scene.h (namespace rptscene).
#undef RPT_NS
#ifdef RPT_PLAIN_PASS
#define RPT_NS rptplain
#else
#define RPT_NS rptdev
#endif
