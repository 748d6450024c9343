/* rpt.h — C ABI of the MI355X-native path-tracing integrator.
 *
 * This is the drop-in boundary for ONE hot path of markusmoenig/rust-pathtracer:
 *     Tracer::render(&mut self, buffer: &mut ColorBuffer)
 *         rust-pathtracer/src/tracer.rs:22-123   (rayon scanline loop + bounce loop)
 * and its callees (direct_light :126, sample_light :173, Disney BSDF :223-626,
 * Scene::sample_lights scene.rs:36-86, Pinhole::gen_ray camera/pinhole.rs:38-60,
 * State/Material finalize globals.rs:50-62 / material.rs:117-131) plus the workload
 * scene renderer/src/analytical.rs:13-204.
 *
 * The reference has no FFI of its own (the path is a plain Rust method), so these
 * entry points are what a Rust `extern "C"` block for that method binds; the
 * binding is shown in INTEGRATION.md and rust/gpu_tracer.rs.
 *
 * Conventions
 *   - every function returns 0 on success or a negative rpt_status; nothing unwinds
 *     across the boundary (the reference's render() cannot fail: tracer.rs:22);
 *   - plain pointers and sizes only; a context is used by one thread at a time
 *     (render takes &mut self in the reference: tracer.rs:22);
 *   - all arithmetic is f32 (rust-pathtracer/src/lib.rs:6).
 */
#ifndef RPT_H
#define RPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPT_ABI_VERSION 4u            /* 4: the A/B-only render flags and the test hooks left this header (include/rpt_test.h) */

typedef enum rpt_status {
    RPT_OK              =  0,
    RPT_ERR_INVALID_ARG = -1,
    RPT_ERR_NO_DEVICE   = -2,   /* no usable gfx950 device: the product has no CPU fallback */
    RPT_ERR_HIP         = -3,   /* a HIP runtime call failed; see rpt_last_error */
    RPT_ERR_NO_SCENE    = -4,
    RPT_ERR_UNSUPPORTED = -5,
    RPT_ERR_RCCL        = -6    /* an RCCL call failed (or librccl.so.1 could not be loaded); see rpt_last_error */
} rpt_status;

/* ---- scene as data ---------------------------------------------------------
 * The reference's Scene is code (trait callbacks, scene.rs:5-90) and cannot run
 * on the device, so the scene crosses the boundary as plain data.  The closed set
 * below covers renderer/src/analytical.rs exactly, including its material
 * layering: closest_hit overwrites fields of state.material each time a primitive
 * is accepted (analytical.rs:56-58, 82-85, 115-116), so a material is a PATCH —
 * a field mask plus values — applied over Material::new() (material.rs:82-114) in
 * primitive order.                                                             */

enum {                                /* rpt_material.mask bits */
    RPT_MAT_RGB             = 1u << 0,
    RPT_MAT_EMISSION        = 1u << 1,
    RPT_MAT_ANISOTROPIC     = 1u << 2,
    RPT_MAT_METALLIC        = 1u << 3,
    RPT_MAT_ROUGHNESS       = 1u << 4,
    RPT_MAT_SUBSURFACE      = 1u << 5,
    RPT_MAT_SPECULAR_TINT   = 1u << 6,
    RPT_MAT_SHEEN           = 1u << 7,
    RPT_MAT_SHEEN_TINT      = 1u << 8,
    RPT_MAT_CLEARCOAT       = 1u << 9,
    RPT_MAT_CLEARCOAT_GLOSS = 1u << 10,
    RPT_MAT_SPEC_TRANS      = 1u << 11,
    RPT_MAT_IOR             = 1u << 12,
    RPT_MAT_ALL             = (1u << 13) - 1u,   /* every BSDF field */
    RPT_MAT_MEDIUM          = 1u << 13           /* Material.medium (all four fields at once); only read under RPT_SCENE_MEDIA */
};

enum { RPT_MEDIUM_NONE = 0, RPT_MEDIUM_ABSORB = 1, RPT_MEDIUM_SCATTER = 2, RPT_MEDIUM_EMISSIVE = 3 };   /* MediumType, material.rs:8-13 */

enum {                                /* rpt_material.proc_kind */
    RPT_PROC_NONE        = 0,
    /* rgb = checker(dir.x/dir.y*s + o, dir.z/dir.y*s + o) ? a : b   (analytical.rs:107-115);
     * proc_params = {s, o, a, b}.  Sets rgb regardless of RPT_MAT_RGB.          */
    RPT_PROC_CHECKER_DIR = 1
};

typedef struct rpt_material {         /* user-set fields of material.rs:48-78 that the tracer reads */
    uint32_t mask;
    uint32_t proc_kind;
    float rgb[3];
    float emission[3];
    float anisotropic;
    float metallic;
    float roughness;
    float subsurface;
    float specular_tint;
    float sheen;
    float sheen_tint;
    float clearcoat;
    float clearcoat_gloss;
    float spec_trans;
    float ior;
    float proc_params[4];
    /* Material.medium (material.rs:16-21, 75, 107): see "participating media" below */
    uint32_t medium_type;
    float medium_density;
    float medium_color[3];
    float medium_anisotropy;
} rpt_material;

typedef struct rpt_sphere {           /* analytical.rs:166-190 */
    float    center[3];
    float    radius;
    uint32_t material;                /* index into rpt_scene_desc.materials */
} rpt_sphere;

typedef struct rpt_plane {            /* analytical.rs:193-204 generalised: dot(point - o, n) / dot(n, d) */
    float    normal[3];
    float    point[3];
    float    min_denom;               /* reject |dot(n,d)| <= min_denom (1e-4 in the reference) */
    uint32_t material;
    float    max_t;                   /* > 0: also reject t > max_t (a floor of finite reach); 0 = the reference's infinite plane */
} rpt_plane;

enum { RPT_LIGHT_RECTANGULAR = 0, RPT_LIGHT_SPHERICAL = 1, RPT_LIGHT_DISTANT = 2 };  /* globals.rs:69-73 */

/* globals.rs:76-84.  The reference samples and intersects only SPHERICAL lights (tracer.rs:175-217, scene.rs:68).
 * With RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES the other two declared types work too — project-defined, after the
 * renderer tracer.rs is a port of (its comments "Required for quad lights with single sided emission",
 * tracer.rs:148, and "No MIS for distant light", tracer.rs:158, are that renderer's), in the operation order
 * written in oracle/rpt_oracle.hpp:
 *   RECTANGULAR  the parallelogram position + a*u + b*v, a,b in [0,1]; area = |u x v| (the caller supplies it);
 *                sampled uniformly (2 draws), pdf = dist^2 / (area * |n.dir|), n = normalize(u x v); emits from the
 *                side n points to; a ray reaching it from that side ends the path like a spherical light does.
 *   DISTANT      direction = normalize(position), no draws, dist = +inf, pdf = 1, area = 0 (so no MIS weight);
 *                never intersected. */
typedef struct rpt_light {
    uint32_t type;
    float    position[3];
    float    emission[3];
    float    u[3];
    float    v[3];
    float    radius;
    float    area;                    /* 4*pi*r*r for a spherical light (light.rs:22) */
} rpt_light;

typedef struct rpt_camera {           /* camera/pinhole.rs:6-25 */
    float origin[3];
    float center[3];
    float fov_deg;
} rpt_camera;

enum {
    RPT_BG_CONSTANT   = 0,            /* colour_a * scale */
    /* t = 0.5*(dir.y+1); to_linear((1-t)*colour_a + t*colour_b) * scale
     * (analytical.rs:28-32, to_linear = powf(gamma) per channel, scene.rs:32-34) */
    RPT_BG_GRADIENT_Y = 1
};

typedef struct rpt_background {
    uint32_t kind;
    float    colour_a[3];
    float    colour_b[3];
    float    gamma;
    float    scale;
} rpt_background;

/* ---- procedural SDF object (BASELINE.json configs[3]; the reference has no SDF scene, Readme.md:18) ----
 * One implicit surface per scene: the polynomial smooth union of a list of primitives,
 *     d = fold(smin_k) over prims,   smin_k(a,b) = min(a,b) - h*h*k*0.25,  h = max(k - |a-b|, 0) * (1/k)
 * (1/k is the f32 quotient 1.0f / k, computed once)
 * found by sphere marching from t = 0:  p = o + t*d;  hit when sdf(p) < hit_eps * t;  t += sdf(p);
 * miss after max_steps steps or when t > max_t.  Normal = normalised tetrahedral gradient with step
 * normal_eps.  The object is tested AFTER the spheres and planes (accepted when nearer) and counts as
 * an occluder in any_hit.  All arithmetic is f32 in the order written in oracle/rpt_oracle.hpp. */
enum { RPT_SDF_SPHERE = 0, RPT_SDF_TORUS_Y = 1 };

typedef struct rpt_sdf_prim {
    uint32_t kind;
    float    center[3];
    float    params[2];               /* sphere: {radius, -}; torus around the y axis: {major R, minor r} */
} rpt_sdf_prim;

typedef struct rpt_sdf {
    uint32_t n_prims;                 /* 0 = no SDF object; at most 8 */
    uint32_t max_steps;               /* at most 65536 */
    uint32_t material;
    float    smooth_k;
    float    hit_eps;
    float    max_t;
    float    normal_eps;
    const rpt_sdf_prim* prims;
} rpt_sdf;

enum {                                /* rpt_scene_desc.flags */
    /* Scene::any_hit honours max_dist.  OFF reproduces analytical.rs:130, which
     * ignores it (anything along the shadow ray occludes). */
    RPT_SCENE_ANYHIT_USES_MAX_DIST = 1u << 0,
    /* Sample and intersect RECTANGULAR and DISTANT lights (project-defined, see rpt_light).  OFF reproduces the
     * reference, whose sample_light and Scene::sample_lights only know LightType::Spherical (tracer.rs:175-217,
     * scene.rs:68): such lights are still picked by the light-index draw but contribute nothing. */
    RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES = 1u << 1,
    /* Participating media (project-defined, see below).  OFF reproduces the reference, which never reads a Medium. */
    RPT_SCENE_MEDIA = 1u << 2
};

/* ---- participating media — PROJECT-DEFINED: THERE IS NO REFERENCE BEHAVIOUR TO MATCH ----------------
 * The reference declares Medium {medium_type, density, color, anisotropy} (material.rs:8-34), carries one in every Material
 * (material.rs:75,107) and one in State (globals.rs:19,37), clamps the anisotropy in Material::finalize (material.rs:126) —
 * and tracer.rs never reads any of it ("Support of mediums / volumetric objects" is a Todo, Readme.md:13).  With
 * RPT_SCENE_MEDIA the fields mean what they mean in the renderer tracer.rs is a port of (its default build: homogeneous
 * media bounded by surfaces, no nesting, binary shadow rays), made consistent in two places (marked *): the medium acts on
 * a segment BEFORE what lies at its end is looked at, and a light sample taken inside a medium is attenuated by it.  The
 * arithmetic (f32, operation order) is the one written in oracle/rpt_oracle.hpp, Tracer::sample_pixel; exp and ln are
 * rpt_expf / rpt_logf of include/rpt_strict_math.h.
 *
 * A path carries `in_medium` (false at the camera) and a copy of the medium it is in (State.medium).  One iteration of
 * the bounce loop (tracer.rs:61-103) becomes:
 *   1. state.is_emitter = false (the reference never clears it, which is harmless only because an emitter hit ends the
 *      path there; here a path can go on after one).  closest_hit; a miss adds the background and ends the path as always
 *      (also inside a medium: a medium acts only on segments that end on something).
 *   2.* if in_medium, over seg = state.hit_dist:
 *        ABSORB    throughput.c *= exp(-(((1 - color.c) * seg) * density))            per channel (Beer-Lambert)
 *        EMISSIVE  radiance += ((color * seg) * density) * throughput
 *        SCATTER   one draw r;  d = min(-ln(r) / density, seg)  (f32::min);  if d < seg the path SCATTERS at p = ray.at(d):
 *                    throughput *= color;
 *                    next-event estimation from p exactly as direct_light (tracer.rs:126-170) with scatter_pos = p (no
 *                    offset) and the phase function in place of the BSDF: f = pdf = phase_hg(dot(-ray.direction,
 *                    light.direction), g), same MIS weight, same `pdf > 0` guard;
 *                    two draws r1, r2;  dir = sample_hg(-ray.direction, g, r1, r2);  scatter_sample.pdf =
 *                    phase_hg(dot(-ray.direction, dir), g);  scatter_sample.l = dir;  ray = Ray(p, dir)  (no eps offset);
 *                    then Russian roulette as after a surface bounce, and the next iteration.  The iteration counts
 *                    against max_depth like a surface bounce; in_medium stays as it is.
 *      g = the medium's anisotropy, clamped to [-0.9, 0.9] by Material::finalize (material.rs:126).
 *   3. (no scatter event) the reference's iteration as it is: State::finalize, emission, the emitter exit — its MIS weight
 *      reads scatter_sample.pdf, which after a medium scatter is the phase pdf —, direct_light, disney_sample, next ray.
 *      * In direct_light, when in_medium and light_sample.dist is finite, the unoccluded light's `li` is multiplied by the
 *      medium's transmittance over light_sample.dist: ABSORB exp(-(((1 - color.c) * dist) * density)) per channel,
 *      SCATTER exp(-(dist * density)), EMISSIVE 1.  (Shadow rays are the scene's binary any_hit, so this matters for lights
 *      INSIDE a medium — use RPT_SCENE_ANYHIT_USES_MAX_DIST there; a medium's own boundary occludes lights outside it.)
 *   4. after the next ray is set (tracer.rs:100-101), if state.material.medium.medium_type != NONE:
 *        in_medium = dot(ray.direction, state.normal) < 0      (the NEW direction against the geometric normal: entering)
 *        and when that is true state.medium = state.material.medium.
 *      Media are entered and left through their boundary surface (a refraction into it, spec_trans > 0, or any scatter
 *      that ends up on the inner side); there is no stack: media do not nest.
 *   phase_hg(c, g)  = INV_4_PI * (1 - g*g) / (d * sqrt(d)),  d = 1 + g*g + 2*g*c,  INV_4_PI = 1 / (4 * PI)  (f32)
 *   sample_hg(v, g, r1, r2):  cos = |g| < 0.001 ? 1 - 2*r2 : -(1 + g*g - q*q) / (2*g),  q = (1 - g*g) / (1 + g - 2*g*r2);
 *                             phi = r1 * TWO_PI;  sin = clamp(sqrt(1 - cos*cos), 0, 1);  (t, b) = onb(v)  (tracer.rs:184-189);
 *                             dir = (sin * cos(phi)) * t + (sin * sin(phi)) * b + cos * v
 * Draw order of an iteration inside a SCATTER medium: the distance draw; then, on a scatter event, light index, light r1,
 * r2 (if the scene has lights), r1, r2 of sample_hg, [roulette]; otherwise the surface bounce's draws as always.
 * Materials of small scenes are patches: RPT_MAT_MEDIUM writes all four medium fields.  rpt_upload_scene rejects a
 * negative or non-finite density.  Kernel forms: every scene class's kernel renders media (megakernel, the compacting
 * kernel, the SDF march kernel, large scenes' megakernel); RPT_RENDER_FAST_MATH and RPT_RENDER_NESTED_LOOPS do not
 * (RPT_ERR_UNSUPPORTED). */

typedef struct rpt_scene_desc {
    uint32_t abi_version;             /* RPT_ABI_VERSION */
    uint32_t flags;
    rpt_camera     camera;
    rpt_background background;
    float    eps;                     /* Tracer.eps = 0.005 (tracer.rs:16) */
    uint32_t max_depth;               /* Scene::recursion_depth() = 4 (scene.rs:28-30); at most 4096 */
    uint32_t n_spheres;   const rpt_sphere*   spheres;    /* tested first, in order */
    uint32_t n_planes;    const rpt_plane*    planes;     /* then planes, in order  */
    uint32_t n_lights;    const rpt_light*    lights;     /* then Scene::sample_lights */
    uint32_t n_materials; const rpt_material* materials;
    rpt_sdf  sdf;                     /* then the SDF object, if any */
} rpt_scene_desc;

/* Fill `out` with renderer/src/analytical.rs's AnalyticalScene (2 spheres, plane,
 * 1 spherical light, Pinhole defaults).  The arrays it points to are static.  */
int rpt_scene_analytical(rpt_scene_desc* out);

/* ---- render flags ----------------------------------------------------------
 * Bits 2-4, 6-7 and 9-10 named measured-slower kernel forms kept for A/B runs until ABI 3 (inline / three-room / pool / compacting SDF
 * marches, the resumable grid walk, the wavefront form of large scenes and its counterpart flag).  Those forms are gone from the
 * library (profiles/NOTES.md has their numbers, the history their code); the bits are reserved and rpt_render* answer them with
 * RPT_ERR_INVALID_ARG. */
enum {
    RPT_RENDER_DEFAULT      = 0u,
    /* Small scenes without an SDF object or media: the nested-loop kernel (sample loop outside, bounce loop inside) instead of
     * the path-regenerating one.  Same image bit for bit; the differential baseline of the reference's own scene class. */
    RPT_RENDER_NESTED_LOOPS = 1u << 0,
    /* Relaxed arithmetic: the same kernels built with hipcc's fast f32 divide/sqrt (~2.5 ulp) and FMA contraction.
     * Not bit-identical to the reference arithmetic (statistically equivalent: SURVEY.md 8c tier T1); off by default; bench.py
     * reports it beside the headline, never as the headline. */
    RPT_RENDER_FAST_MATH    = 1u << 1,
    /* Russian roulette (project-defined; the reference's bounce loop is a fixed `for _ in 0..depth` with three
     * early exits, tracer.rs:61-103).  After the throughput update and the next-ray set-up of bounce b (0-based),
     * when b + 1 >= 2 and b + 1 < depth:  q = clamp(max(throughput.x, throughput.y, throughput.z), 0.05, 1)
     * (f32::max, so a NaN component is ignored);  one more draw r (after the bounce's other draws);  r >= q ends
     * the path, otherwise throughput = throughput / q.  Same expectation, different samples: OFF (the default) is
     * the reference.  Worth it for deep paths (max_depth > 4). */
    RPT_RENDER_RUSSIAN_ROULETTE = 1u << 5,
    /* Small scenes without an SDF object: use the kernel that keeps the workgroup's 256 paths in LDS and re-deals them to its
     * threads before every stage.  It is what a launch of ONE sample per pixel takes by default — the reference's
     * own usage, one render() per redraw, where the megakernel has nothing to regenerate over and its waves drain (+4 % at 1080p,
     * more on small frames) — and slower than the megakernel from two samples per launch up; the flag forces it at any sample
     * count (tests).  Same image bit for bit. */
    RPT_RENDER_SMALL_COMPACT = 1u << 8,
    RPT_RENDER_ALL_FLAGS = RPT_RENDER_NESTED_LOOPS | RPT_RENDER_FAST_MATH | RPT_RENDER_RUSSIAN_ROULETTE | RPT_RENDER_SMALL_COMPACT
};

/* ---- context --------------------------------------------------------------- */
typedef struct rpt_ctx rpt_ctx;

/* Create a context on HIP device `device_id` (must be gfx950).  Replaces
 * Tracer::new (tracer.rs:13-19) together with rpt_upload_scene. */
int rpt_create(rpt_ctx** out, int device_id);
void rpt_destroy(rpt_ctx* ctx);
const char* rpt_last_error(const rpt_ctx* ctx);   /* valid until the next call on ctx; ctx may be NULL */
uint32_t rpt_abi_version(void);
/* sizeof(rpt_scene_desc) as this library was built: a binding in another language asserts it against its own
 * mirror of the struct before the first rpt_upload_scene (rust/gpu_tracer.rs does). */
uint32_t rpt_sizeof_scene_desc(void);
/* 0 for the shipped library (librpt_hip.so: exactly this header).  1 for the test build (librpt_hip_test.so): the same objects
 * linked with the hooks of include/rpt_test.h — per-function probes, grid-query probes, dispatch read-outs — which the parity
 * tests use to localise a mismatch. */
uint32_t rpt_build_has_test_hooks(void);

/* ---- the GPUs of one node (what replaces rayon's fan-out, tracer.rs:29-32) -----------------------------
 * The reference's only parallel construct is INSIDE render(): one rayon task per scanline.  Here the image is
 * row-tiled over the GPUs: rows are dealt cyclically in blocks of `tile_rows` rows (default 2; block b -> rank
 * b % world; sky rows are ~10x cheaper than floor rows, so contiguous slabs would not balance), each GPU keeps
 * its rows as a compact tile in its own HBM, ranks exchange nothing while rendering, and one RCCL gather over
 * xGMI brings the tiles to the root (rank 0), where a small kernel scatters them into the top-down image.
 * The RNG is keyed by the global pixel, so the image does not depend on the number of GPUs.
 *
 * Two ways to get there, same entry points afterwards:
 *   rpt_create_multi   ONE process drives n devices (one host thread, one stream + tile per device,
 *                      ncclCommInitAll): rpt_render / rpt_resident_* on such a context fan out over the
 *                      devices inside the call, exactly where the reference fans out over threads.
 *   rpt_create_rank    one process per GPU (torch.distributed.run, MPI, ...): rank 0 calls
 *                      rpt_comm_unique_id, the host distributes those 128 bytes over any channel it has, every
 *                      rank calls rpt_create_rank (ncclCommInitRank: collective).  rpt_resident_render,
 *                      rpt_resident_gather_device, rpt_resident_download[_u8] are then collective calls: every
 *                      rank makes them in the same order; destinations are only written on rank 0.
 * A context from rpt_create is world = 1.
 * rpt_create_multi accepts a device more than once: each entry is a rank of its own (own stream, own tile), and the launches of
 * one GPU's ranks run side by side, so that in a progressive render (rpt_resident_render called again and again) one rank's
 * launch fills the tail of the other's: +5 % on a resident 1920x1080 frame with the device listed twice, +22 % on 3840x270.
 * Such contexts gather their tiles with peer / device copies instead of RCCL (which needs one device per rank).             */
#define RPT_UNIQUE_ID_BYTES 128
typedef struct rpt_unique_id { char bytes[RPT_UNIQUE_ID_BYTES]; } rpt_unique_id;

int rpt_create_multi(rpt_ctx** out, const int* device_ids, int n_devices);
int rpt_comm_unique_id(rpt_unique_id* out);
int rpt_create_rank(rpt_ctx** out, int device_id, int rank, int world, const rpt_unique_id* id);
/* rank of this context's first device, number of ranks in all, number of devices this context drives */
int rpt_world(const rpt_ctx* ctx, int* rank, int* world, int* n_local);
/* Rows per cyclic block for the NEXT resident buffer / rpt_render call (0 < tile_rows). */
int rpt_set_tile_rows(rpt_ctx* ctx, uint32_t tile_rows);

/* How the context's launches are dispatched.  None of this changes a pixel: it decides when and where a sample is computed.
 * A workgroup of the render kernels computes a UNIT: one 16x16 tile x one chunk of the launch's samples.
 *   cost_order     1 (default): within a chunk the tiles are dispatched most expensive first, as timed by the context's previous
 *                  launch of the same shape, so that the cheap ones fill the launch's tail; 0: bottom rows first, always
 *   unit_rounds    a launch whose tiles are fewer than this many rounds of workgroups on the device (default 12) is cut into
 *                  chunks of samples until they are — a tile's chunks are handed from workgroup to workgroup through HBM, in
 *                  order —; 0: one unit per tile (and launches of more samples than the kernel's sample tables hold — 512, 192 for
 *                  scenes with the SDF object — are still one launch: chunks of that many)
 *   unit_min_spp   ... but no chunk shorter than this many samples (default 64)
 *   unit_slots     workgroups the device holds at once; 0 (default): 5 per compute unit
 * Environment defaults: RPT_DISPATCH_ORDER, RPT_UNIT_ROUNDS, RPT_UNIT_MIN_SPP.                                               */
int rpt_set_dispatch(rpt_ctx* ctx, uint32_t cost_order, uint32_t unit_rounds, uint32_t unit_min_spp, uint32_t unit_slots);

/* Copy the scene into the context (Tracer owns its scene: tracer.rs:8). */
int rpt_upload_scene(rpt_ctx* ctx, const rpt_scene_desc* scene);

/* Tracer::render (tracer.rs:22-123) on a HOST ColorBuffer.
 *   pixels      in/out, width*height*4 f32, RGBA, row 0 = top (buffer.rs:6-26)
 *   frames_done ColorBuffer.frames before the call; the caller adds `spp` afterwards
 *   spp         number of render() calls to fold into this one; spp = 1 is exactly
 *               one reference render(); spp = S is bit-identical to S calls
 *   seed        RNG seed (the reference's thread_rng, tracer.rs:44, is unseedable)
 * Blocks until `pixels` holds the result. */
int rpt_render(rpt_ctx* ctx, float* pixels, uint32_t width, uint32_t height,
               uint64_t frames_done, uint32_t spp, uint64_t seed, uint32_t flags);

/* ---- resident ColorBuffer (the reference's interactive loop without the PCIe round trip) --------------
 * renderer/src/main.rs:113-124 does, per redraw: pt.render(&mut buffer); buffer.convert_to_u8(frame).  With
 * rpt_render that moves 2 x 16 B per pixel over PCIe per frame.  Here the context owns a device ColorBuffer
 * (pixels + frames, buffer.rs:6-14): render accumulates into it, and the host fetches either the f32 pixels or
 * directly the gamma-encoded u8 frame (4 B per pixel).  Changing width/height or calling rpt_resident_reset
 * starts a new buffer (ColorBuffer::new, buffer.rs:18-26). */
int rpt_resident_render(rpt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp, uint64_t seed, uint32_t flags);
int rpt_resident_frames(const rpt_ctx* ctx, uint64_t* frames);          /* ColorBuffer.frames */
/* Device time of the last rpt_resident_render's launches (HIP events on the stream they ran on; the slowest of this
 * context's devices).  Waits for them. */
int rpt_resident_kernel_ms(rpt_ctx* ctx, float* ms);
int rpt_resident_download(rpt_ctx* ctx, float* pixels);                 /* width*height*4 f32 */
int rpt_resident_download_u8(rpt_ctx* ctx, uint8_t* frame);             /* convert_to_u8 on the device, width*height*4 bytes */
int rpt_resident_reset(rpt_ctx* ctx);
/* Page-lock a host buffer the caller will hand to rpt_resident_download[_u8] / rpt_render again and again (the redraw loop's
 * frame, renderer/src/main.rs:122): copies to and from page-locked memory are one DMA at the link's rate (8.3 MB of a 1080p u8
 * frame: ~0.2 ms), copies to pageable memory are staged by the runtime at ~7 GB/s (1.2 ms).  The buffer must stay allocated
 * until rpt_host_unpin; a buffer the caller has registered with HIP itself is as good.  Not needed for correctness. */
int rpt_host_pin(void* buffer, size_t bytes);
int rpt_host_unpin(void* buffer);
/* The resident image assembled ON THE ROOT DEVICE (no PCIe): RCCL gather of the tiles + scatter kernel, enqueued
 * behind the renders on the context's streams.  image_dev: width*height*4 f32 on rank 0's device (ignored on other
 * ranks; NULL = into the context's own staging image).  Returns without waiting; rpt_resident_sync waits.
 * world = 1: a device-to-device copy. */
int rpt_resident_gather_device(rpt_ctx* ctx, float* image_dev);
/* Block until everything enqueued on this context's devices has finished. */
int rpt_resident_sync(rpt_ctx* ctx);
/* Start the resident buffer from a host ColorBuffer (pixels + frames): what resuming a cloned ColorBuffer is in the
 * reference (buffer.rs:5 derives Clone).  Each rank uploads only the rows it owns. */
int rpt_resident_upload(rpt_ctx* ctx, const float* pixels, uint32_t width, uint32_t height, uint64_t frames);

/* Same on a DEVICE-resident buffer, asynchronously on `stream` (a hipStream_t; NULL is
 * HIP's null stream).  With world > 1 the image is row-tiled: rows are
 * dealt in blocks of `tile_rows` rows, block b to rank b % world, and `pixels` is
 * this rank's COMPACT tile buffer (rpt_tile_row_count(...) rows of `width` RGBA
 * pixels).  The RNG is keyed by the global pixel, so the image does not depend on
 * world.  world = 1, rank = 0 renders the whole image in place.
 * Calls on one context are ordered by the caller (one thread at a time).  The wavefront form of large scenes keeps its paths in
 * buffers of the context: a launch on another stream than the previous one waits (on the device) for that one to finish with
 * them; use one context per render that is to run concurrently. */
int rpt_render_device(rpt_ctx* ctx, float* pixels_dev, uint32_t width, uint32_t height,
                      uint64_t frames_done, uint32_t spp, uint64_t seed, uint32_t flags,
                      uint32_t tile_rows, uint32_t rank, uint32_t world, void* stream);

/* Number of image rows rank `rank` owns under the cyclic row-block tiling. */
uint32_t rpt_tile_row_count(uint32_t height, uint32_t tile_rows, uint32_t rank, uint32_t world);
/* Global image row of local row `local_row` of rank `rank`. */
uint32_t rpt_tile_global_row(uint32_t local_row, uint32_t tile_rows, uint32_t rank, uint32_t world);
/* Rows of the largest tile: what every rank's tile buffer is padded to, so that the gather moves equal counts. */
uint32_t rpt_tile_rows_padded(uint32_t height, uint32_t tile_rows, uint32_t world);
/* How a rank's rows move between a top-down host image and its compact tile (what rpt_render / rpt_resident_upload
 * do per device, each over its own PCIe link): `full_blocks` blocks of `block_rows` rows, block i at host row
 * host_row0 + i * host_row_stride and at tile row i * block_rows (ONE strided copy), plus `ragged_rows` rows of the
 * image's short last block from host row ragged_host_row0 to tile row ragged_tile_row0 when this rank owns it. */
typedef struct rpt_tile_plan {
    uint32_t full_blocks, block_rows, host_row0, host_row_stride;
    uint32_t ragged_rows, ragged_host_row0, ragged_tile_row0;
} rpt_tile_plan;
int rpt_tile_copy_plan(uint32_t height, uint32_t tile_rows, uint32_t rank, uint32_t world, rpt_tile_plan* out);

/* Scatter a rank-major concatenation of compact tiles (what an all-gather of the
 * per-rank tile buffers yields, each padded to `rows_padded` rows) into the full
 * top-down image, on the device. */
int rpt_untile_device(rpt_ctx* ctx, const float* gathered_dev, float* image_dev,
                      uint32_t width, uint32_t height, uint32_t tile_rows,
                      uint32_t world, uint32_t rows_padded, void* stream);

/* ColorBuffer::convert_to_u8 (buffer.rs:55-64): powf(0.4545)*255 saturating cast to
 * u8 for r,g,b; a*255 for alpha.  Device buffers; the step that follows render in the
 * reference's only caller (renderer/src/main.rs:118-122). */
int rpt_convert_to_u8_device(rpt_ctx* ctx, const float* pixels_dev, uint8_t* out_dev,
                             uint32_t width, uint32_t height, void* stream);

/* ColorBuffer::convert_to_u8_at (buffer.rs:67-89): blit the buffer into a larger u8 frame (frame_width x
 * frame_height, exactly that many RGBA bytes) at offset (at_x, at_y): no gamma, p*255 saturating cast, only
 * x in (at_x, at_x + width) and y in (at_y, at_y + height) with y = frame row + 1 — the reference's bounds. Pixels
 * outside keep their previous contents.  Device buffers. */
int rpt_convert_to_u8_at_device(rpt_ctx* ctx, const float* pixels_dev, uint32_t width, uint32_t height, uint8_t* frame_dev,
                                uint32_t at_x, uint32_t at_y, uint32_t frame_width, uint32_t frame_height, void* stream);

/* ---- denoiser ("Implement a denoiser" is a Todo of the reference, Readme.md:14) — PROJECT-DEFINED -----------
 * An edge-avoiding a-trous wavelet filter (Dammertz et al., HPG 2010) on the colour buffer alone, run in a compressed colour
 * space so that fireflies do not dominate: a separate pass over a ColorBuffer, never part of render().  There is no reference
 * behaviour to match; the arithmetic (f32, this operation order) is the one in oracle/rpt_oracle.hpp, denoise():
 *   c' = c / (1 + c)                                       per r, g, b of every pixel (alpha is not filtered)
 *   iteration i = 0 .. iterations-1, step s = 2^i, k_i = edge_k * 4^i; for every pixel p, over the 3x3 taps
 *   q = p + s * (dx, dy), dy = -1..1 outer, dx = -1..1 inner, q inside the image:
 *       d = c'_p - c'_q;  d2 = fma(d.r, d.r, fma(d.g, d.g, d.b*d.b));  a tap whose d2 is NaN is skipped;
 *       t = fma(-d2, k_i, 1);  g = t > 0 ? t : 0;  wt = (H[dy+1] * H[dx+1]) * (g * g),  H = {0.25, 0.5, 0.25};
 *       acc = fma(c'_q, wt, acc);  wsum += wt          (fma: ONE rounding, written out on both sides since round 4)
 *   c'_p <- wsum > 0 ? acc / wsum : c'_p                   (all pixels at once: out of place)
 *   finally c = c' / (1 - c'), alpha = the input's; a pixel whose input r, g or b is not finite is copied through unchanged
 *   (and, being NaN in the compressed space, takes no part in its neighbours' sums).
 * The filter is a weighted mean of compressed colours: it darkens a noisy region slightly (1-4 % at 1-4 spp on the
 * reference's scene; tests/test_denoise.py) — the price of taming fireflies without auxiliary buffers.
 * iterations 1..6 (footprint 2^(iterations+1) - 1 pixels), edge_k > 0 (larger = sharper edges, less smoothing; 2 is a good
 * default at 1-16 spp).  Per iteration the pass reads and writes the buffer once: 32 B per pixel of HBM traffic.
 * pixels_dev and out_dev are width*height*4 f32 device buffers and must not overlap; the context keeps one more buffer of
 * that size between calls. */
int rpt_denoise_device(rpt_ctx* ctx, const float* pixels_dev, float* out_dev, uint32_t width, uint32_t height,
                       uint32_t iterations, float edge_k, void* stream);
/* The same on HOST buffers (upload, filter, download; blocks). */
int rpt_denoise(rpt_ctx* ctx, const float* pixels, float* out, uint32_t width, uint32_t height, uint32_t iterations, float edge_k);

/* The same on HOST buffers (upload, convert, download; blocks): what ColorBuffer::convert_to_u8
 * does for a caller that owns a host ColorBuffer (buffer.rs:55-64, frame = width*height*4 bytes). */
int rpt_convert_to_u8(rpt_ctx* ctx, const float* pixels, uint8_t* frame, uint32_t width, uint32_t height);

int rpt_synchronize(rpt_ctx* ctx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RPT_H */
