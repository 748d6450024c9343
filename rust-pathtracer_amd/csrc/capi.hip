// capi.hip — the C ABI of include/rpt.h: contexts, scene upload, launches.  Host code only; the kernels
// and their launch wrappers are in kernels.hip (launch.h).
//
// There is NO CPU fallback: without a gfx950 device every entry point that computes returns
// RPT_ERR_NO_DEVICE / RPT_ERR_HIP.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rpt.h"
#include "host_scene.h"
#include "launch.h"

using namespace rptdev;
using rpthost::HostGrid;
using rpthost::build_grid;
using rpthost::make_camera;

struct rpt_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool has_scene = false;
    bool large = false;               // scene exceeds the kernarg tables: SceneLarge + device tables
    SceneSmallSdf scene;              // camera part is filled per launch (depends on width/height); sdf.n_prims == 0: plain
    SceneLarge scene_large;
    void* tables = nullptr;           // one device allocation holding the large scene's tables
    rpt_camera camera;
    float* fb = nullptr;              // device framebuffer for the host-pointer API
    size_t fb_bytes = 0;
    float* res = nullptr;             // resident ColorBuffer: pixels (f32 RGBA) followed by the u8 frame
    uint32_t res_w = 0, res_h = 0;
    uint64_t res_frames = 0;
    std::string err;
};

static thread_local std::string g_err;

static void set_err(rpt_ctx* ctx, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_err = buf;
}

#define RPT_HIP_CHECK(ctx, call)                                                                  \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            set_err(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RPT_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

// Every entry point runs on its context's device and puts the caller's current device back afterwards
// (the caller may be a torch process with its own idea of the current device).
struct DeviceGuard {
    int prev = -1;
    hipError_t status;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        status = (prev == device) ? hipSuccess : hipSetDevice(device);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define RPT_ON_DEVICE(ctx)                 \
    DeviceGuard guard_((ctx)->device);     \
    RPT_HIP_CHECK(ctx, guard_.status)

// Lanes that must be parked on a surface hit before a wave runs its shading block (1..64).
// RPT_SHADE_THRESHOLD overrides the default for tuning runs.
static uint32_t env_lanes(const char* name, long dflt)
{
    const char* e = getenv(name);
    long t = e ? strtol(e, nullptr, 10) : dflt;
    return (uint32_t)(t < 1 ? 1 : (t > 64 ? 64 : t));
}
// Scheduling knobs (they change when work runs, never its result): lanes that must be waiting before a wave
// runs a block.
static uint32_t shade_threshold() { static const uint32_t v = env_lanes("RPT_SHADE_THRESHOLD", 56); return v; }
static uint32_t sdf_march_min_lanes() { static const uint32_t v = env_lanes("RPT_SDF_MARCH_MIN_LANES", 8); return v; }
static uint32_t sdf_pool_shade_lanes() { static const uint32_t v = env_lanes("RPT_SDF_POOL_SHADE_LANES", 48); return v; }
static uint32_t sdf_pool_resolve_lanes() { static const uint32_t v = env_lanes("RPT_SDF_POOL_RESOLVE_LANES", 16); return v; }
static uint32_t sdf_pool_min_batch() { static const uint32_t v = env_lanes("RPT_SDF_POOL_MIN_BATCH", 32); return v; }
static uint32_t sdf_pool_patience() { static const uint32_t v = getenv("RPT_SDF_POOL_PATIENCE") ? (uint32_t)atoi(getenv("RPT_SDF_POOL_PATIENCE")) : 8u; return v; }
static uint32_t grid_walk_min_lanes() { static const uint32_t v = env_lanes("RPT_GRID_WALK_MIN_LANES", 8); return v; }

extern "C" {

uint32_t rpt_abi_version(void) { return RPT_ABI_VERSION; }

const char* rpt_last_error(const rpt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

// renderer/src/analytical.rs as data (see include/rpt.h)
int rpt_scene_analytical(rpt_scene_desc* out)
{
    if (!out) return RPT_ERR_INVALID_ARG;
    static rpt_sphere spheres[2];
    static rpt_plane planes[1];
    static rpt_light lights[1];
    static rpt_material mats[3];
    memset(spheres, 0, sizeof(spheres));
    memset(planes, 0, sizeof(planes));
    memset(lights, 0, sizeof(lights));
    memset(mats, 0, sizeof(mats));
    memset(out, 0, sizeof(*out));

    spheres[0] = rpt_sphere{{-1.1f, 0.0f, 0.0f}, 1.0f, 0};           // analytical.rs:41
    spheres[1] = rpt_sphere{{1.1f, 0.0f, 0.0f}, 1.0f, 1};            // analytical.rs:70
    planes[0] = rpt_plane{{0.0f, 1.0f, 0.0f}, {0.0f, -1.0f, 0.0f}, 0.0001f, 2, 0.0f};   // analytical.rs:194-198

    mats[0].mask = RPT_MAT_RGB | RPT_MAT_ROUGHNESS | RPT_MAT_METALLIC;              // analytical.rs:56-58
    mats[0].rgb[0] = mats[0].rgb[1] = mats[0].rgb[2] = 1.0f;
    mats[0].roughness = 0.05f;
    mats[0].metallic = 1.0f;
    mats[1].mask = RPT_MAT_RGB | RPT_MAT_CLEARCOAT | RPT_MAT_CLEARCOAT_GLOSS | RPT_MAT_ROUGHNESS;   // analytical.rs:82-85
    mats[1].rgb[0] = 1.0f; mats[1].rgb[1] = 0.186f; mats[1].rgb[2] = 0.0f;
    mats[1].clearcoat = 1.0f;
    mats[1].clearcoat_gloss = 1.0f;
    mats[1].roughness = 0.1f;
    mats[2].mask = RPT_MAT_ROUGHNESS;                                               // analytical.rs:107-116
    mats[2].roughness = 1.0f;
    mats[2].proc_kind = RPT_PROC_CHECKER_DIR;
    mats[2].proc_params[0] = 0.5f; mats[2].proc_params[1] = 100.0f;
    mats[2].proc_params[2] = 0.25f; mats[2].proc_params[3] = 0.1f;

    lights[0].type = RPT_LIGHT_SPHERICAL;                                           // analytical.rs:15-16
    lights[0].position[0] = 3.0f; lights[0].position[1] = 2.0f; lights[0].position[2] = 2.0f;
    lights[0].emission[0] = lights[0].emission[1] = lights[0].emission[2] = 3.0f;
    lights[0].radius = 1.0f;
    lights[0].area = 4.0f * 3.14159265358979323846f * lights[0].radius * lights[0].radius;   // light.rs:22

    out->abi_version = RPT_ABI_VERSION;
    out->flags = 0;
    out->camera.origin[2] = 3.0f;                                                   // pinhole.rs:16
    out->camera.fov_deg = 80.0f;                                                    // pinhole.rs:23
    out->background.kind = RPT_BG_GRADIENT_Y;                                       // analytical.rs:28-32
    out->background.colour_a[0] = out->background.colour_a[1] = out->background.colour_a[2] = 1.0f;
    out->background.colour_b[0] = 0.5f; out->background.colour_b[1] = 0.7f; out->background.colour_b[2] = 1.0f;
    out->background.gamma = 2.2f;
    out->background.scale = 0.5f;
    out->eps = 0.005f;                                                              // tracer.rs:16
    out->max_depth = 4;                                                             // scene.rs:29
    out->n_spheres = 2; out->spheres = spheres;
    out->n_planes = 1; out->planes = planes;
    out->n_lights = 1; out->lights = lights;
    out->n_materials = 3; out->materials = mats;
    return RPT_OK;
}

int rpt_create(rpt_ctx** out, int device_id)
{
    if (!out) { set_err(nullptr, "rpt_create: out is NULL"); return RPT_ERR_INVALID_ARG; }
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_err(nullptr, "rpt_create: no HIP device (%s); this library has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return RPT_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= count) { set_err(nullptr, "rpt_create: device %d out of range [0,%d)", device_id, count); return RPT_ERR_INVALID_ARG; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { set_err(nullptr, "rpt_create: hipGetDeviceProperties failed"); return RPT_ERR_HIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(nullptr, "rpt_create: device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
        return RPT_ERR_NO_DEVICE;
    }
    rpt_ctx* ctx = new (std::nothrow) rpt_ctx();
    if (!ctx) return RPT_ERR_HIP;
    ctx->device = device_id;
    {
        DeviceGuard guard(device_id);
        if (guard.status != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            set_err(nullptr, "rpt_create: cannot create a stream on device %d", device_id);
            delete ctx;
            return RPT_ERR_HIP;
        }
    }
    *out = ctx;
    return RPT_OK;
}

void rpt_destroy(rpt_ctx* ctx)
{
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    if (ctx->fb) (void)hipFree(ctx->fb);
    if (ctx->res) (void)hipFree(ctx->res);
    if (ctx->tables) (void)hipFree(ctx->tables);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rpt_upload_scene(rpt_ctx* ctx, const rpt_scene_desc* s)
{
    if (!ctx || !s) { set_err(ctx, "rpt_upload_scene: NULL argument"); return RPT_ERR_INVALID_ARG; }
    if (s->abi_version != RPT_ABI_VERSION) { set_err(ctx, "rpt_upload_scene: abi_version %u != %u", s->abi_version, RPT_ABI_VERSION); return RPT_ERR_INVALID_ARG; }
    if ((s->n_spheres && !s->spheres) || (s->n_planes && !s->planes) || (s->n_lights && !s->lights) || (s->n_materials && !s->materials)) {
        set_err(ctx, "rpt_upload_scene: a table pointer is NULL");
        return RPT_ERR_INVALID_ARG;
    }
    // bounded loop counts: a wave must always reach the end of its kernel
    if (s->max_depth > 4096u) { set_err(ctx, "rpt_upload_scene: max_depth %u exceeds the supported 4096", s->max_depth); return RPT_ERR_INVALID_ARG; }
    if (s->sdf.n_prims && s->sdf.max_steps > 65536u) { set_err(ctx, "rpt_upload_scene: sdf.max_steps %u exceeds the supported 65536", s->sdf.max_steps); return RPT_ERR_INVALID_ARG; }
    const bool large = s->n_spheres > (uint32_t)kMaxSpheres || s->n_lights > (uint32_t)kMaxLights || s->n_materials > (uint32_t)kMaxMaterials;
    if (s->n_planes > (uint32_t)kMaxPlanes) {
        set_err(ctx, "rpt_upload_scene: at most %d planes are supported", kMaxPlanes);
        return RPT_ERR_UNSUPPORTED;
    }
    for (uint32_t i = 0; i < s->n_spheres; ++i)
        if (s->spheres[i].material >= s->n_materials) { set_err(ctx, "rpt_upload_scene: sphere %u material out of range", i); return RPT_ERR_INVALID_ARG; }
    for (uint32_t i = 0; i < s->n_planes; ++i)
        if (s->planes[i].material >= s->n_materials) { set_err(ctx, "rpt_upload_scene: plane %u material out of range", i); return RPT_ERR_INVALID_ARG; }

    auto dev_plane = [](const rpt_plane& a) { return DevPlane{a.normal[0], a.normal[1], a.normal[2], a.point[0], a.point[1], a.point[2], a.min_denom, a.material, a.max_t}; };
    auto dev_light = [](const rpt_light& a) { return DevLight{a.type, a.position[0], a.position[1], a.position[2], a.emission[0], a.emission[1], a.emission[2], a.radius, a.area}; };
    auto dev_material = [](const rpt_material& a) {
        DevMaterial m;
        m.mask = a.mask; m.proc_kind = a.proc_kind;
        for (int k = 0; k < 3; ++k) { m.rgb[k] = a.rgb[k]; m.emission[k] = a.emission[k]; }
        m.anisotropic = a.anisotropic; m.metallic = a.metallic; m.roughness = a.roughness; m.subsurface = a.subsurface;
        m.specular_tint = a.specular_tint; m.sheen = a.sheen; m.sheen_tint = a.sheen_tint; m.clearcoat = a.clearcoat;
        m.clearcoat_gloss = a.clearcoat_gloss; m.spec_trans = a.spec_trans; m.ior = a.ior;
        for (int k = 0; k < 4; ++k) m.proc_params[k] = a.proc_params[k];
        return m;
    };
    auto dev_background = [](const rpt_background& b) {
        return DevBackground{b.kind, b.colour_a[0], b.colour_a[1], b.colour_a[2], b.colour_b[0], b.colour_b[1], b.colour_b[2], b.gamma, b.scale};
    };

    if (s->sdf.n_prims) {
        if (s->sdf.n_prims > (uint32_t)kMaxSdfPrims || !s->sdf.prims || s->sdf.material >= s->n_materials || !(s->sdf.smooth_k > 0.0f)) {
            set_err(ctx, "rpt_upload_scene: bad SDF object (1..%d prims, material in range, smooth_k > 0)", kMaxSdfPrims);
            return RPT_ERR_INVALID_ARG;
        }
        for (uint32_t i = 0; i < s->sdf.n_prims; ++i)
            if (s->sdf.prims[i].kind > RPT_SDF_TORUS_Y) { set_err(ctx, "rpt_upload_scene: unknown SDF primitive kind"); return RPT_ERR_INVALID_ARG; }
        if (large) { set_err(ctx, "rpt_upload_scene: the SDF object is only supported in small scenes"); return RPT_ERR_UNSUPPORTED; }
    }
    if (large) {
        if (s->n_spheres >= kNoSphere) { set_err(ctx, "rpt_upload_scene: at most 2^28 - 2 spheres"); return RPT_ERR_UNSUPPORTED; }
        // Layered patches need a bit per primitive; large scenes must use full sphere materials.
        for (uint32_t i = 0; i < s->n_spheres; ++i) {
            const rpt_material& m = s->materials[s->spheres[i].material];
            if ((m.mask & RPT_MAT_ALL) != RPT_MAT_ALL || m.proc_kind != RPT_PROC_NONE) {
                set_err(ctx, "rpt_upload_scene: scenes beyond %d spheres / %d lights / %d materials need full sphere materials "
                             "(mask == RPT_MAT_ALL, no procedural part); sphere %u does not", kMaxSpheres, kMaxLights, kMaxMaterials, i);
                return RPT_ERR_UNSUPPORTED;
            }
        }
        RPT_ON_DEVICE(ctx);
        const size_t sz_sph = sizeof(float4) * s->n_spheres;
        const size_t sz_smat = (sizeof(uint32_t) * s->n_spheres + 15) & ~(size_t)15;
        const size_t sz_lights = (sizeof(DevLight) * (s->n_lights ? s->n_lights : 1) + 15) & ~(size_t)15;
        const size_t sz_mats = sizeof(DevMaterial) * (s->n_materials ? s->n_materials : 1);
        const bool use_grid = s->n_spheres >= 64 && !getenv("RPT_NO_GRID");
        HostGrid grid;
        if (use_grid) { const char* e = getenv("RPT_GRID_SPHERES_PER_CELL"); grid = build_grid(s->spheres, s->n_spheres, e ? atof(e) : 1.0); }
        const size_t sz_tables = (sz_sph + sz_smat + sz_lights + sz_mats + 15) & ~(size_t)15;
        const size_t sz_cstart = (sizeof(uint32_t) * grid.cell_start.size() + 15) & ~(size_t)15;
        const size_t sz_items = (sizeof(uint32_t) * grid.items.size() + 15) & ~(size_t)15;
        const size_t sz_cell_sph = sizeof(float4) * grid.items.size();
        std::vector<unsigned char> host(sz_tables + sz_cstart + sz_items + sz_cell_sph, 0);
        float4* h_sph = reinterpret_cast<float4*>(host.data());
        uint32_t* h_smat = reinterpret_cast<uint32_t*>(host.data() + sz_sph);
        DevLight* h_lights = reinterpret_cast<DevLight*>(host.data() + sz_sph + sz_smat);
        DevMaterial* h_mats = reinterpret_cast<DevMaterial*>(host.data() + sz_sph + sz_smat + sz_lights);
        for (uint32_t i = 0; i < s->n_spheres; ++i) {
            h_sph[i] = make_float4(s->spheres[i].center[0], s->spheres[i].center[1], s->spheres[i].center[2], s->spheres[i].radius);
            h_smat[i] = s->spheres[i].material;
        }
        for (uint32_t i = 0; i < s->n_lights; ++i) h_lights[i] = dev_light(s->lights[i]);
        for (uint32_t i = 0; i < s->n_materials; ++i) h_mats[i] = dev_material(s->materials[i]);
        if (use_grid) {
            memcpy(host.data() + sz_tables, grid.cell_start.data(), sizeof(uint32_t) * grid.cell_start.size());
            memcpy(host.data() + sz_tables + sz_cstart, grid.items.data(), sizeof(uint32_t) * grid.items.size());
            // the spheres again, in cell-list order: the walk reads a cell's spheres without going through the index
            float4* h_cell_sph = reinterpret_cast<float4*>(host.data() + sz_tables + sz_cstart + sz_items);
            for (size_t k = 0; k < grid.items.size(); ++k) h_cell_sph[k] = h_sph[grid.items[k]];
        }
        if (ctx->tables) { RPT_HIP_CHECK(ctx, hipFree(ctx->tables)); ctx->tables = nullptr; }
        RPT_HIP_CHECK(ctx, hipMalloc(&ctx->tables, host.size()));
        RPT_HIP_CHECK(ctx, hipMemcpy(ctx->tables, host.data(), host.size(), hipMemcpyHostToDevice));
        unsigned char* base = reinterpret_cast<unsigned char*>(ctx->tables);
        SceneLarge& L = ctx->scene_large;
        memset(&L, 0, sizeof(L));
        L.n_spheres = s->n_spheres; L.n_planes = s->n_planes; L.n_lights = s->n_lights; L.n_materials = s->n_materials;
        L.flags = s->flags; L.max_depth = s->max_depth; L.eps = s->eps; L.n_lights_f = (float)s->n_lights;
        L.bg = dev_background(s->background);
        L.spheres = reinterpret_cast<const float4*>(base);
        L.sphere_material = reinterpret_cast<const uint32_t*>(base + sz_sph);
        L.lights = reinterpret_cast<const DevLight*>(base + sz_sph + sz_smat);
        L.materials = reinterpret_cast<const DevMaterial*>(base + sz_sph + sz_smat + sz_lights);
        for (uint32_t i = 0; i < s->n_planes; ++i) L.planes[i] = dev_plane(s->planes[i]);
        L.use_grid = use_grid ? 1u : 0u;
        if (use_grid) {
            for (int a = 0; a < 3; ++a) {
                L.gn[a] = grid.n[a]; L.gmin[a] = grid.gmin[a]; L.gmax[a] = grid.gmax[a];
                L.cell_size[a] = grid.cs[a]; L.inv_cell_size[a] = grid.inv_cs[a];
                L.gcenter[a] = grid.center[a];
            }
            L.safe_r2 = grid.safe_r2;
            L.cell_start = reinterpret_cast<const uint32_t*>(base + sz_tables);
            L.cell_items = reinterpret_cast<const uint32_t*>(base + sz_tables + sz_cstart);
            L.cell_spheres = reinterpret_cast<const float4*>(base + sz_tables + sz_cstart + sz_items);
        }
        ctx->camera = s->camera;
        ctx->large = true;
        ctx->has_scene = true;
        return RPT_OK;
    }

    SceneSmallSdf& d = ctx->scene;
    memset(&d, 0, sizeof(d));
    d.n_spheres = s->n_spheres; d.n_planes = s->n_planes; d.n_lights = s->n_lights; d.n_materials = s->n_materials;
    d.flags = s->flags;
    d.max_depth = s->max_depth;
    d.eps = s->eps;
    d.n_lights_f = (float)s->n_lights;
    d.bg.kind = s->background.kind;
    d.bg.ax = s->background.colour_a[0]; d.bg.ay = s->background.colour_a[1]; d.bg.az = s->background.colour_a[2];
    d.bg.bx = s->background.colour_b[0]; d.bg.by = s->background.colour_b[1]; d.bg.bz = s->background.colour_b[2];
    d.bg.gamma = s->background.gamma;
    d.bg.scale = s->background.scale;
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const rpt_sphere& a = s->spheres[i];
        d.spheres[i] = DevSphere{a.center[0], a.center[1], a.center[2], a.radius, a.material};
    }
    for (uint32_t i = 0; i < s->n_planes; ++i) {
        const rpt_plane& a = s->planes[i];
        d.planes[i] = DevPlane{a.normal[0], a.normal[1], a.normal[2], a.point[0], a.point[1], a.point[2], a.min_denom, a.material, a.max_t};
    }
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        const rpt_light& a = s->lights[i];
        d.lights[i] = DevLight{a.type, a.position[0], a.position[1], a.position[2], a.emission[0], a.emission[1], a.emission[2], a.radius, a.area};
    }
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const rpt_material& a = s->materials[i];
        DevMaterial& m = d.materials[i];
        m.mask = a.mask; m.proc_kind = a.proc_kind;
        for (int k = 0; k < 3; ++k) { m.rgb[k] = a.rgb[k]; m.emission[k] = a.emission[k]; }
        m.anisotropic = a.anisotropic; m.metallic = a.metallic; m.roughness = a.roughness; m.subsurface = a.subsurface;
        m.specular_tint = a.specular_tint; m.sheen = a.sheen; m.sheen_tint = a.sheen_tint; m.clearcoat = a.clearcoat;
        m.clearcoat_gloss = a.clearcoat_gloss; m.spec_trans = a.spec_trans; m.ior = a.ior;
        for (int k = 0; k < 4; ++k) m.proc_params[k] = a.proc_params[k];
    }
    d.sdf.n_prims = s->sdf.n_prims; d.sdf.max_steps = s->sdf.max_steps; d.sdf.material = s->sdf.material;
    d.sdf.smooth_k = s->sdf.smooth_k; d.sdf.hit_eps = s->sdf.hit_eps; d.sdf.max_t = s->sdf.max_t; d.sdf.normal_eps = s->sdf.normal_eps;
    d.sdf.inv_smooth_k = s->sdf.n_prims ? 1.0f / s->sdf.smooth_k : 0.0f;
    for (uint32_t i = 0; i < s->sdf.n_prims; ++i) {
        const rpt_sdf_prim& a = s->sdf.prims[i];
        d.sdf.prims[i] = DevSdfPrim{a.kind, a.center[0], a.center[1], a.center[2], a.params[0], a.params[1]};
    }
    ctx->camera = s->camera;
    ctx->large = false;
    ctx->has_scene = true;
    return RPT_OK;
}

uint32_t rpt_tile_row_count(uint32_t height, uint32_t tile_rows, uint32_t rank, uint32_t world)
{
    return tile_row_count(height, tile_rows, rank, world);
}

uint32_t rpt_tile_global_row(uint32_t local_row, uint32_t tile_rows, uint32_t rank, uint32_t world)
{
    if (tile_rows == 0 || world == 0) return 0;
    return tile_global_row(local_row, tile_rows, rank, world);
}

int rpt_render_device(rpt_ctx* ctx, float* pixels_dev, uint32_t width, uint32_t height, uint64_t frames_done, uint32_t spp,
                      uint64_t seed, uint32_t flags, uint32_t tile_rows, uint32_t rank, uint32_t world, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_render_device: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_scene) { set_err(ctx, "rpt_render_device: no scene uploaded"); return RPT_ERR_NO_SCENE; }
    if (!pixels_dev || width == 0 || height == 0 || world == 0 || rank >= world || tile_rows == 0) {
        set_err(ctx, "rpt_render_device: invalid argument (pixels=%p width=%u height=%u tile_rows=%u rank=%u world=%u)",
                (void*)pixels_dev, width, height, tile_rows, rank, world);
        return RPT_ERR_INVALID_ARG;
    }
    if ((uint64_t)width * height > 0xFFFFFFFFull) { set_err(ctx, "rpt_render_device: image too large for 32-bit pixel indices"); return RPT_ERR_INVALID_ARG; }
    if (((uintptr_t)pixels_dev & 15u) != 0) { set_err(ctx, "rpt_render_device: pixels must be 16-byte aligned"); return RPT_ERR_INVALID_ARG; }
    if (spp == 0) return RPT_OK;
    if (world == 1) tile_rows = height;                              // one block: local row == global row

    RPT_ON_DEVICE(ctx);
    SceneSmallSdf scs = ctx->scene;
    SceneLarge scl = ctx->scene_large;
    scs.cam = scl.cam = make_camera(ctx->camera, (float)width, (float)height);

    RenderParams rp;
    rp.pixels = pixels_dev;
    rp.width = width; rp.height = height;
    rp.rows_local = tile_row_count(height, tile_rows, rank, world);
    rp.tile_rows = tile_rows; rp.rank = rank; rp.world = world;
    rp.seed = seed;
    rp.tiles_x = (width + 15u) / 16u;
    rp.sdf_resumable_march = (flags & RPT_RENDER_SDF_INLINE_MARCH) ? 0u : ((flags & RPT_RENDER_SDF_POOL_MARCH) ? 2u : 1u);
    rp.pool_shade_lanes = sdf_pool_shade_lanes();
    rp.pool_resolve_lanes = sdf_pool_resolve_lanes();
    rp.pool_min_batch = sdf_pool_min_batch();
    rp.pool_patience = sdf_pool_patience();
    rp.shade_threshold = shade_threshold();
    rp.march_min_lanes = sdf_march_min_lanes();
    rp.walk_min_lanes = grid_walk_min_lanes();
    rp.grid_resumable_walk = (flags & RPT_RENDER_GRID_RESUMABLE_WALK) ? 1u : 0u;
    if (rp.rows_local == 0) return RPT_OK;
    const uint32_t tiles_y = (rp.rows_local + 15u) / 16u;
    const uint64_t nblocks = (uint64_t)rp.tiles_x * tiles_y;
    if (nblocks > 0x7FFFFFFFull) { set_err(ctx, "rpt_render_device: grid too large"); return RPT_ERR_INVALID_ARG; }

    // The LDS tables of the regenerating kernel hold a bounded number of samples: larger batches are
    // split into consecutive launches (the running mean carries over in the framebuffer).
    const uint32_t max_chunk = rptlaunch::max_spp_per_launch();
    for (uint32_t done = 0; done < spp;) {
        const uint32_t chunk = (spp - done > max_chunk) ? max_chunk : (spp - done);
        rp.spp = chunk;
        rp.frames_done = frames_done + done;
        const bool nested = (flags & RPT_RENDER_NESTED_LOOPS) != 0;
        if (flags & RPT_RENDER_FAST_MATH) RPT_HIP_CHECK(ctx, rptlaunch_fast::render(scs, scl, ctx->large, nested, rp, (uint32_t)nblocks, (hipStream_t)stream));
        else RPT_HIP_CHECK(ctx, rptlaunch::render(scs, scl, ctx->large, nested, rp, (uint32_t)nblocks, (hipStream_t)stream));
        done += chunk;
    }
    return RPT_OK;
}

int rpt_render(rpt_ctx* ctx, float* pixels, uint32_t width, uint32_t height, uint64_t frames_done, uint32_t spp, uint64_t seed,
               uint32_t flags)
{
    if (!ctx) { set_err(nullptr, "rpt_render: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels || width == 0 || height == 0) { set_err(ctx, "rpt_render: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_scene) { set_err(ctx, "rpt_render: no scene uploaded"); return RPT_ERR_NO_SCENE; }
    RPT_ON_DEVICE(ctx);
    const size_t bytes = (size_t)width * height * 4 * sizeof(float);
    if (bytes > ctx->fb_bytes) {
        if (ctx->fb) { RPT_HIP_CHECK(ctx, hipFree(ctx->fb)); ctx->fb = nullptr; ctx->fb_bytes = 0; }
        RPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->fb, bytes));
        ctx->fb_bytes = bytes;
    }
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->fb, pixels, bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = rpt_render_device(ctx, ctx->fb, width, height, frames_done, spp, seed, flags, height, 0, 1, ctx->stream);
    if (rc != RPT_OK) return rc;
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(pixels, ctx->fb, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return RPT_OK;
}

int rpt_resident_reset(rpt_ctx* ctx)
{
    if (!ctx) { set_err(nullptr, "rpt_resident_reset: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    if (ctx->res) { RPT_HIP_CHECK(ctx, hipFree(ctx->res)); ctx->res = nullptr; }
    ctx->res_w = ctx->res_h = 0;
    ctx->res_frames = 0;
    return RPT_OK;
}

int rpt_resident_render(rpt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp, uint64_t seed, uint32_t flags)
{
    if (!ctx) { set_err(nullptr, "rpt_resident_render: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (width == 0 || height == 0) { set_err(ctx, "rpt_resident_render: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_scene) { set_err(ctx, "rpt_resident_render: no scene uploaded"); return RPT_ERR_NO_SCENE; }
    RPT_ON_DEVICE(ctx);
    if (!ctx->res || ctx->res_w != width || ctx->res_h != height) {             // ColorBuffer::new(width, height)
        int rc = rpt_resident_reset(ctx);
        if (rc != RPT_OK) return rc;
        const size_t n = (size_t)width * height;
        RPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->res, n * 16 + n * 4));
        RPT_HIP_CHECK(ctx, hipMemsetAsync(ctx->res, 0, n * 16, ctx->stream));
        ctx->res_w = width; ctx->res_h = height;
    }
    int rc = rpt_render_device(ctx, ctx->res, width, height, ctx->res_frames, spp, seed, flags, height, 0, 1, ctx->stream);
    if (rc != RPT_OK) return rc;
    ctx->res_frames += spp;                                                      // tracer.rs:121
    return RPT_OK;
}

int rpt_resident_frames(const rpt_ctx* ctx, uint64_t* frames)
{
    if (!ctx || !frames) return RPT_ERR_INVALID_ARG;
    *frames = ctx->res_frames;
    return RPT_OK;
}

int rpt_resident_download(rpt_ctx* ctx, float* pixels)
{
    if (!ctx) { set_err(nullptr, "rpt_resident_download: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels || !ctx->res) { set_err(ctx, "rpt_resident_download: no resident buffer or NULL destination"); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    const size_t n = (size_t)ctx->res_w * ctx->res_h;
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(pixels, ctx->res, n * 16, hipMemcpyDeviceToHost, ctx->stream));
    RPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return RPT_OK;
}

int rpt_resident_download_u8(rpt_ctx* ctx, uint8_t* frame)
{
    if (!ctx) { set_err(nullptr, "rpt_resident_download_u8: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!frame || !ctx->res) { set_err(ctx, "rpt_resident_download_u8: no resident buffer or NULL destination"); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    const size_t n = (size_t)ctx->res_w * ctx->res_h;
    uint8_t* out_dev = reinterpret_cast<uint8_t*>(ctx->res) + n * 16;
    int rc = rpt_convert_to_u8_device(ctx, ctx->res, out_dev, ctx->res_w, ctx->res_h, ctx->stream);
    if (rc != RPT_OK) return rc;
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(frame, out_dev, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return RPT_OK;
}

int rpt_untile_device(rpt_ctx* ctx, const float* gathered_dev, float* image_dev, uint32_t width, uint32_t height,
                      uint32_t tile_rows, uint32_t world, uint32_t rows_padded, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_untile_device: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!gathered_dev || !image_dev || width == 0 || height == 0 || tile_rows == 0 || world == 0) { set_err(ctx, "rpt_untile_device: invalid argument"); return RPT_ERR_INVALID_ARG; }
    for (uint32_t r = 0; r < world; ++r)
        if (tile_row_count(height, tile_rows, r, world) > rows_padded) { set_err(ctx, "rpt_untile_device: rows_padded %u too small", rows_padded); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    RPT_HIP_CHECK(ctx, rptlaunch::untile(gathered_dev, image_dev, width, height, tile_rows, world, rows_padded, (hipStream_t)stream));
    return RPT_OK;
}

int rpt_convert_to_u8_device(rpt_ctx* ctx, const float* pixels_dev, uint8_t* out_dev, uint32_t width, uint32_t height, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_convert_to_u8_device: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels_dev || !out_dev || width == 0 || height == 0) { set_err(ctx, "rpt_convert_to_u8_device: invalid argument"); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    RPT_HIP_CHECK(ctx, rptlaunch::convert_to_u8(pixels_dev, out_dev, (uint64_t)width * height, (hipStream_t)stream));
    return RPT_OK;
}

int rpt_convert_to_u8_at_device(rpt_ctx* ctx, const float* pixels_dev, uint32_t width, uint32_t height, uint8_t* frame_dev, uint32_t at_x,
                                uint32_t at_y, uint32_t frame_width, uint32_t frame_height, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_convert_to_u8_at_device: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels_dev || !frame_dev || width == 0 || height == 0 || frame_width == 0 || frame_height == 0) {
        set_err(ctx, "rpt_convert_to_u8_at_device: invalid argument");
        return RPT_ERR_INVALID_ARG;
    }
    RPT_ON_DEVICE(ctx);
    RPT_HIP_CHECK(ctx, rptlaunch::convert_to_u8_at(pixels_dev, width, height, frame_dev, at_x, at_y, frame_width, frame_height, (hipStream_t)stream));
    return RPT_OK;
}

int rpt_convert_to_u8(rpt_ctx* ctx, const float* pixels, uint8_t* frame, uint32_t width, uint32_t height)
{
    if (!ctx) { set_err(nullptr, "rpt_convert_to_u8: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels || !frame || width == 0 || height == 0) { set_err(ctx, "rpt_convert_to_u8: invalid argument"); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    const size_t n = (size_t)width * height;
    const size_t bytes = n * 16 + n * 4;                              // f32 RGBA in, u8 RGBA out, one allocation
    if (bytes > ctx->fb_bytes) {
        if (ctx->fb) { RPT_HIP_CHECK(ctx, hipFree(ctx->fb)); ctx->fb = nullptr; ctx->fb_bytes = 0; }
        RPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->fb, bytes));
        ctx->fb_bytes = bytes;
    }
    uint8_t* out_dev = reinterpret_cast<uint8_t*>(ctx->fb) + n * 16;
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->fb, pixels, n * 16, hipMemcpyHostToDevice, ctx->stream));
    int rc = rpt_convert_to_u8_device(ctx, ctx->fb, out_dev, width, height, ctx->stream);
    if (rc != RPT_OK) return rc;
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(frame, out_dev, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    RPT_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return RPT_OK;
}

int rpt_synchronize(rpt_ctx* ctx, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_synchronize: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    RPT_HIP_CHECK(ctx, hipStreamSynchronize((hipStream_t)stream));
    return RPT_OK;
}

int rpt_probe_rays(rpt_ctx* ctx, const float* rays_dev, uint32_t* out_dev, uint64_t n, uint32_t use_grid, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_probe_rays: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_scene || !ctx->large) { set_err(ctx, "rpt_probe_rays: needs an uploaded large scene"); return RPT_ERR_NO_SCENE; }
    if (!rays_dev || !out_dev) { set_err(ctx, "rpt_probe_rays: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if (n == 0) return RPT_OK;
    RPT_ON_DEVICE(ctx);
    SceneLarge sc = ctx->scene_large;
    if (!use_grid) sc.use_grid = 0;
    RPT_HIP_CHECK(ctx, rptlaunch::probe_rays(sc, rays_dev, out_dev, n, (hipStream_t)stream));
    return RPT_OK;
}

int rpt_probe_math(rpt_ctx* ctx, uint32_t fn, const float* a_dev, const float* b_dev, float* out_dev, uint64_t n, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_probe_math: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!a_dev || !b_dev || !out_dev || fn > RPT_PROBE_RNG) { set_err(ctx, "rpt_probe_math: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if (n == 0) return RPT_OK;
    RPT_ON_DEVICE(ctx);
    RPT_HIP_CHECK(ctx, rptlaunch::probe_math(fn, a_dev, b_dev, out_dev, n, (hipStream_t)stream));
    return RPT_OK;
}

}  // extern "C"

#ifdef RPT_PROFILE_BLOCKS
// Development build only (dev_prof.h, tools/block_profile.py): not part of include/rpt.h.
namespace rptlaunch { hipError_t prof_read(unsigned long long* out); }
extern "C" int rpt_prof_read(unsigned long long* out)
{
    return rptlaunch::prof_read(out) == hipSuccess ? RPT_OK : RPT_ERR_HIP;
}
#endif
