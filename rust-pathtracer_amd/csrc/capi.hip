// capi.hip — the C ABI of include/rpt.h: contexts (one device, the devices of a node in one process, or one rank of a
// multi-process job), scene upload, launches, the RCCL gather.  Host code only; the kernels and their launch functions
// are in the k_*.hip translation units (launch.h); every environment variable the library reads is in knobs.h.
//
// There is NO CPU fallback: without a gfx950 device every entry point that computes returns
// RPT_ERR_NO_DEVICE / RPT_ERR_HIP.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>              // types and prototypes only: librccl.so.1 is loaded on demand (rccl_api)

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rpt.h"
#include "host_scene.h"
#include "knobs.h"
#include "launch.h"
#ifdef RPT_TEST_HOOKS
#include "../../include/rpt_test.h"
#endif

using namespace rptdev;
using rpthost::HostAccel;
using rpthost::build_accel;
using rpthost::make_camera;
using rpthost::knobs;

// What one device of a context owns.
struct DevState {
    int device = -1;
    int rank = 0;                     // this device's rank in the world (row blocks b with b % world == rank)
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_ready = nullptr;
    SceneLarge scene_large;           // device pointers into `tables`
    void* tables = nullptr;           // one allocation holding a large scene's tables
    float* fb = nullptr;              // staging for the host-pointer API (this device's rows, or a whole image)
    size_t fb_bytes = 0;
    float* tile = nullptr;            // resident ColorBuffer rows of this rank: rows_padded x width RGBA f32
    // The gather runs beside the next render (gather_to_root): it sends a SNAPSHOT of the tile on a stream of its own.
    float* snap = nullptr;            // the tile as it was when the gather was asked for
    hipStream_t comm_stream = nullptr;
    hipEvent_t snap_ready = nullptr;  // on `stream`: the snapshot is taken
    hipEvent_t snap_free = nullptr;   // on `comm_stream`: the snapshot has been sent (the next one may overwrite it)
    bool snap_used = false;
    float* dn = nullptr;              // the denoiser's intermediate buffer, grown on demand
    size_t dn_bytes = 0;
    // `dn` is scratch of the CONTEXT, while rpt_denoise_device runs on whatever stream the caller passes: a use on another stream
    // than the previous one waits for that one's event (same stream: ordered anyway)
    hipEvent_t dn_done = nullptr;
    hipStream_t dn_stream = nullptr;
    bool dn_used = false;
    // dispatch (kernel_common.h, "Dispatch: units, their order, their hand-off"), for launches of `sched_tiles` tiles: per tile 4 dwords of
    // cost, 1 of order, 1 of sorting scratch, 4 of start stamps (development), then the hand-off words (SchedLayout)
    uint32_t* sched = nullptr;        // the tables of the CURRENT launch shape (an entry of sched_cache)
    uint32_t last_choice = 0;         // the last launch's KernelChoice as bits (include/rpt_test.h, rpt_debug_kernel_choice)
    uint32_t sched_tiles = 0;
    uint64_t sched_launches = 0;      // launches since the order was last started from scratch (the costs are re-sorted after the 1st, 2nd, 4th, ...)
    // A context that alternates between launch shapes (a viewer's preview and full frames, bench.py's legs, one rank's tile and the
    // whole frame) keeps each shape's learned order: up to kSchedCache tables, keyed by what decides a tile's cost (sched_for).
    struct SchedEntry { uint64_t key[2]; uint32_t* buf; uint32_t tiles; uint64_t launches; uint64_t stamp; };
    std::vector<SchedEntry> sched_cache;
    uint64_t sched_key[2] = {0, 0};
    uint64_t sched_clock = 0;
    hipEvent_t sched_done = nullptr;
    hipStream_t sched_stream = nullptr;
    bool sched_used = false;
    bool sync_used = false;           // a chunked launch has run: the hand-off's timeout word is worth a look (check_handoffs)
    ncclComm_t comm = nullptr;
};

struct rpt_ctx {
    std::vector<DevState> devs;       // devs[0] is the context's "own" device (rank 0 in a multi context)
    int world = 1;
    bool use_comm = false;            // tiles are gathered through RCCL (false: world 1, or peer copies)
    bool peer_gather = false;         // single process, RPT_GATHER=p2p: hipMemcpyPeerAsync instead of RCCL
    uint32_t tile_rows = 2;
    uint32_t dispatch[4] = {0xFFFFFFFFu, 0, 0, 0};   // rpt_set_dispatch: cost_order (0xFFFFFFFF: the environment's defaults), unit_rounds, unit_min_spp, unit_slots
    bool has_scene = false;
    bool large = false;               // scene exceeds the kernarg tables: SceneLarge + device tables
    bool media = false;               // RPT_SCENE_MEDIA and some material carries a medium: the media kernels (dev_media.h)
    SceneSmallSdf scene;              // camera part is filled per launch (depends on width/height); sdf.n_prims == 0: plain
    bool class_map_ok = false;        // small scenes of 5-12 primitives: their accepted sets fall into at most 16 classes of equal material
    MatClassMap class_map = {};       // (launch.h; `cls` is filled per device at launch: the 4 096-byte map is DevState::tables of such a scene)
    rpt_camera camera;
    // resident ColorBuffer (buffer.rs:6-14): pixels as per-rank tiles + frames
    uint32_t res_w = 0, res_h = 0, res_tile_rows = 0, res_rows_padded = 0;
    uint64_t res_frames = 0;
    bool has_res = false;
    // on the root device: rank-major gathered tiles, the assembled image, the u8 frame
    float* gathered = nullptr;
    float* image = nullptr;
    uint8_t* frame_u8 = nullptr;
    void* stage = nullptr;            // page-locked host staging for downloads into pageable buffers (download_to_host)
    size_t stage_bytes = 0;
    hipEvent_t gather_done = nullptr;       // on the root's comm_stream: the assembled image of the last gather is complete
    bool gather_issued = false;
    bool timed = false;               // ev_begin / ev_end bracket a render
    std::string err;

    bool is_root() const { return devs[0].rank == 0; }
    bool plain() const { return world == 1 && !use_comm; }           // rpt_create: the tile IS the image
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

#define RPT_CHECK_RC(call) do { const int rc_ = (call); if (rc_ != RPT_OK) return rc_; } while (0)

// Every entry point runs on its context's device(s) and puts the caller's current device back afterwards
// (the caller may be a torch process with its own idea of the current device).
struct DeviceGuard {
    int prev = -1;
    hipError_t status;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        status = (prev == device) ? hipSuccess : hipSetDevice(device);
    }
    hipError_t to(int device) { return hipSetDevice(device); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define RPT_ON_DEVICE(ctx)                        \
    DeviceGuard guard_((ctx)->devs[0].device);    \
    RPT_HIP_CHECK(ctx, guard_.status)

// ---- RCCL, loaded on demand ------------------------------------------------------------------------------
// Only multi-GPU contexts need it, and a host process (torch) may already have its own copy of librccl.so.1 loaded:
// dlopen by soname then returns that one instead of bringing in a second runtime.
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

// Loads librccl once per process (thread-safe: a function-local static's initialiser); false when it cannot be loaded.
static bool rccl_load(RcclApi& api)
{
    const char* forced = knobs().rccl_lib.empty() ? nullptr : knobs().rccl_lib.c_str();   // tests: a name that cannot be loaded exercises the error path
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (forced) api.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    else
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
    if (!api.handle) {
        const char* e = dlerror();                                   // (the call clears the pending error: read it once)
        api.error = e ? e : "dlopen(librccl.so.1) failed";
        return false;
    }
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(api.handle, name); if (!p) { ok = false; api.error = std::string("librccl: missing symbol ") + name; } return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(api.handle); api.handle = nullptr; return false; }
    return true;
}

static RcclApi g_rccl;
static RcclApi* rccl_api()
{
    static const bool loaded = rccl_load(g_rccl);
    return loaded ? &g_rccl : nullptr;
}
static const char* rccl_why() { return g_rccl.error.empty() ? "unknown reason" : g_rccl.error.c_str(); }

#define RPT_RCCL_CHECK(ctx, api, call)                                                            \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess) {                                                                  \
            set_err(ctx, "%s failed: %s (%s:%d)", #call, (api)->GetErrorString(r_), __FILE__, __LINE__); \
            return RPT_ERR_RCCL;                                                                  \
        }                                                                                         \
    } while (0)

// Where things are in DevState::sched (dwords), for n tiles.
struct SchedLayout {
    size_t n;
    size_t cost() const { return 0; }
    size_t order() const { return 4 * n; }
    size_t sorted() const { return 5 * n; }
    size_t start() const { return 6 * n; }
    size_t sync() const { return 10 * n; }                          // kSyncTimeout (kept), then from kSyncTicket on: zeroed before a chunked launch
    size_t total() const { return 10 * n + 32 + n; }
};
constexpr size_t kSyncTimeoutWord = 0, kSyncZeroFrom = 16, kSyncDoneFrom = 32;     // = kernel_common.h's kSyncTimeout / kSyncTicket / kSyncDone

// Dispatch policy of a context (include/rpt.h, rpt_set_dispatch); the environment gives the defaults.
struct DispatchPolicy {
    uint32_t cost_order, unit_rounds, unit_min_spp, unit_slots;
};
static DispatchPolicy default_dispatch() { return DispatchPolicy{knobs().dispatch_order, knobs().unit_rounds, knobs().unit_min_spp, 0u}; }

static DispatchPolicy policy_of(const rpt_ctx* ctx)
{
    if (ctx->dispatch[0] == 0xFFFFFFFFu) return default_dispatch();
    return DispatchPolicy{ctx->dispatch[0], ctx->dispatch[1], ctx->dispatch[2], ctx->dispatch[3]};
}

// How many chunks of samples a launch of `nblocks` tiles x `spp` samples is cut into (kernel_common.h, units): enough for
// `unit_rounds` rounds of workgroups on the device, no chunk shorter than `unit_min_spp` samples; 1 when the tiles alone are
// that many rounds, or fit the device at once (then nothing waits for a slot and there is nothing to balance).  Measured
// (tools/tile_rows_time.py, tools/launch_size_time.py): one rank's share of configs[2] (3.2 rounds, 1 024 spp) 1 / 2 / 4 / 8
// chunks 11.05 / 10.92 / 11.22 / 11.24 Gsamples/s; 800x600 x 128 spp (1.5 rounds) 1 / 4 chunks 8.75 / 10.2; configs[1] (6.4
// rounds) 1 / 2 / 4 chunks 11.75 / 11.73 / 11.47; configs[3] (64 spp) 1 / 2 chunks 3.05 / 2.97: a chunk's end drains every wave.
static uint32_t unit_chunks(const DispatchPolicy& pol, uint64_t nblocks, uint32_t spp, uint32_t slots)
{
    if (pol.unit_rounds == 0u || pol.unit_min_spp == 0u || nblocks <= slots || spp < 2u * pol.unit_min_spp) return 1u;
    // From a third of the target on (configs[1]: 6.4 rounds) cutting buys nothing — 11.75 against 11.73 Gsamples/s — and every chunk
    // reads and writes the pixels once more (HBM traffic per launch 148 MB instead of 80): such launches stay whole.
    if (nblocks * 3u >= (uint64_t)pol.unit_rounds * slots) return 1u;
    const uint64_t want = ((uint64_t)pol.unit_rounds * slots + nblocks - 1u) / nblocks;
    const uint64_t most = spp / pol.unit_min_spp;
    const uint64_t n = want < most ? want : most;
    return n < 1u ? 1u : (uint32_t)n;
}

// Launches of at least this many samples per pixel re-sort the order from their own costs every time (one small kernel behind
// the launch); shorter ones only after the 1st, 2nd, 4th, 8th ... launch since the order was started.
constexpr uint32_t kOrderAlwaysFromSpp = 16;
// Small scenes: launches of at most knobs().compact_max_spp samples per pixel take the compacting kernel (k_compact.hip).
// (1 since round 3: 1080p, 1 spp 7.12 vs 6.83 Gsamples/s for the megakernel, 2 spp 7.01 vs 7.39: profiles/r3/spp_curve.txt)
// ---- descriptor -> device tables ---------------------------------------------------------------------------
static DevPlane dev_plane(const rpt_plane& a) { return DevPlane{a.normal[0], a.normal[1], a.normal[2], a.point[0], a.point[1], a.point[2], a.min_denom, a.material, a.max_t}; }
static DevLight dev_light(const rpt_light& a)
{
    return DevLight{a.type, a.position[0], a.position[1], a.position[2], a.emission[0], a.emission[1], a.emission[2], a.radius, a.area,
                    a.u[0], a.u[1], a.u[2], a.v[0], a.v[1], a.v[2]};
}
static DevMaterial dev_material(const rpt_material& a)
{
    DevMaterial m;
    m.mask = a.mask; m.proc_kind = a.proc_kind;
    for (int k = 0; k < 3; ++k) { m.rgb[k] = a.rgb[k]; m.emission[k] = a.emission[k]; }
    m.anisotropic = a.anisotropic; m.metallic = a.metallic; m.roughness = a.roughness; m.subsurface = a.subsurface;
    m.specular_tint = a.specular_tint; m.sheen = a.sheen; m.sheen_tint = a.sheen_tint; m.clearcoat = a.clearcoat;
    m.clearcoat_gloss = a.clearcoat_gloss; m.spec_trans = a.spec_trans; m.ior = a.ior;
    for (int k = 0; k < 4; ++k) m.proc_params[k] = a.proc_params[k];
    m.medium_type = a.medium_type; m.medium_density = a.medium_density; m.medium_anisotropy = a.medium_anisotropy;
    for (int k = 0; k < 3; ++k) m.medium_color[k] = a.medium_color[k];
    return m;
}
static DevBackground dev_background(const rpt_background& b)
{
    return DevBackground{b.kind, b.colour_a[0], b.colour_a[1], b.colour_a[2], b.colour_b[0], b.colour_b[1], b.colour_b[2], b.gamma, b.scale};
}

static void free_dev(DevState& d)
{
    DeviceGuard guard(d.device);
    if (d.fb) (void)hipFree(d.fb);
    if (d.tile) (void)hipFree(d.tile);
    if (d.tables) (void)hipFree(d.tables);
    if (d.dn) (void)hipFree(d.dn);
    for (DevState::SchedEntry& e : d.sched_cache) if (e.buf) (void)hipFree(e.buf);
    if (d.sched_done) (void)hipEventDestroy(d.sched_done);
    if (d.ev_begin) (void)hipEventDestroy(d.ev_begin);
    if (d.ev_end) (void)hipEventDestroy(d.ev_end);
    if (d.ev_ready) (void)hipEventDestroy(d.ev_ready);
    if (d.snap) (void)hipFree(d.snap);
    if (d.snap_ready) (void)hipEventDestroy(d.snap_ready);
    if (d.snap_free) (void)hipEventDestroy(d.snap_free);
    if (d.comm_stream) (void)hipStreamDestroy(d.comm_stream);
    if (d.dn_done) (void)hipEventDestroy(d.dn_done);
    if (d.stream) (void)hipStreamDestroy(d.stream);
    d = DevState();
}

// device checks + stream/events for one device of a context
static int open_dev(DevState& d, int device_id, int rank, const char* who)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_err(nullptr, "%s: no HIP device (%s); this library has no CPU fallback", who, e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return RPT_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= count) { set_err(nullptr, "%s: device %d out of range [0,%d)", who, device_id, count); return RPT_ERR_INVALID_ARG; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) { set_err(nullptr, "%s: hipGetDeviceProperties failed", who); return RPT_ERR_HIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(nullptr, "%s: device %d is %s; this library is built for gfx950 only", who, device_id, prop.gcnArchName);
        return RPT_ERR_NO_DEVICE;
    }
    d.device = device_id;
    d.rank = rank;
    DeviceGuard guard(device_id);
    if (guard.status != hipSuccess || hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&d.ev_begin) != hipSuccess || hipEventCreate(&d.ev_end) != hipSuccess ||
        hipEventCreateWithFlags(&d.ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&d.comm_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&d.snap_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d.snap_free, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d.dn_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&d.sched_done, hipEventDisableTiming) != hipSuccess) {
        set_err(nullptr, "%s: cannot create a stream on device %d", who, device_id);
        free_dev(d);
        return RPT_ERR_HIP;
    }
    return RPT_OK;
}

static void free_resident(rpt_ctx* ctx)
{
    for (DevState& d : ctx->devs) {
        DeviceGuard guard(d.device);
        if (d.tile) { (void)hipFree(d.tile); d.tile = nullptr; }
        if (d.snap) { (void)hipFree(d.snap); d.snap = nullptr; }
        d.snap_used = false;
    }
    DeviceGuard guard(ctx->devs[0].device);
    if (ctx->gathered) { (void)hipFree(ctx->gathered); ctx->gathered = nullptr; }
    if (ctx->image) { (void)hipFree(ctx->image); ctx->image = nullptr; }
    if (ctx->frame_u8) { (void)hipFree(ctx->frame_u8); ctx->frame_u8 = nullptr; }
    if (ctx->stage) { (void)hipHostFree(ctx->stage); ctx->stage = nullptr; ctx->stage_bytes = 0; }
    if (ctx->gather_done) { (void)hipEventDestroy(ctx->gather_done); ctx->gather_done = nullptr; }
    ctx->gather_issued = false;
    ctx->has_res = false;
    ctx->res_w = ctx->res_h = ctx->res_tile_rows = ctx->res_rows_padded = 0;
    ctx->res_frames = 0;
}

static uint32_t rows_padded_for(uint32_t height, uint32_t tile_rows, uint32_t world) { return tile_rows_padded(height, tile_rows, world); }

// Copy the rows rank `rank` owns between a host top-down image and its compact tile (either direction), following
// rpt_tile_copy_plan: one strided copy for the full blocks plus one plain copy when the rank owns the image's short last block.
static hipError_t copy_rank_rows(bool to_device, float* host_image, float* tile, uint32_t width, uint32_t height, uint32_t tile_rows,
                                 uint32_t rank, uint32_t world, hipStream_t st)
{
    const size_t row_bytes = (size_t)width * 16u;
    const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    rpt_tile_plan p;
    if (rpt_tile_copy_plan(height, tile_rows, rank, world, &p) != RPT_OK) return hipErrorInvalidValue;
    char* h0 = reinterpret_cast<char*>(host_image);
    char* t0 = reinterpret_cast<char*>(tile);
    if (p.full_blocks) {
        const size_t block_bytes = row_bytes * p.block_rows;
        char* h = h0 + row_bytes * p.host_row0;
        const size_t hpitch = row_bytes * p.host_row_stride;
        hipError_t e;
        if (p.full_blocks == 1) e = to_device ? hipMemcpyAsync(t0, h, block_bytes, kind, st) : hipMemcpyAsync(h, t0, block_bytes, kind, st);
        else e = to_device ? hipMemcpy2DAsync(t0, block_bytes, h, hpitch, block_bytes, p.full_blocks, kind, st)
                           : hipMemcpy2DAsync(h, hpitch, t0, block_bytes, block_bytes, p.full_blocks, kind, st);
        if (e != hipSuccess) return e;
    }
    if (p.ragged_rows) {
        char* h = h0 + row_bytes * p.ragged_host_row0;
        char* t = t0 + row_bytes * p.ragged_tile_row0;
        return to_device ? hipMemcpyAsync(t, h, row_bytes * p.ragged_rows, kind, st) : hipMemcpyAsync(h, t, row_bytes * p.ragged_rows, kind, st);
    }
    return hipSuccess;
}

// The dispatch tables for a launch of `nblocks` tiles of this shape: the context's current ones if the shape is the same, else the
// cached ones of that shape (their learned order intact), else new ones (bottom rows first, no costs).  Nothing is freed — and so
// nothing waits for the device — unless kSchedCache shapes are already held; then the least recently used goes.
constexpr size_t kSchedCache = 6;
static int sched_for(rpt_ctx* ctx, DevState& d, uint32_t nblocks, uint32_t width, uint32_t rows_local, uint32_t tile_rows, uint32_t rank, uint32_t world, hipStream_t stream)
{
    const uint64_t key[2] = {((uint64_t)width << 32) | rows_local, ((uint64_t)tile_rows << 40) ^ ((uint64_t)rank << 20) ^ (uint64_t)world};
    d.sched_clock += 1;
    if (d.sched && d.sched_key[0] == key[0] && d.sched_key[1] == key[1] && d.sched_tiles == nblocks) {
        for (DevState::SchedEntry& e : d.sched_cache) if (e.buf == d.sched) e.stamp = d.sched_clock;
        return RPT_OK;
    }
    for (DevState::SchedEntry& e : d.sched_cache) if (e.buf == d.sched) e.launches = d.sched_launches;     // park the current shape
    DevState::SchedEntry* hit = nullptr;
    for (DevState::SchedEntry& e : d.sched_cache) if (e.key[0] == key[0] && e.key[1] == key[1] && e.tiles == nblocks) hit = &e;
    if (!hit) {
        if (d.sched_cache.size() >= kSchedCache) {
            size_t lru = 0;
            for (size_t i = 1; i < d.sched_cache.size(); ++i) if (d.sched_cache[i].stamp < d.sched_cache[lru].stamp) lru = i;
            RPT_HIP_CHECK(ctx, hipFree(d.sched_cache[lru].buf));     // (hipFree waits for the device: rare by construction)
            d.sched_cache.erase(d.sched_cache.begin() + (long)lru);
        }
        const SchedLayout lay{(size_t)nblocks};
        uint32_t* buf = nullptr;
        RPT_HIP_CHECK(ctx, hipMalloc((void**)&buf, lay.total() * sizeof(uint32_t)));
        d.sched_cache.push_back(DevState::SchedEntry{{key[0], key[1]}, buf, nblocks, 0, d.sched_clock});
        hit = &d.sched_cache.back();
        RPT_HIP_CHECK(ctx, hipMemsetAsync(buf + lay.sync(), 0, (32 + lay.n) * sizeof(uint32_t), stream));
        RPT_HIP_CHECK(ctx, rptlaunch::sched_init(buf + lay.cost(), buf + lay.order(), nblocks, stream));
    }
    hit->stamp = d.sched_clock;
    d.sched = hit->buf;
    d.sched_tiles = hit->tiles;
    d.sched_launches = hit->launches;
    d.sched_key[0] = key[0]; d.sched_key[1] = key[1];
    return RPT_OK;
}

// The classes of accepted sets of a small scene of 5-12 primitives (launch.h, MatClassMap).  The material of a hit is Material::new()
// overwritten field by field by the accepted primitives in order (apply_patch_fields; a procedural patch writes rgb whatever its
// mask says: apply_patch_row), so two sets give the same material when every field has the same last writer in both.  False: the
// scene is not one the mapped table serves (fewer than 5 primitives, two procedural materials, more than 16 classes).
static bool material_class_map(const SceneSmall& sc, MatClassMap& map, std::vector<uint8_t>& cls)
{
    const uint32_t ns = sc.n_spheres, np = sc.n_planes, nb = ns + np;
    if (nb < 5u || nb > 12u) return false;
    uint32_t n_procedural = 0;
    uint32_t mask_of[kMaxSpheres + kMaxPlanes];
    for (uint32_t i = 0; i < nb; ++i) {
        const DevMaterial& m = sc.materials[i < ns ? sc.spheres[i].material : sc.planes[i - ns].material];
        n_procedural += m.proc_kind != 0u;
        mask_of[i] = (m.mask & (uint32_t)RPT_MAT_ALL) | (m.proc_kind == RPT_PROC_CHECKER_DIR ? (uint32_t)RPT_MAT_RGB : 0u);
    }
    if (n_procedural > 1u) return false;
    memset(&map, 0, sizeof(map));
    cls.assign(4096, 0);
    struct Signature { uint8_t last[13]; bool operator==(const Signature& o) const { return memcmp(last, o.last, sizeof(last)) == 0; } };
    std::vector<Signature> classes;
    for (uint32_t set = 0; set < (1u << nb); ++set) {
        Signature sig;
        memset(sig.last, 0xFF, sizeof(sig.last));
        for (uint32_t i = 0; i < nb; ++i)
            if ((set >> i) & 1u)
                for (uint32_t f = 0; f < 13u; ++f) if ((mask_of[i] >> f) & 1u) sig.last[f] = (uint8_t)i;
        size_t c = 0;
        while (c < classes.size() && !(classes[c] == sig)) ++c;
        if (c == classes.size()) {
            if (classes.size() == kMatClasses) return false;
            classes.push_back(sig);
            map.class_set[c] = (uint16_t)((set & ((1u << ns) - 1u)) | ((set >> ns) << kMaxSpheres));      // (GeomHit.code's layout: planes from bit 8)
        }
        cls[set] = (uint8_t)c;
    }
    map.n_classes = (uint32_t)classes.size();
    return true;
}

// One render launch sequence on one device.
static int launch_render(rpt_ctx* ctx, DevState& d, float* pixels_dev, uint32_t width, uint32_t height, uint64_t frames_done, uint32_t spp,
                         uint64_t seed, uint32_t flags, uint32_t tile_rows, uint32_t rank, uint32_t world, hipStream_t stream)
{
    if (world == 1) tile_rows = height;                              // one block: local row == global row
    if (flags & ~(uint32_t)RPT_RENDER_ALL_FLAGS) {
        set_err(ctx, "render: unknown flag bits 0x%x (bits 2-4, 6-7, 9-10 named A/B kernel forms until ABI 3; they are gone, include/rpt.h)", flags & ~(uint32_t)RPT_RENDER_ALL_FLAGS);
        return RPT_ERR_INVALID_ARG;
    }
    SceneSmallSdf scs = ctx->scene;
    SceneLarge scl = d.scene_large;
    scs.cam = scl.cam = make_camera(ctx->camera, (float)width, (float)height);
    const bool has_sdf = !ctx->large && scs.sdf.n_prims > 0;
    const bool nested = (flags & RPT_RENDER_NESTED_LOOPS) != 0;
    const bool fast = (flags & RPT_RENDER_FAST_MATH) != 0;

    RenderParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.pixels = pixels_dev;
    rp.width = width; rp.height = height;
    rp.rows_local = tile_row_count(height, tile_rows, rank, world);
    rp.tile_rows = tile_rows; rp.rank = rank; rp.world = world;
    rp.seed = seed;
    rp.tiles_x = (width + 15u) / 16u;
    rp.shade_threshold = has_sdf ? knobs().sdf_shade_room : knobs().shade_threshold;      // (k_sdf.hip: the second room's threshold, 0 = one block)
    // (small scenes' megakernel only; 8 ... 48 are within 2 % of each other, +5.9 % over finishing un-voted)
    rp.finish_threshold = knobs().finish_threshold;
    rp.march_min_lanes = knobs().sdf_march_min_lanes;
    rp.compact = (!ctx->large && !has_sdf && ((flags & RPT_RENDER_SMALL_COMPACT) || spp <= knobs().compact_max_spp)) ? 1u : 0u;
    if (flags & RPT_RENDER_RUSSIAN_ROULETTE) { scs.flags |= kSceneFlagRussianRoulette; scl.flags |= kSceneFlagRussianRoulette; }
    if (rp.rows_local == 0) return RPT_OK;
    const uint32_t tiles_y = (rp.rows_local + 15u) / 16u;
    const uint64_t nblocks = (uint64_t)rp.tiles_x * tiles_y;
    if (nblocks > 0x7FFFFFFFull) { set_err(ctx, "render: grid too large"); return RPT_ERR_INVALID_ARG; }
    // The nested-loop kernel is the differential baseline of the reference's own scene class; the other classes have one form.
    if (nested && (ctx->large || has_sdf || ctx->media)) {
        set_err(ctx, "render: RPT_RENDER_NESTED_LOOPS exists for small scenes without an SDF object or media only");
        return RPT_ERR_UNSUPPORTED;
    }
    if (ctx->media && fast) {
        set_err(ctx, "render: scenes with participating media (RPT_SCENE_MEDIA) have no relaxed-arithmetic kernel form");
        return RPT_ERR_UNSUPPORTED;
    }

    // Which instantiation (launch.h, KernelChoice): the kernels that know the reference scene's table sizes, and those that read a
    // hit's material from a table (at most three primitives, at most one of them with a procedural material).
    KernelChoice kc;
    {
        const SceneSmall& sc = scs;
        const bool can_size = !knobs().no_sized_kernels && !ctx->media && !ctx->large && !nested;
        kc.sized = can_size && !has_sdf && sc.n_spheres == 2u && sc.n_planes == 1u && sc.n_lights == 1u;      // (kernel_common.h, RPT_REFERENCE_SIZES)
        kc.sized_sdf = (can_size && has_sdf && sc.n_planes == 1u && sc.n_lights == 1u && scs.sdf.n_prims <= 4u) ? scs.sdf.n_prims : 0u;
        kc.material_table = !knobs().no_material_table && !ctx->media && !ctx->large && !nested && rptlaunch::material_table_fits_small(scs, has_sdf);      // (with or without the sizes)
        kc.material_table_wide = !knobs().no_material_table && !ctx->media && !ctx->large && !nested && !has_sdf && !rp.compact && rptlaunch::material_table_fits_small(scs, false, 4u);
        // five to twelve primitives: the table by class of accepted set (launch.h, MatClassMap), in the megakernel of small scenes
        if (!kc.material_table && !kc.material_table_wide && !knobs().no_material_table && !ctx->media && !ctx->large && !nested && !has_sdf && !rp.compact &&
            ctx->class_map_ok && d.tables) {
            kc.material_table_mapped = true;
            kc.class_map = ctx->class_map;
            kc.class_map.cls = reinterpret_cast<const uint8_t*>(d.tables);
        }
        kc.extra_lds = knobs().debug_extra_lds;
        d.last_choice = (kc.sized ? 1u : 0u) | (kc.material_table ? 2u : 0u) | (kc.material_table_wide ? 4u : 0u) | (kc.material_table_mapped ? 8u : 0u) |
                        ((kc.material_table_mapped ? kc.class_map.n_classes : 0u) << 8) | (kc.sized_sdf << 16);
    }
    const auto launch = [&](uint32_t grid) -> hipError_t {
        if (ctx->large) return fast ? rptlaunch_fast::render_large(scl, false, rp, grid, stream) : rptlaunch::render_large(scl, ctx->media, rp, grid, stream);
        if (has_sdf) return fast ? rptlaunch_fast::render_sdf(scs, false, rp, grid, stream, kc) : rptlaunch::render_sdf(scs, ctx->media, rp, grid, stream, kc);
        if (rp.compact && !nested) return fast ? rptlaunch_fast::render_compact(scs, false, rp, grid, stream, kc) : rptlaunch::render_compact(scs, ctx->media, rp, grid, stream, kc);
        return fast ? rptlaunch_fast::render_small(scs, false, nested, rp, grid, stream, kc) : rptlaunch::render_small(scs, ctx->media, nested, rp, grid, stream, kc);
    };

    // Dispatch (kernel_common.h): this device's launches of `nblocks` tiles run most expensive tile first, as measured by the previous
    // one, and in units of one tile x one chunk of the samples.
    // Kernels without units: nested loops, the compacting kernel of small scenes.
    const bool unit_kernel = !nested && !rp.compact;
    const SchedLayout lay{(size_t)nblocks};
    bool reorder = false;
    DispatchPolicy pol = policy_of(ctx);
    // workgroup slots of the device: 5 workgroups per CU (__launch_bounds__(256, 5))
    static const int n_cu = []() { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const uint32_t slots = pol.unit_slots ? pol.unit_slots : (uint32_t)n_cu * 5u;
    // The compacting kernel of one-sample launches below three rounds of workgroups stays bottom rows first: most expensive first
    // costs 8 % at the reference's 800x600 window (1.5 rounds; 0.081 -> 0.088 ms) and gains 2 % at 1920x1080 (6.4 rounds).
    if (!unit_kernel && nblocks < 3ull * slots) pol.cost_order = 0u;
    if (!nested && (pol.cost_order != 0u || unit_kernel)) {
        RPT_CHECK_RC(sched_for(ctx, d, (uint32_t)nblocks, width, rp.rows_local, tile_rows, rank, world, stream));
        if (d.sched_used && d.sched_stream != stream) RPT_HIP_CHECK(ctx, hipStreamWaitEvent(stream, d.sched_done, 0));
        if (pol.cost_order != 0u) {
            d.sched_launches += 1;
            reorder = spp >= kOrderAlwaysFromSpp || (d.sched_launches & (d.sched_launches - 1u)) == 0u;
            rp.tile_order = d.sched + lay.order();
            rp.tile_cost = reorder ? d.sched + lay.cost() : nullptr;
            rp.tile_start = (reorder && knobs().dispatch_timeline) ? d.sched + lay.start() : nullptr;
        }
        rp.sched_sync = d.sched + lay.sync();
    }

    const uint32_t max_chunk = rptlaunch::max_spp_per_launch(has_sdf);
    if (unit_kernel) {
        // ONE launch whatever spp is: the LDS tables of the state-machine kernels hold a chunk's samples, and a launch is as many
        // chunks as it takes.
        uint32_t n_chunks = unit_chunks(pol, nblocks, spp, slots);
        uint32_t chunk_spp = (spp + n_chunks - 1u) / n_chunks;
        if (chunk_spp > max_chunk) chunk_spp = max_chunk;
        n_chunks = (spp + chunk_spp - 1u) / chunk_spp;
        if (nblocks * n_chunks > 0x7FFFFFFFull) { set_err(ctx, "render: grid too large (%llu tiles x %u chunks of samples)", (unsigned long long)nblocks, n_chunks); return RPT_ERR_INVALID_ARG; }
        rp.n_chunks = n_chunks;
        rp.chunk_spp = chunk_spp;
        rp.spp = spp;
        rp.frames_done = frames_done;
        if (n_chunks > 1u) RPT_HIP_CHECK(ctx, hipMemsetAsync(d.sched + lay.sync() + kSyncZeroFrom, 0, (32 - kSyncZeroFrom + lay.n) * sizeof(uint32_t), stream));
        RPT_HIP_CHECK(ctx, launch((uint32_t)(nblocks * n_chunks)));
        d.sync_used = d.sync_used || n_chunks > 1u;
    } else
    // Kernels without units: batches beyond what one launch holds are split into consecutive launches (the running mean carries
    // over in the framebuffer).
    for (uint32_t done = 0; done < spp;) {
        const uint32_t chunk = (spp - done > max_chunk) ? max_chunk : (spp - done);
        rp.spp = chunk;
        rp.frames_done = frames_done + done;
        rp.n_chunks = 0u;
        RPT_HIP_CHECK(ctx, launch((uint32_t)nblocks));
        done += chunk;
    }
    if (rp.tile_order || rp.sched_sync) {
        if (reorder && pol.cost_order != 2u)
            RPT_HIP_CHECK(ctx, rptlaunch::sched_order(d.sched + lay.cost(), d.sched + lay.order(), d.sched_tiles, stream));
        RPT_HIP_CHECK(ctx, hipEventRecord(d.sched_done, stream));
        d.sched_stream = stream;
        d.sched_used = true;
    }
    return RPT_OK;
}

// Behind a wait for the device: did a unit of a chunked launch give up waiting for its tile's previous chunk (kernel_common.h,
// unit_begin)?  It cannot happen by construction (the predecessor holds an earlier ticket); if it ever does the image is wrong —
// the unit has poisoned its pixels with NaN — and the caller must know.  The word is cleared once it has been reported.
static int check_handoffs(rpt_ctx* ctx, DevState& d)
{
    if (!d.sync_used) return RPT_OK;
    d.sync_used = false;
    bool any = false;
    for (DevState::SchedEntry& e : d.sched_cache) {
        uint32_t* word = e.buf + SchedLayout{(size_t)e.tiles}.sync() + kSyncTimeoutWord;
        uint32_t timed_out = 0;
        RPT_HIP_CHECK(ctx, hipMemcpy(&timed_out, word, sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (timed_out) { any = true; RPT_HIP_CHECK(ctx, hipMemset(word, 0, sizeof(uint32_t))); }
    }
    if (any) {
        set_err(ctx, "render: a workgroup timed out waiting for its tile's previous chunk of samples on device %d; the tile's pixels are NaN", d.device);
        return RPT_ERR_HIP;
    }
    return RPT_OK;
}
static int check_handoffs_all(rpt_ctx* ctx)
{
    DeviceGuard guard(ctx->devs[0].device);
    for (DevState& d : ctx->devs) {
        if (!d.sync_used) continue;
        RPT_HIP_CHECK(ctx, guard.to(d.device));
        RPT_CHECK_RC(check_handoffs(ctx, d));
    }
    return RPT_OK;
}

static int ensure_fb(rpt_ctx* ctx, DevState& d, size_t bytes)
{
    if (bytes > d.fb_bytes) {
        if (d.fb) { RPT_HIP_CHECK(ctx, hipFree(d.fb)); d.fb = nullptr; d.fb_bytes = 0; }
        RPT_HIP_CHECK(ctx, hipMalloc((void**)&d.fb, bytes));
        d.fb_bytes = bytes;
    }
    return RPT_OK;
}

// Page-locks a caller's host buffer for the duration of one call (rpt_render on several devices).  A buffer the caller
// has registered itself stays as it is; any other failure leaves the buffer pageable (the copies then take HIP's
// staging path: correct, less overlap).
struct HostPin {
    void* p = nullptr;
    void lock(void* ptr, size_t bytes)
    {
        const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterPortable);
        if (e == hipSuccess) p = ptr;
        else (void)hipGetLastError();
    }
    ~HostPin() { if (p) (void)hipHostUnregister(p); }
};

// The library is built with -fvisibility=hidden: what include/rpt.h (and, in the test build, include/rpt_test.h) declares is ALL it exports.
#pragma GCC visibility push(default)
extern "C" {

uint32_t rpt_abi_version(void) { return RPT_ABI_VERSION; }
uint32_t rpt_sizeof_scene_desc(void) { return (uint32_t)sizeof(rpt_scene_desc); }
uint32_t rpt_build_has_test_hooks(void)
{
#ifdef RPT_TEST_HOOKS
    return 1u;
#else
    return 0u;
#endif
}

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
    rpt_ctx* ctx = new (std::nothrow) rpt_ctx();
    if (!ctx) return RPT_ERR_HIP;
    ctx->devs.resize(1);
    int rc = open_dev(ctx->devs[0], device_id, 0, "rpt_create");
    if (rc != RPT_OK) { delete ctx; return rc; }
    *out = ctx;
    return RPT_OK;
}

int rpt_create_multi(rpt_ctx** out, const int* device_ids, int n_devices)
{
    if (!out) { set_err(nullptr, "rpt_create_multi: out is NULL"); return RPT_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) { set_err(nullptr, "rpt_create_multi: need 1..64 device ids"); return RPT_ERR_INVALID_ARG; }
    // A device may be listed more than once: every entry is a rank with its own stream and its share of the rows, and the
    // kernels of one GPU's ranks run side by side — a progressive render's launches then fill each other's tails (one MI355X, the
    // resident 1920x1080 frame: 11.4 -> 12.0 Gsamples/s with the device listed twice; 3840x270: 7.6 -> 9.3).  RCCL needs one device
    // per rank, so such contexts gather with peer copies (on one device: device-to-device copies); RPT_GATHER=p2p asks for that
    // with distinct devices too.
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j) distinct = distinct && device_ids[i] != device_ids[j];
    const bool peer = !distinct || knobs().gather == "p2p";
    rpt_ctx* ctx = new (std::nothrow) rpt_ctx();
    if (!ctx) return RPT_ERR_HIP;
    ctx->devs.resize((size_t)n_devices);
    ctx->world = n_devices;
    ctx->peer_gather = peer;
    ctx->use_comm = !peer;
    for (int i = 0; i < n_devices; ++i) {
        int rc = open_dev(ctx->devs[(size_t)i], device_ids[i], i, "rpt_create_multi");
        if (rc != RPT_OK) { for (DevState& d : ctx->devs) if (d.device >= 0) free_dev(d); delete ctx; return rc; }
    }
    if (ctx->use_comm) {
        RcclApi* api = rccl_api();
        int rc = RPT_OK;
        if (!api) { set_err(nullptr, "rpt_create_multi: cannot load RCCL: %s", rccl_why()); rc = RPT_ERR_RCCL; }
        else {
            std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
            DeviceGuard guard(device_ids[0]);
            ncclResult_t r = api->CommInitAll(comms.data(), n_devices, device_ids);
            if (r != ncclSuccess) { set_err(nullptr, "rpt_create_multi: ncclCommInitAll failed: %s", api->GetErrorString(r)); rc = RPT_ERR_RCCL; }
            else for (int i = 0; i < n_devices; ++i) ctx->devs[(size_t)i].comm = comms[(size_t)i];
        }
        if (rc != RPT_OK) { for (DevState& d : ctx->devs) free_dev(d); delete ctx; return rc; }
    }
    *out = ctx;
    return RPT_OK;
}

int rpt_comm_unique_id(rpt_unique_id* out)
{
    static_assert(sizeof(rpt_unique_id) == sizeof(ncclUniqueId), "rpt_unique_id carries an ncclUniqueId");
    if (!out) { set_err(nullptr, "rpt_comm_unique_id: out is NULL"); return RPT_ERR_INVALID_ARG; }
    RcclApi* api = rccl_api();
    if (!api) { set_err(nullptr, "rpt_comm_unique_id: cannot load RCCL: %s", rccl_why()); return RPT_ERR_RCCL; }
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) { set_err(nullptr, "rpt_comm_unique_id: ncclGetUniqueId failed: %s", api->GetErrorString(r)); return RPT_ERR_RCCL; }
    memcpy(out->bytes, &id, sizeof(id));
    return RPT_OK;
}

int rpt_create_rank(rpt_ctx** out, int device_id, int rank, int world, const rpt_unique_id* id)
{
    if (!out) { set_err(nullptr, "rpt_create_rank: out is NULL"); return RPT_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) { set_err(nullptr, "rpt_create_rank: invalid argument (rank %d of %d)", rank, world); return RPT_ERR_INVALID_ARG; }
    RcclApi* api = rccl_api();
    if (!api) { set_err(nullptr, "rpt_create_rank: cannot load RCCL: %s", rccl_why()); return RPT_ERR_RCCL; }
    rpt_ctx* ctx = new (std::nothrow) rpt_ctx();
    if (!ctx) return RPT_ERR_HIP;
    ctx->devs.resize(1);
    ctx->world = world;
    ctx->use_comm = true;
    int rc = open_dev(ctx->devs[0], device_id, rank, "rpt_create_rank");
    if (rc != RPT_OK) { delete ctx; return rc; }
    {
        DeviceGuard guard(device_id);
        ncclUniqueId nid;
        memcpy(&nid, id->bytes, sizeof(nid));
        ncclResult_t r = api->CommInitRank(&ctx->devs[0].comm, world, nid, rank);
        if (r != ncclSuccess) {
            set_err(nullptr, "rpt_create_rank: ncclCommInitRank failed: %s", api->GetErrorString(r));
            free_dev(ctx->devs[0]);
            delete ctx;
            return RPT_ERR_RCCL;
        }
    }
    *out = ctx;
    return RPT_OK;
}

int rpt_world(const rpt_ctx* ctx, int* rank, int* world, int* n_local)
{
    if (!ctx) return RPT_ERR_INVALID_ARG;
    if (rank) *rank = ctx->devs[0].rank;
    if (world) *world = ctx->world;
    if (n_local) *n_local = (int)ctx->devs.size();
    return RPT_OK;
}

int rpt_set_tile_rows(rpt_ctx* ctx, uint32_t tile_rows)
{
    if (!ctx || tile_rows == 0) { set_err(ctx, "rpt_set_tile_rows: invalid argument"); return RPT_ERR_INVALID_ARG; }
    ctx->tile_rows = tile_rows;
    return RPT_OK;
}

int rpt_set_dispatch(rpt_ctx* ctx, uint32_t cost_order, uint32_t unit_rounds, uint32_t unit_min_spp, uint32_t unit_slots)
{
    if (!ctx || cost_order > 2u) { set_err(ctx, "rpt_set_dispatch: invalid argument"); return RPT_ERR_INVALID_ARG; }
    ctx->dispatch[0] = cost_order; ctx->dispatch[1] = unit_rounds; ctx->dispatch[2] = unit_min_spp; ctx->dispatch[3] = unit_slots;
    return RPT_OK;
}

void rpt_destroy(rpt_ctx* ctx)
{
    if (!ctx) return;
    for (DevState& d : ctx->devs) { DeviceGuard guard(d.device); (void)hipStreamSynchronize(d.stream); (void)hipStreamSynchronize(d.comm_stream); }
    free_resident(ctx);
    RcclApi* api = ctx->use_comm ? rccl_api() : nullptr;
    for (DevState& d : ctx->devs) {
        if (d.comm && api) { DeviceGuard guard(d.device); (void)api->CommDestroy(d.comm); d.comm = nullptr; }
        free_dev(d);
    }
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
    for (uint32_t i = 0; i < s->n_lights; ++i)
        if (s->lights[i].type > RPT_LIGHT_DISTANT) { set_err(ctx, "rpt_upload_scene: light %u has an unknown type", i); return RPT_ERR_INVALID_ARG; }
    // participating media (include/rpt.h): used only under RPT_SCENE_MEDIA, and then only when some material carries one
    bool media = false;
    if (s->flags & RPT_SCENE_MEDIA) {
        for (uint32_t i = 0; i < s->n_materials; ++i) {
            const rpt_material& m = s->materials[i];
            if (!(m.mask & RPT_MAT_MEDIUM)) continue;
            if (m.medium_type > RPT_MEDIUM_EMISSIVE) { set_err(ctx, "rpt_upload_scene: material %u has an unknown medium type", i); return RPT_ERR_INVALID_ARG; }
            if (!(m.medium_density >= 0.0f) || !std::isfinite(m.medium_density)) {
                set_err(ctx, "rpt_upload_scene: material %u: the medium's density must be finite and >= 0", i);
                return RPT_ERR_INVALID_ARG;
            }
            media = media || m.medium_type != RPT_MEDIUM_NONE;
        }
        if (media && s->n_materials > kMaxMediaMaterials) { set_err(ctx, "rpt_upload_scene: scenes with media can have at most %u materials", kMaxMediaMaterials); return RPT_ERR_UNSUPPORTED; }
    }

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
            if (media && !(m.mask & RPT_MAT_MEDIUM)) {
                // (with patches a nearer sphere WITHOUT a medium would inherit the medium of a farther one accepted before it)
                set_err(ctx, "rpt_upload_scene: in a large scene with media every sphere material must set RPT_MAT_MEDIUM "
                             "(medium_type RPT_MEDIUM_NONE for none); sphere %u does not", i);
                return RPT_ERR_UNSUPPORTED;
            }
            if ((m.mask & RPT_MAT_ALL) != RPT_MAT_ALL || m.proc_kind != RPT_PROC_NONE) {
                set_err(ctx, "rpt_upload_scene: scenes beyond %d spheres / %d lights / %d materials need full sphere materials "
                             "(mask == RPT_MAT_ALL, no procedural part); sphere %u does not", kMaxSpheres, kMaxLights, kMaxMaterials, i);
                return RPT_ERR_UNSUPPORTED;
            }
            // the acceleration structure is built from these numbers: they must be numbers
            const rpt_sphere& sp = s->spheres[i];
            if (!std::isfinite(sp.center[0]) || !std::isfinite(sp.center[1]) || !std::isfinite(sp.center[2]) || !std::isfinite(sp.radius) || sp.radius < 0.0f) {
                set_err(ctx, "rpt_upload_scene: sphere %u has a non-finite centre or a negative / non-finite radius", i);
                return RPT_ERR_INVALID_ARG;
            }
        }
        const size_t sz_sph = sizeof(float4) * s->n_spheres;
        const size_t sz_smat = (sizeof(uint32_t) * s->n_spheres + 15) & ~(size_t)15;
        const size_t sz_lights = (sizeof(DevLight) * (s->n_lights ? s->n_lights : 1) + 15) & ~(size_t)15;
        const size_t sz_mats = (sizeof(DevMaterial) * (s->n_materials ? s->n_materials : 1) + 15) & ~(size_t)15;
        const bool use_accel = s->n_spheres >= 64 && !knobs().no_grid;
        HostAccel accel;
        if (use_accel) {
            std::string why;
            if (!build_accel(s->spheres, s->n_spheres, accel, why)) { set_err(ctx, "rpt_upload_scene: %s", why.c_str()); return RPT_ERR_UNSUPPORTED; }
        }
        // The spherical lights once more as {centre, radius * radius} records with their indices, padded to whole groups of four: what
        // Scene::sample_lights' loop streams (dev_scene_large.h, closest_geom_finish).  Only when every light that DOES something
        // in sample_lights is spherical: always, unless the scene samples the other light types and has a rectangular one.
        bool lights_fast = true;
        std::vector<float> lsph;
        std::vector<uint32_t> lids;
        for (uint32_t i = 0; i < s->n_lights; ++i) {
            const rpt_light& l = s->lights[i];
            if (l.type == RPT_LIGHT_SPHERICAL) { lsph.insert(lsph.end(), {l.position[0], l.position[1], l.position[2], l.radius * l.radius}); lids.push_back(i); }
            else if (l.type == RPT_LIGHT_RECTANGULAR && (s->flags & RPT_SCENE_SAMPLE_ALL_LIGHT_TYPES)) lights_fast = false;
        }
        const uint32_t n_light_spheres = (uint32_t)lids.size();
        while (lids.size() % 4u) { lsph.insert(lsph.end(), {0.0f, 0.0f, 0.0f, 0.0f}); lids.push_back(0u); }
        const size_t sz_lsph = sizeof(float) * lsph.size(), sz_lids = (sizeof(uint32_t) * lids.size() + 15) & ~(size_t)15;
        const size_t sz_tables = sz_sph + sz_smat + sz_lights + sz_mats + sz_lsph + sz_lids;
        const size_t sz_accel = accel.bytes();
        // every table is addressed with 32-bit byte offsets from its own base (dev_scene_large.h, gather32)
        if ((uint64_t)sz_tables + sz_accel >= (1ull << 32)) { set_err(ctx, "rpt_upload_scene: the scene's tables exceed 4 GiB"); return RPT_ERR_UNSUPPORTED; }
        std::vector<unsigned char> host(sz_tables + sz_accel, 0);
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
        const size_t off_lsph = sz_sph + sz_smat + sz_lights + sz_mats;
        if (!lsph.empty()) memcpy(host.data() + off_lsph, lsph.data(), sz_lsph);
        if (!lids.empty()) memcpy(host.data() + off_lsph + sz_lsph, lids.data(), sizeof(uint32_t) * lids.size());
        if (use_accel) accel.write(host.data() + sz_tables);
        for (DevState& d : ctx->devs) {
            DeviceGuard guard(d.device);
            RPT_HIP_CHECK(ctx, guard.status);
            RPT_HIP_CHECK(ctx, hipStreamSynchronize(d.stream));       // a running launch may still read the old tables
            if (d.tables) { RPT_HIP_CHECK(ctx, hipFree(d.tables)); d.tables = nullptr; }
            RPT_HIP_CHECK(ctx, hipMalloc(&d.tables, host.size()));
            RPT_HIP_CHECK(ctx, hipMemcpy(d.tables, host.data(), host.size(), hipMemcpyHostToDevice));
            unsigned char* base = reinterpret_cast<unsigned char*>(d.tables);
            SceneLarge& L = d.scene_large;
            memset(&L, 0, sizeof(L));
            L.n_spheres = s->n_spheres; L.n_planes = s->n_planes; L.n_lights = s->n_lights; L.n_materials = s->n_materials;
            L.flags = s->flags; L.max_depth = s->max_depth; L.eps = s->eps; L.n_lights_f = (float)s->n_lights;
            L.bg = dev_background(s->background);
            L.spheres = reinterpret_cast<const float4*>(base);
            L.sphere_material = reinterpret_cast<const uint32_t*>(base + sz_sph);
            L.lights = reinterpret_cast<const DevLight*>(base + sz_sph + sz_smat);
            L.materials = reinterpret_cast<const DevMaterial*>(base + sz_sph + sz_smat + sz_lights);
            L.light_spheres = reinterpret_cast<const float4*>(base + off_lsph);
            L.light_sphere_ids = reinterpret_cast<const uint32_t*>(base + off_lsph + sz_lsph);
            L.n_light_spheres = lights_fast ? n_light_spheres : 0xFFFFFFFFu;
            for (uint32_t i = 0; i < s->n_planes; ++i) L.planes[i] = dev_plane(s->planes[i]);
            L.use_accel = use_accel ? 1u : 0u;
            if (use_accel) accel.bind(L, base + sz_tables);
        }
        ctx->camera = s->camera;
        ctx->class_map_ok = false;
        ctx->large = true;
        ctx->media = media;
        ctx->has_scene = true;
        for (DevState& dv : ctx->devs) { dv.sched_launches = 0; for (DevState::SchedEntry& e : dv.sched_cache) e.launches = 0; }   // a new scene: the dispatch order is learned again
        return RPT_OK;
    }

    for (DevState& d : ctx->devs) {                                 // drop the previous scene's tables (a large scene's, or a small one's class map)
        if (!d.tables) continue;
        DeviceGuard guard(d.device);
        RPT_HIP_CHECK(ctx, guard.status);
        RPT_HIP_CHECK(ctx, hipStreamSynchronize(d.stream));
        RPT_HIP_CHECK(ctx, hipFree(d.tables)); d.tables = nullptr;
    }
    SceneSmallSdf& d = ctx->scene;
    memset(&d, 0, sizeof(d));
    d.n_spheres = s->n_spheres; d.n_planes = s->n_planes; d.n_lights = s->n_lights; d.n_materials = s->n_materials;
    d.flags = s->flags;
    d.max_depth = s->max_depth;
    d.eps = s->eps;
    d.n_lights_f = (float)s->n_lights;
    d.bg = dev_background(s->background);
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const rpt_sphere& a = s->spheres[i];
        d.spheres[i] = DevSphere{a.center[0], a.center[1], a.center[2], a.radius, a.material};
    }
    for (uint32_t i = 0; i < s->n_planes; ++i) d.planes[i] = dev_plane(s->planes[i]);
    for (uint32_t i = 0; i < s->n_lights; ++i) d.lights[i] = dev_light(s->lights[i]);
    for (uint32_t i = 0; i < s->n_materials; ++i) d.materials[i] = dev_material(s->materials[i]);
    d.sdf.n_prims = s->sdf.n_prims; d.sdf.max_steps = s->sdf.max_steps; d.sdf.material = s->sdf.material;
    d.sdf.smooth_k = s->sdf.smooth_k; d.sdf.hit_eps = s->sdf.hit_eps; d.sdf.max_t = s->sdf.max_t; d.sdf.normal_eps = s->sdf.normal_eps;
    d.sdf.inv_smooth_k = s->sdf.n_prims ? 1.0f / s->sdf.smooth_k : 0.0f;
    for (uint32_t i = 0; i < s->sdf.n_prims; ++i) {
        const rpt_sdf_prim& a = s->sdf.prims[i];
        d.sdf.prims[i] = DevSdfPrim{a.center[0], a.center[1], a.center[2], a.params[0], a.params[1], a.kind, {0u, 0u}};
    }
    // five to twelve primitives: the classes of accepted sets the material table is indexed by (launch.h, MatClassMap), once per scene;
    // the 4 096-byte map lives in each device's `tables`
    std::vector<uint8_t> cls;
    ctx->class_map_ok = s->sdf.n_prims == 0 && !media && material_class_map(static_cast<const SceneSmall&>(d), ctx->class_map, cls);
    if (ctx->class_map_ok) {
        for (DevState& dv : ctx->devs) {
            DeviceGuard guard(dv.device);
            RPT_HIP_CHECK(ctx, guard.status);
            RPT_HIP_CHECK(ctx, hipMalloc(&dv.tables, cls.size()));
            RPT_HIP_CHECK(ctx, hipMemcpy(dv.tables, cls.data(), cls.size(), hipMemcpyHostToDevice));
        }
    }
    ctx->camera = s->camera;
    ctx->large = false;
    ctx->media = media;
    ctx->has_scene = true;
    for (DevState& dv : ctx->devs) dv.sched_launches = 0;
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

uint32_t rpt_tile_rows_padded(uint32_t height, uint32_t tile_rows, uint32_t world)
{
    if (tile_rows == 0 || world == 0) return 0;
    return rows_padded_for(height, tile_rows, world);
}

int rpt_tile_copy_plan(uint32_t height, uint32_t tile_rows, uint32_t rank, uint32_t world, rpt_tile_plan* out)
{
    return tile_copy_plan(height, tile_rows, rank, world, out);
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
    RPT_ON_DEVICE(ctx);
    return launch_render(ctx, ctx->devs[0], pixels_dev, width, height, frames_done, spp, seed, flags, tile_rows, rank, world, (hipStream_t)stream);
}

int rpt_render(rpt_ctx* ctx, float* pixels, uint32_t width, uint32_t height, uint64_t frames_done, uint32_t spp, uint64_t seed,
               uint32_t flags)
{
    if (!ctx) { set_err(nullptr, "rpt_render: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels || width == 0 || height == 0) { set_err(ctx, "rpt_render: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if ((uint64_t)width * height > 0xFFFFFFFFull) { set_err(ctx, "rpt_render: image too large for 32-bit pixel indices"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_scene) { set_err(ctx, "rpt_render: no scene uploaded"); return RPT_ERR_NO_SCENE; }
    if ((size_t)ctx->world != ctx->devs.size()) {
        set_err(ctx, "rpt_render: a host ColorBuffer needs every rank in this process (rpt_create / rpt_create_multi); "
                     "with one process per GPU use the resident buffer (rpt_resident_*)");
        return RPT_ERR_UNSUPPORTED;
    }
    // The fan-out of tracer.rs:29-32: every device takes its rows of the caller's buffer (one strided copy each way,
    // each device over its own PCIe link) and all devices render concurrently, driven by one host thread.  Two things make
    // that true on a caller's pageable Vec<f32>: (1) with more than one device the buffer is page-locked for the duration
    // of the call (hipHostRegister), so every copy is a real asynchronous DMA — a copy to or from pageable memory returns
    // only when it is done, i.e. after that device's kernel; (2) the copies back are enqueued in a second pass, after
    // EVERY device has its upload and its launches, so even without the page lock (RPT_PIN_HOST=0, or a registration
    // that fails) no device waits for another one's kernel before it starts.
    DeviceGuard guard(ctx->devs[0].device);
    const uint32_t world = (uint32_t)ctx->world;
    const uint32_t tile_rows = world == 1 ? height : ctx->tile_rows;
    const uint32_t rows_padded = rows_padded_for(height, tile_rows, world);
    HostPin pin;
    if (ctx->devs.size() > 1 && knobs().pin_host) pin.lock(pixels, (size_t)width * height * 16u);
    const auto enqueue = [&]() -> int {
        for (DevState& d : ctx->devs) {
            RPT_HIP_CHECK(ctx, guard.to(d.device));
            int rc = ensure_fb(ctx, d, (size_t)rows_padded * width * 16u);
            if (rc != RPT_OK) return rc;
            RPT_HIP_CHECK(ctx, hipEventRecord(d.ev_begin, d.stream));
            RPT_HIP_CHECK(ctx, copy_rank_rows(true, pixels, d.fb, width, height, tile_rows, (uint32_t)d.rank, world, d.stream));
            rc = launch_render(ctx, d, d.fb, width, height, frames_done, spp, seed, flags, tile_rows, (uint32_t)d.rank, world, d.stream);
            if (rc != RPT_OK) return rc;
            RPT_HIP_CHECK(ctx, hipEventRecord(d.ev_end, d.stream));
        }
        ctx->timed = true;
        for (DevState& d : ctx->devs) {
            RPT_HIP_CHECK(ctx, guard.to(d.device));
            RPT_HIP_CHECK(ctx, copy_rank_rows(false, pixels, d.fb, width, height, tile_rows, (uint32_t)d.rank, world, d.stream));
        }
        return RPT_OK;
    };
    const int rc = enqueue();
    // wait for every device whatever happened: copies already enqueued still use the caller's (page-locked) buffer
    int rc_sync = RPT_OK;
    for (DevState& d : ctx->devs) {
        if (guard.to(d.device) != hipSuccess || hipStreamSynchronize(d.stream) != hipSuccess) {
            if (rc == RPT_OK && rc_sync == RPT_OK) set_err(ctx, "rpt_render: waiting for device %d failed: %s", d.device, hipGetErrorString(hipGetLastError()));
            rc_sync = RPT_ERR_HIP;
        }
    }
    if (rc == RPT_OK && rc_sync == RPT_OK)
        for (DevState& d : ctx->devs) {
            if (guard.to(d.device) != hipSuccess) return RPT_ERR_HIP;
            const int rc_h = check_handoffs(ctx, d);
            if (rc_h != RPT_OK) return rc_h;
        }
    return rc != RPT_OK ? rc : rc_sync;
}

#ifdef RPT_TEST_HOOKS    // include/rpt_test.h: the test build only (librpt_hip_test.so)
// Test / development probe (include/rpt_test.h): device 0's tile costs (4 per tile), dispatch order, development data.
int rpt_debug_sched_read(rpt_ctx* ctx, uint32_t* out, uint32_t capacity_tiles, uint32_t* n_tiles)
{
    if (!ctx || !out || !n_tiles) return RPT_ERR_INVALID_ARG;
    DevState& d = ctx->devs[0];
    *n_tiles = d.sched_tiles;
    if (!d.sched || d.sched_tiles > capacity_tiles) { set_err(ctx, "rpt_debug_sched_read: no launch yet, or %u tiles do not fit", d.sched_tiles); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    RPT_HIP_CHECK(ctx, hipDeviceSynchronize());
    RPT_HIP_CHECK(ctx, hipMemcpy(out, d.sched, (size_t)d.sched_tiles * 10u * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RPT_OK;
}

// (include/rpt_test.h) the last launch's KernelChoice on device 0
int rpt_debug_kernel_choice(rpt_ctx* ctx, uint32_t* out)
{
    if (!ctx || !out) return RPT_ERR_INVALID_ARG;
    *out = ctx->devs[0].last_choice;
    return RPT_OK;
}

// (include/rpt_test.h) read the environment's knobs again
int rpt_debug_reload_knobs(void) { rpthost::reload_knobs(); return RPT_OK; }

// Test probe (include/rpt_test.h): how long before device `a`'s last render ENDED device `b`'s began.
int rpt_debug_render_overlap_ms(rpt_ctx* ctx, int a, int b, float* ms)
{
    if (!ctx || !ms || a < 0 || b < 0 || (size_t)a >= ctx->devs.size() || (size_t)b >= ctx->devs.size()) {
        set_err(ctx, "rpt_debug_render_overlap_ms: invalid argument");
        return RPT_ERR_INVALID_ARG;
    }
    if (!ctx->timed) { set_err(ctx, "rpt_debug_render_overlap_ms: no render yet"); return RPT_ERR_INVALID_ARG; }
    DevState &da = ctx->devs[(size_t)a], &db = ctx->devs[(size_t)b];
    if (da.device != db.device) { set_err(ctx, "rpt_debug_render_overlap_ms: events of two physical devices cannot be compared"); return RPT_ERR_UNSUPPORTED; }
    DeviceGuard guard(da.device);
    RPT_HIP_CHECK(ctx, guard.status);
    RPT_HIP_CHECK(ctx, hipEventSynchronize(da.ev_end));
    RPT_HIP_CHECK(ctx, hipEventSynchronize(db.ev_begin));
    RPT_HIP_CHECK(ctx, hipEventElapsedTime(ms, db.ev_begin, da.ev_end));
    return RPT_OK;
}

#endif  // RPT_TEST_HOOKS

// ---- resident ColorBuffer ------------------------------------------------------------------------------------

int rpt_resident_reset(rpt_ctx* ctx)
{
    if (!ctx) { set_err(nullptr, "rpt_resident_reset: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    for (DevState& d : ctx->devs) {
        DeviceGuard guard(d.device);
        RPT_HIP_CHECK(ctx, guard.status);
        RPT_HIP_CHECK(ctx, hipStreamSynchronize(d.stream));
        RPT_HIP_CHECK(ctx, hipStreamSynchronize(d.comm_stream));
    }
    free_resident(ctx);
    return RPT_OK;
}

// ColorBuffer::new(width, height) (buffer.rs:18-26) as per-rank tiles
static int resident_begin(rpt_ctx* ctx, uint32_t width, uint32_t height)
{
    const uint32_t world = (uint32_t)ctx->world;
    const uint32_t tile_rows = ctx->plain() ? height : ctx->tile_rows;
    if (ctx->has_res && ctx->res_w == width && ctx->res_h == height && ctx->res_tile_rows == tile_rows) return RPT_OK;
    int rc = rpt_resident_reset(ctx);
    if (rc != RPT_OK) return rc;
    const uint32_t rows_padded = rows_padded_for(height, tile_rows, world);
    const size_t tile_bytes = (size_t)rows_padded * width * 16u;
    DeviceGuard guard(ctx->devs[0].device);
    for (DevState& d : ctx->devs) {
        RPT_HIP_CHECK(ctx, guard.to(d.device));
        RPT_HIP_CHECK(ctx, hipMalloc((void**)&d.tile, tile_bytes));
        RPT_HIP_CHECK(ctx, hipMemsetAsync(d.tile, 0, tile_bytes, d.stream));
    }
    ctx->res_w = width; ctx->res_h = height; ctx->res_tile_rows = tile_rows; ctx->res_rows_padded = rows_padded;
    ctx->res_frames = 0;
    ctx->has_res = true;
    return RPT_OK;
}

int rpt_resident_render(rpt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp, uint64_t seed, uint32_t flags)
{
    if (!ctx) { set_err(nullptr, "rpt_resident_render: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (width == 0 || height == 0) { set_err(ctx, "rpt_resident_render: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if ((uint64_t)width * height > 0xFFFFFFFFull) { set_err(ctx, "rpt_resident_render: image too large for 32-bit pixel indices"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_scene) { set_err(ctx, "rpt_resident_render: no scene uploaded"); return RPT_ERR_NO_SCENE; }
    int rc = resident_begin(ctx, width, height);
    if (rc != RPT_OK) return rc;
    DeviceGuard guard(ctx->devs[0].device);
    for (DevState& d : ctx->devs) {
        RPT_HIP_CHECK(ctx, guard.to(d.device));
        RPT_HIP_CHECK(ctx, hipEventRecord(d.ev_begin, d.stream));
        rc = launch_render(ctx, d, d.tile, width, height, ctx->res_frames, spp, seed, flags, ctx->res_tile_rows, (uint32_t)d.rank, (uint32_t)ctx->world, d.stream);
        if (rc != RPT_OK) return rc;
        RPT_HIP_CHECK(ctx, hipEventRecord(d.ev_end, d.stream));
    }
    ctx->timed = true;
    ctx->res_frames += spp;                                                      // tracer.rs:121
    return RPT_OK;
}

int rpt_resident_kernel_ms(rpt_ctx* ctx, float* ms)
{
    if (!ctx || !ms) { set_err(ctx, "rpt_resident_kernel_ms: NULL argument"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->timed) { set_err(ctx, "rpt_resident_kernel_ms: no rpt_resident_render yet"); return RPT_ERR_INVALID_ARG; }
    float worst = 0.0f;
    DeviceGuard guard(ctx->devs[0].device);
    for (DevState& d : ctx->devs) {
        RPT_HIP_CHECK(ctx, guard.to(d.device));
        RPT_HIP_CHECK(ctx, hipEventSynchronize(d.ev_end));
        float t = 0.0f;
        RPT_HIP_CHECK(ctx, hipEventElapsedTime(&t, d.ev_begin, d.ev_end));
        worst = t > worst ? t : worst;
    }
    *ms = worst;
    return RPT_OK;
}

int rpt_resident_frames(const rpt_ctx* ctx, uint64_t* frames)
{
    if (!ctx || !frames) return RPT_ERR_INVALID_ARG;
    *frames = ctx->res_frames;
    return RPT_OK;
}

int rpt_resident_upload(rpt_ctx* ctx, const float* pixels, uint32_t width, uint32_t height, uint64_t frames)
{
    if (!ctx) { set_err(nullptr, "rpt_resident_upload: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels || width == 0 || height == 0) { set_err(ctx, "rpt_resident_upload: invalid argument"); return RPT_ERR_INVALID_ARG; }
    int rc = resident_begin(ctx, width, height);
    if (rc != RPT_OK) return rc;
    DeviceGuard guard(ctx->devs[0].device);
    for (DevState& d : ctx->devs) {
        RPT_HIP_CHECK(ctx, guard.to(d.device));
        RPT_HIP_CHECK(ctx, copy_rank_rows(true, const_cast<float*>(pixels), d.tile, width, height, ctx->res_tile_rows, (uint32_t)d.rank, (uint32_t)ctx->world, d.stream));
        RPT_HIP_CHECK(ctx, hipStreamSynchronize(d.stream));          // the caller may free `pixels` on return
    }
    ctx->res_frames = frames;
    return RPT_OK;
}

// Tiles -> rank-major `gathered` on the root -> top-down image on the root, all enqueued (no host wait), and BESIDE the renders
// that follow: every rank copies its tile into a snapshot on its render stream (a device copy: 16.6 MB for configs[2]'s share,
// ~10 us) and everything else — the RCCL send / receive or the peer copy, the scatter kernel on the root — runs on the rank's
// second stream, `comm_stream`, behind an event.  The next render waits for nothing but that copy; gather k overlaps render k + 1
// and ranks no longer meet at every step (round 4).  The image is complete when the root's comm_stream has drained:
// rpt_resident_sync waits for both streams, the download paths wait for `gather_done` on the device.
// Returns the device pointer of the assembled image in *image_out (root only).
static int gather_to_root(rpt_ctx* ctx, float* image_dst, float** image_out)
{
    const uint32_t w = ctx->res_w, h = ctx->res_h, world = (uint32_t)ctx->world;
    DevState& root = ctx->devs[0];
    DeviceGuard guard(root.device);
    RPT_HIP_CHECK(ctx, guard.status);
    if (ctx->plain()) {                                              // the tile is the image
        if (image_dst && image_dst != root.tile) RPT_HIP_CHECK(ctx, hipMemcpyAsync(image_dst, root.tile, (size_t)w * h * 16u, hipMemcpyDeviceToDevice, root.stream));
        if (image_out) *image_out = image_dst ? image_dst : root.tile;
        return RPT_OK;
    }
    const size_t count = (size_t)ctx->res_rows_padded * w * 4u;      // floats per rank
    if (ctx->is_root()) {
        if (!ctx->gathered) RPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->gathered, count * 4u * world));
        if (!image_dst && !ctx->image) RPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->image, (size_t)w * h * 16u));
        if (!ctx->gather_done) RPT_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->gather_done, hipEventDisableTiming));
    }
    // 1. snapshots, on the render streams
    for (DevState& d : ctx->devs) {
        RPT_HIP_CHECK(ctx, guard.to(d.device));
        if (!d.snap) RPT_HIP_CHECK(ctx, hipMalloc((void**)&d.snap, count * 4u));
        if (d.snap_used) RPT_HIP_CHECK(ctx, hipStreamWaitEvent(d.stream, d.snap_free, 0));     // the previous gather still sends from it
        RPT_HIP_CHECK(ctx, hipMemcpyAsync(d.snap, d.tile, count * 4u, hipMemcpyDeviceToDevice, d.stream));
        RPT_HIP_CHECK(ctx, hipEventRecord(d.snap_ready, d.stream));
        RPT_HIP_CHECK(ctx, hipStreamWaitEvent(d.comm_stream, d.snap_ready, 0));
        d.snap_used = true;
    }
    // 2. the exchange, on the comm streams
    if (ctx->peer_gather) {
        // single process: peer copies over xGMI into the root's buffer, each on its source device's comm stream (which is ordered
        // behind the root's previous scatter by `gather_done`: that kernel still reads `gathered`)
        for (DevState& d : ctx->devs) {
            RPT_HIP_CHECK(ctx, guard.to(d.device));
            if (ctx->gather_issued) RPT_HIP_CHECK(ctx, hipStreamWaitEvent(d.comm_stream, ctx->gather_done, 0));
            RPT_HIP_CHECK(ctx, hipMemcpyPeerAsync(ctx->gathered + (size_t)d.rank * count, root.device, d.snap, d.device, count * 4u, d.comm_stream));
            RPT_HIP_CHECK(ctx, hipEventRecord(d.snap_free, d.comm_stream));
        }
        RPT_HIP_CHECK(ctx, guard.to(root.device));
        for (DevState& d : ctx->devs) RPT_HIP_CHECK(ctx, hipStreamWaitEvent(root.comm_stream, d.snap_free, 0));
    } else {
        // RCCL over xGMI: every other rank sends its snapshot to rank 0, which posts one receive per sender; one group, so the 7
        // incoming transfers use 7 links at once.  Rank 0's own tile is a device copy, not a send to itself.
        RcclApi* api = rccl_api();
        if (!api) { set_err(ctx, "gather: cannot load RCCL: %s", rccl_why()); return RPT_ERR_RCCL; }
        RPT_RCCL_CHECK(ctx, api, api->GroupStart());
        // From here to GroupEnd nothing returns: a group left open would swallow every later RCCL call of the process.
        ncclResult_t posted = ncclSuccess;
        const char* what = "";
        for (DevState& d : ctx->devs) {
            if (posted != ncclSuccess) break;
            if (d.rank == 0) {
                for (uint32_t r = 1; r < world && posted == ncclSuccess; ++r) {
                    posted = api->Recv(ctx->gathered + (size_t)r * count, count, ncclFloat, (int)r, d.comm, d.comm_stream);
                    what = "ncclRecv";
                }
            } else {
                posted = api->Send(d.snap, count, ncclFloat, 0, d.comm, d.comm_stream);
                what = "ncclSend";
            }
        }
        const ncclResult_t closed = api->GroupEnd();
        if (posted != ncclSuccess) { set_err(ctx, "gather: %s failed: %s", what, api->GetErrorString(posted)); return RPT_ERR_RCCL; }
        if (closed != ncclSuccess) { set_err(ctx, "gather: ncclGroupEnd failed: %s", api->GetErrorString(closed)); return RPT_ERR_RCCL; }
        if (ctx->is_root()) {
            RPT_HIP_CHECK(ctx, guard.to(root.device));
            RPT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->gathered, root.snap, count * 4u, hipMemcpyDeviceToDevice, root.comm_stream));
        }
        for (DevState& d : ctx->devs) {
            RPT_HIP_CHECK(ctx, guard.to(d.device));
            RPT_HIP_CHECK(ctx, hipEventRecord(d.snap_free, d.comm_stream));
        }
    }
    // 3. the scatter into the top-down image, on the root's comm stream
    if (ctx->is_root()) {
        RPT_HIP_CHECK(ctx, guard.to(root.device));
        float* img = image_dst ? image_dst : ctx->image;
        RPT_HIP_CHECK(ctx, rptlaunch::untile(ctx->gathered, img, w, h, ctx->res_tile_rows, world, ctx->res_rows_padded, root.comm_stream));
        RPT_HIP_CHECK(ctx, hipEventRecord(ctx->gather_done, root.comm_stream));
        ctx->gather_issued = true;
        if (image_out) *image_out = img;
    } else if (image_out) *image_out = nullptr;
    return RPT_OK;
}

int rpt_resident_gather_device(rpt_ctx* ctx, float* image_dev)
{
    if (!ctx) { set_err(nullptr, "rpt_resident_gather_device: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_res) { set_err(ctx, "rpt_resident_gather_device: no resident buffer"); return RPT_ERR_INVALID_ARG; }
    if (((uintptr_t)image_dev & 15u) != 0) { set_err(ctx, "rpt_resident_gather_device: image must be 16-byte aligned"); return RPT_ERR_INVALID_ARG; }
    return gather_to_root(ctx, image_dev, nullptr);
}

int rpt_resident_sync(rpt_ctx* ctx)
{
    if (!ctx) { set_err(nullptr, "rpt_resident_sync: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    DeviceGuard guard(ctx->devs[0].device);
    for (DevState& d : ctx->devs) {
        RPT_HIP_CHECK(ctx, guard.to(d.device));
        RPT_HIP_CHECK(ctx, hipStreamSynchronize(d.stream));
        RPT_HIP_CHECK(ctx, hipStreamSynchronize(d.comm_stream));    // (a gather in flight: gather_to_root)
        const int rc = check_handoffs(ctx, d);
        if (rc != RPT_OK) return rc;
    }
    return RPT_OK;
}

int rpt_host_pin(void* buffer, size_t bytes)
{
    if (!buffer || bytes == 0) { set_err(nullptr, "rpt_host_pin: invalid argument"); return RPT_ERR_INVALID_ARG; }
    const hipError_t e = hipHostRegister(buffer, bytes, hipHostRegisterPortable);
    if (e == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return RPT_OK; }
    if (e != hipSuccess) { (void)hipGetLastError(); set_err(nullptr, "rpt_host_pin: hipHostRegister failed: %s", hipGetErrorString(e)); return RPT_ERR_HIP; }
    return RPT_OK;
}

int rpt_host_unpin(void* buffer)
{
    if (!buffer) { set_err(nullptr, "rpt_host_unpin: invalid argument"); return RPT_ERR_INVALID_ARG; }
    const hipError_t e = hipHostUnregister(buffer);
    if (e != hipSuccess) { (void)hipGetLastError(); set_err(nullptr, "rpt_host_unpin: hipHostUnregister failed: %s", hipGetErrorString(e)); return RPT_ERR_HIP; }
    return RPT_OK;
}

// Is `p` page-locked host memory (hipHostMalloc / hipHostRegister)?  Then a copy to it is one DMA.
static bool host_is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// Device -> the caller's host buffer on the root's stream, then wait.  Page-locked destinations take the copy directly; a
// pageable one goes through the context's page-locked staging buffer (one DMA + one memcpy: the runtime's own staging of
// pageable copies runs at ~7 GB/s on this host, a third of that).
static int download_to_host(rpt_ctx* ctx, void* dst, const void* src_dev, size_t bytes)
{
    DevState& root = ctx->devs[0];
    if (host_is_pinned(dst)) {
        RPT_HIP_CHECK(ctx, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, root.stream));
        return rpt_resident_sync(ctx);
    }
    if (bytes > ctx->stage_bytes) {
        if (ctx->stage) { RPT_HIP_CHECK(ctx, hipHostFree(ctx->stage)); ctx->stage = nullptr; ctx->stage_bytes = 0; }
        RPT_HIP_CHECK(ctx, hipHostMalloc(&ctx->stage, bytes, hipHostMallocDefault));
        ctx->stage_bytes = bytes;
    }
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(ctx->stage, src_dev, bytes, hipMemcpyDeviceToHost, root.stream));
    const int rc = rpt_resident_sync(ctx);
    if (rc != RPT_OK) return rc;
    memcpy(dst, ctx->stage, bytes);
    return RPT_OK;
}

int rpt_resident_download(rpt_ctx* ctx, float* pixels)
{
    if (!ctx) { set_err(nullptr, "rpt_resident_download: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_res || (ctx->is_root() && !pixels)) { set_err(ctx, "rpt_resident_download: no resident buffer or NULL destination"); return RPT_ERR_INVALID_ARG; }
    float* img = nullptr;
    int rc = gather_to_root(ctx, nullptr, &img);
    if (rc != RPT_OK) return rc;
    DevState& root = ctx->devs[0];
    DeviceGuard guard(root.device);
    if (ctx->is_root() && !ctx->plain()) RPT_HIP_CHECK(ctx, hipStreamWaitEvent(root.stream, ctx->gather_done, 0));   // the image is assembled on the comm stream
    if (ctx->is_root()) return download_to_host(ctx, pixels, img, (size_t)ctx->res_w * ctx->res_h * 16u);
    return rpt_resident_sync(ctx);
}

int rpt_resident_download_u8(rpt_ctx* ctx, uint8_t* frame)
{
    if (!ctx) { set_err(nullptr, "rpt_resident_download_u8: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_res || (ctx->is_root() && !frame)) { set_err(ctx, "rpt_resident_download_u8: no resident buffer or NULL destination"); return RPT_ERR_INVALID_ARG; }
    float* img = nullptr;
    int rc = gather_to_root(ctx, nullptr, &img);
    if (rc != RPT_OK) return rc;
    DevState& root = ctx->devs[0];
    DeviceGuard guard(root.device);
    if (ctx->is_root()) {
        const size_t n = (size_t)ctx->res_w * ctx->res_h;
        if (!ctx->plain()) RPT_HIP_CHECK(ctx, hipStreamWaitEvent(root.stream, ctx->gather_done, 0));
        if (!ctx->frame_u8) RPT_HIP_CHECK(ctx, hipMalloc((void**)&ctx->frame_u8, n * 4u));
        RPT_HIP_CHECK(ctx, rptlaunch::convert_to_u8(img, ctx->frame_u8, n, root.stream));
        return download_to_host(ctx, frame, ctx->frame_u8, n * 4u);
    }
    return rpt_resident_sync(ctx);
}

int rpt_untile_device(rpt_ctx* ctx, const float* gathered_dev, float* image_dev, uint32_t width, uint32_t height,
                      uint32_t tile_rows, uint32_t world, uint32_t rows_padded, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_untile_device: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!gathered_dev || !image_dev || width == 0 || height == 0 || tile_rows == 0 || world == 0) { set_err(ctx, "rpt_untile_device: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if (rows_padded_for(height, tile_rows, world) > rows_padded) { set_err(ctx, "rpt_untile_device: rows_padded %u too small", rows_padded); return RPT_ERR_INVALID_ARG; }
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

int rpt_denoise_device(rpt_ctx* ctx, const float* pixels_dev, float* out_dev, uint32_t width, uint32_t height, uint32_t iterations,
                       float edge_k, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_denoise_device: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    const size_t bytes = (size_t)width * height * 16u;
    if (!pixels_dev || !out_dev || width == 0 || height == 0 || iterations < 1u || iterations > 6u || !(edge_k > 0.0f) || !std::isfinite(edge_k)) {
        set_err(ctx, "rpt_denoise_device: invalid argument (iterations 1..6, edge_k > 0)");
        return RPT_ERR_INVALID_ARG;
    }
    if ((((uintptr_t)pixels_dev | (uintptr_t)out_dev) & 15u) != 0) { set_err(ctx, "rpt_denoise_device: buffers must be 16-byte aligned"); return RPT_ERR_INVALID_ARG; }
    const char *a = (const char*)pixels_dev, *b = (const char*)out_dev;
    if (a < b + bytes && b < a + bytes) { set_err(ctx, "rpt_denoise_device: pixels_dev and out_dev overlap"); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    DevState& d = ctx->devs[0];
    if (iterations > 1u && bytes > d.dn_bytes) {
        if (d.dn) { RPT_HIP_CHECK(ctx, hipStreamSynchronize((hipStream_t)stream)); RPT_HIP_CHECK(ctx, hipFree(d.dn)); d.dn = nullptr; d.dn_bytes = 0; }
        RPT_HIP_CHECK(ctx, hipMalloc((void**)&d.dn, bytes));
        d.dn_bytes = bytes;
    }
    const bool uses_scratch = iterations > 1u;
    if (uses_scratch && d.dn_used && d.dn_stream != (hipStream_t)stream) RPT_HIP_CHECK(ctx, hipStreamWaitEvent((hipStream_t)stream, d.dn_done, 0));
    RPT_HIP_CHECK(ctx, rptlaunch::denoise(pixels_dev, out_dev, d.dn, width, height, iterations, edge_k, (hipStream_t)stream));
    if (uses_scratch) {
        RPT_HIP_CHECK(ctx, hipEventRecord(d.dn_done, (hipStream_t)stream));
        d.dn_stream = (hipStream_t)stream;
        d.dn_used = true;
    }
    return RPT_OK;
}

int rpt_denoise(rpt_ctx* ctx, const float* pixels, float* out, uint32_t width, uint32_t height, uint32_t iterations, float edge_k)
{
    if (!ctx) { set_err(nullptr, "rpt_denoise: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels || !out || width == 0 || height == 0) { set_err(ctx, "rpt_denoise: invalid argument"); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    DevState& d = ctx->devs[0];
    const size_t bytes = (size_t)width * height * 16u;
    int rc = ensure_fb(ctx, d, 2u * bytes);                           // input and output, one allocation
    if (rc != RPT_OK) return rc;
    float* out_dev = reinterpret_cast<float*>(reinterpret_cast<char*>(d.fb) + bytes);
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(d.fb, pixels, bytes, hipMemcpyHostToDevice, d.stream));
    rc = rpt_denoise_device(ctx, d.fb, out_dev, width, height, iterations, edge_k, d.stream);
    if (rc != RPT_OK) return rc;
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(out, out_dev, bytes, hipMemcpyDeviceToHost, d.stream));
    RPT_HIP_CHECK(ctx, hipStreamSynchronize(d.stream));
    return RPT_OK;
}

int rpt_convert_to_u8(rpt_ctx* ctx, const float* pixels, uint8_t* frame, uint32_t width, uint32_t height)
{
    if (!ctx) { set_err(nullptr, "rpt_convert_to_u8: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!pixels || !frame || width == 0 || height == 0) { set_err(ctx, "rpt_convert_to_u8: invalid argument"); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    DevState& d = ctx->devs[0];
    const size_t n = (size_t)width * height;
    int rc = ensure_fb(ctx, d, n * 16 + n * 4);                       // f32 RGBA in, u8 RGBA out, one allocation
    if (rc != RPT_OK) return rc;
    uint8_t* out_dev = reinterpret_cast<uint8_t*>(d.fb) + n * 16;
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(d.fb, pixels, n * 16, hipMemcpyHostToDevice, d.stream));
    rc = rpt_convert_to_u8_device(ctx, d.fb, out_dev, width, height, d.stream);
    if (rc != RPT_OK) return rc;
    RPT_HIP_CHECK(ctx, hipMemcpyAsync(frame, out_dev, n * 4, hipMemcpyDeviceToHost, d.stream));
    RPT_HIP_CHECK(ctx, hipStreamSynchronize(d.stream));
    return RPT_OK;
}

int rpt_synchronize(rpt_ctx* ctx, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_synchronize: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    RPT_ON_DEVICE(ctx);
    RPT_HIP_CHECK(ctx, hipStreamSynchronize((hipStream_t)stream));
    return check_handoffs_all(ctx);
}

#ifdef RPT_TEST_HOOKS    // include/rpt_test.h: the test build only
int rpt_probe_rays(rpt_ctx* ctx, const float* rays_dev, uint32_t* out_dev, uint64_t n, uint32_t use_grid, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_probe_rays: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!ctx->has_scene || !ctx->large) { set_err(ctx, "rpt_probe_rays: needs an uploaded large scene"); return RPT_ERR_NO_SCENE; }
    if (!rays_dev || !out_dev) { set_err(ctx, "rpt_probe_rays: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if (n == 0) return RPT_OK;
    RPT_ON_DEVICE(ctx);
    SceneLarge sc = ctx->devs[0].scene_large;
    if (!use_grid) sc.use_accel = 0;
    RPT_HIP_CHECK(ctx, rptlaunch::probe_rays(sc, rays_dev, out_dev, n, (hipStream_t)stream));
    return RPT_OK;
}

int rpt_probe_math(rpt_ctx* ctx, uint32_t fn, const float* a_dev, const float* b_dev, float* out_dev, uint64_t n, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_probe_math: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!a_dev || !b_dev || !out_dev || fn > RPT_PROBE_DIV3) { set_err(ctx, "rpt_probe_math: invalid argument"); return RPT_ERR_INVALID_ARG; }
    if (n == 0) return RPT_OK;
    RPT_ON_DEVICE(ctx);
    RPT_HIP_CHECK(ctx, rptlaunch::probe_math(fn, a_dev, b_dev, out_dev, n, (hipStream_t)stream));
    return RPT_OK;
}

int rpt_probe_fn(rpt_ctx* ctx, uint32_t fn, const float* in_dev, float* out_dev, uint64_t n, const float* params, void* stream)
{
    if (!ctx) { set_err(nullptr, "rpt_probe_fn: ctx is NULL"); return RPT_ERR_INVALID_ARG; }
    if (!in_dev || !out_dev || fn >= RPT_PROBE_FN_COUNT) { set_err(ctx, "rpt_probe_fn: invalid argument"); return RPT_ERR_INVALID_ARG; }
    DevCamera cam;
    memset(&cam, 0, sizeof(cam));
    if (fn == RPT_PROBE_FN_GEN_RAY) {
        if (!ctx->has_scene) { set_err(ctx, "rpt_probe_fn: GEN_RAY uses the uploaded scene's camera"); return RPT_ERR_NO_SCENE; }
        if (!params) { set_err(ctx, "rpt_probe_fn: GEN_RAY needs params = {width, height}"); return RPT_ERR_INVALID_ARG; }
        cam = make_camera(ctx->camera, params[0], params[1]);
    }
    if (n == 0) return RPT_OK;
    RPT_ON_DEVICE(ctx);
    RPT_HIP_CHECK(ctx, rptlaunch::probe_fn(fn, cam, in_dev, out_dev, n, (hipStream_t)stream));
    return RPT_OK;
}

#endif  // RPT_TEST_HOOKS

}  // extern "C"
#pragma GCC visibility pop

#ifdef RPT_PROFILE_BLOCKS
// Development build only (dev_prof.h, tools/block_profile.py): not part of include/rpt.h.  Every kernel class's object keeps its own
// counters; a profiled run uses one class, so their sum is that class's table.
extern "C" __attribute__((visibility("default"))) int rpt_prof_read(unsigned long long* out)
{
    unsigned long long part[rptdev::PB_COUNT * 3];
    for (uint32_t i = 0; i < rptdev::PB_COUNT * 3; ++i) out[i] = 0;
    hipError_t (*const readers[3])(unsigned long long*) = {rptlaunch::prof_read_small, rptlaunch::prof_read_sdf, rptlaunch::prof_read_large};
    for (auto rd : readers) {
        if (rd(part) != hipSuccess) return RPT_ERR_HIP;
        for (uint32_t i = 0; i < rptdev::PB_COUNT * 3; ++i) out[i] += part[i];
    }
    return RPT_OK;
}
#endif
