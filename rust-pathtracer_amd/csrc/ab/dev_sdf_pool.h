// Workgroup-wide march pool for scenes with the SDF object (BASELINE.json configs[3]).
//
// In dev_sdf_path.h a lane marches its own ray, so a wave's march steps run with the lanes that happen to be
// marching (measured: 45 % of the wave, 43 % of the kernel's time) and the RESOLVE / SHADE blocks fire half empty
// because waiting for more lanes would idle the ones that wait.  A march, unlike a path, is a tiny self-contained
// job: origin, direction, t, the step count, where to stop (10 dwords in, t and a hit flag out).  So the marches of
// all four waves of a workgroup go through one queue in LDS:
//
//   * a lane that needs a march SUBMITS it (request record in LDS + its id pushed to a ring; the push is aggregated
//     per wave: ballot, prefix count, one atomic add by the first lane) and then only polls for the answer;
//   * a wave with nothing better to do SERVES the queue: every lane without a job takes one (one compare-and-swap
//     on the ring's head by the first lane for the whole wave), all lanes step their jobs together — whoever the
//     job belongs to — and a finished job's result goes back to its owner's record;
//   * so lanes that wait for their march, or for their block to fill, do the workgroup's march work meanwhile: the
//     march steps run with ~all lanes, and the blocks can afford to wait for more lanes.
//
// MEASURED SLOWER than the wave-private march (configs[3], 16 spp: 1.38-1.52 vs 2.18 Gsamples/s across the policies
// below).  The march steps do get fuller (52-60 % of the wave instead of 45 %), but (a) the blocks cannot: RESOLVE and
// SHADE work is pinned to its lane, a wave runs one block at a time, and paths alternate between the two, so their
// fills add up to <= 100 % whatever the marches do; (b) a wave that waits (for its block to fill, or for a batch of
// jobs worth serving) holds one of the SIMD's five wave slots, and this code loses 12 % at three active waves and
// 30 % at two; (c) the queue itself costs: atomics, polling, 168 B of scratch for the job registers.  Kept behind
// RPT_RENDER_SDF_POOL_MARCH for A/B; the default is the wave-private march.
//
// Arithmetic is untouched: a job is stepped with march_step() exactly like a lane's own march, in the same order,
// and the outcome feeds the same closest_hit / any_hit code, so images stay bit-identical
// (tests/test_gpu_parity.py, SDF cases run all three kernel forms).
#pragma once
#include "../dev_sdf_path.h"

namespace rptdev {

constexpr uint32_t kPoolRing = 1024;                                // 4 x the outstanding requests a workgroup can have
constexpr uint32_t kPoolEmpty = 0xFFFFFFFFu;

struct MarchPool {
    float4 a[256];             // request of lane i: origin.xyz, t (in: 0 or where a put-back job stands; out: the result)
    float4 b[256];             // direction.xyz, t_useful
    uint32_t c[256];           // steps << 2 | hit << 1 | done
    uint32_t ring[kPoolRing];  // ids of the requests waiting for a server; kPoolEmpty = free slot
    uint32_t head, tail;       // monotonically increasing; slot = counter % kPoolRing
};

RPT_DEV void pool_init(MarchPool& pool)
{
    for (uint32_t i = threadIdx.x; i < kPoolRing; i += 256u) pool.ring[i] = kPoolEmpty;
    pool.c[threadIdx.x] = 0u;
    if (threadIdx.x == 0) { pool.head = 0u; pool.tail = 0u; }
}

RPT_DEV uint32_t wave_rank(uint64_t mask)                            // number of set bits of `mask` below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Push request `id` for every ACTIVE lane (call under the condition that selects the submitting lanes).
RPT_DEV void pool_push(MarchPool& pool, uint32_t id)
{
    const uint64_t mask = __ballot(1);
    const uint32_t n = (uint32_t)__popcll(mask);
    const uint32_t rank = wave_rank(mask);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // the request record before its id
    uint32_t base = 0;
    if (rank == 0u) base = atomicAdd(&pool.tail, n);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    atomicExch(&pool.ring[(base + rank) % kPoolRing], id);
}

// Submit the march of (o, d) up to t_useful for lane `id` (= threadIdx.x) — active lanes only.
RPT_DEV void pool_submit(MarchPool& pool, uint32_t id, v3 o, v3 d, float t_useful)
{
    pool.a[id] = make_float4(o.x, o.y, o.z, 0.0f);
    pool.b[id] = make_float4(d.x, d.y, d.z, t_useful);
    pool.c[id] = 0u;
    pool_push(pool, id);
}

// A march a lane is stepping for somebody (possibly itself).
struct MarchJob {
    uint32_t owner;            // kPoolEmpty: no job
    v3 o;
    MarchRegs m;               // d, t, t_useful, steps, hit
};

// Every lane of the wave without a job tries to take one.  Converged call (all lanes of the wave that are still in
// the kernel).  `spin_cap`: bound on the wait for a producer that has reserved a ring slot but not written it yet.
RPT_DEV void pool_take(MarchPool& pool, MarchJob& job)
{
    const bool want = (job.owner == kPoolEmpty);
    const uint64_t wmask = __ballot(want);
    if (wmask == 0ull) return;
    const uint32_t n = (uint32_t)__popcll(wmask);
    const uint32_t rank = wave_rank(wmask);
    const uint32_t first = (uint32_t)(__ffsll((unsigned long long)__ballot(1)) - 1);
    uint32_t base = 0, take = 0;
    if (__lane_id() == first) {
        for (int tries = 0; tries < 16; ++tries) {
            const uint32_t h = atomicAdd(&pool.head, 0u);
            const uint32_t t = atomicAdd(&pool.tail, 0u);
            const uint32_t avail = t - h;
            const uint32_t k = n < avail ? n : avail;
            if (k == 0u || avail > kPoolRing) break;
            if (atomicCAS(&pool.head, h, h + k) == h) { base = h; take = k; break; }
        }
    }
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    take = (uint32_t)__builtin_amdgcn_readfirstlane((int)take);
    if (want && rank < take) {
        const uint32_t slot = (base + rank) % kPoolRing;
        uint32_t id = kPoolEmpty;
        for (int spin = 0; spin < (1 << 20) && id == kPoolEmpty; ++spin) id = atomicExch(&pool.ring[slot], kPoolEmpty);
        if (id != kPoolEmpty) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const volatile float4* pa = &pool.a[id];
            const volatile float4* pb = &pool.b[id];
            const uint32_t c = *(const volatile uint32_t*)&pool.c[id];
            job.owner = id;
            job.o = mk3(pa->x, pa->y, pa->z);
            job.m.t = pa->w;
            job.m.d = mk3(pb->x, pb->y, pb->z);
            job.m.t_useful = pb->w;
            job.m.steps = c >> 2;
            job.m.hit = false;
        }
    }
}

// One march step for every lane that holds a job; finished jobs report to their owners.
RPT_DEV void pool_step(MarchPool& pool, const DevSdf& sd, MarchJob& job)
{
    if (job.owner != kPoolEmpty) {
        RPT_PROF(PB_CLOSEST);                                          // (block profile: one march step of the wave)
        if (march_step(sd, job.o, job.m)) {
            ((volatile float4*)&pool.a[job.owner])->w = job.m.t;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the result before the done flag
            *(volatile uint32_t*)&pool.c[job.owner] = (job.m.steps << 2) | (job.m.hit ? 2u : 0u) | 1u;
            job.owner = kPoolEmpty;
        }
    }
}

// Before the wave runs a block of its own: unfinished jobs go back to the queue (only t and the step count changed).
RPT_DEV void pool_put_back(MarchPool& pool, MarchJob& job)
{
    if (job.owner != kPoolEmpty) {
        ((volatile float4*)&pool.a[job.owner])->w = job.m.t;
        *(volatile uint32_t*)&pool.c[job.owner] = job.m.steps << 2;
        pool_push(pool, job.owner);
        job.owner = kPoolEmpty;
    }
}

// Has lane `id`'s own request been answered?  (m.t / m.hit are set when it has.)
RPT_DEV bool pool_poll(MarchPool& pool, uint32_t id, MarchRegs& m)
{
    const uint32_t c = *(const volatile uint32_t*)&pool.c[id];
    if ((c & 1u) == 0u) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    m.t = ((const volatile float4*)&pool.a[id])->w;
    m.hit = (c & 2u) != 0u;
    return true;
}

// Requests waiting for a server (a snapshot: other waves push and take concurrently).
RPT_DEV uint32_t pool_avail(MarchPool& pool)
{
    const uint32_t h = *(const volatile uint32_t*)&pool.head;
    const uint32_t t = *(const volatile uint32_t*)&pool.tail;
    const uint32_t n = t - h;
    return n > 256u ? 0u : n;                                       // (head read before a concurrent take, tail after: transient)
}

// closest_hit march of the path's current ray: the analytic part now (it bounds the march), the march via the pool.
RPT_DEV void pool_begin_primary(MarchPool& pool, const SceneSmallSdf& sc, const PathRegs& p, MarchRegs& m)
{
    march_begin_primary(sc, p, m);                                  // m.t_useful, m.accepted
    pool_submit(pool, threadIdx.x, p.ray.o, m.d, m.t_useful);
}

// Shadow march of next-event estimation, as march_begin_shadow (dev_sdf_path.h) but the ray goes to the pool instead
// of borrowing p.ray.o.  True when a march was submitted.
RPT_DEV bool pool_begin_shadow(MarchPool& pool, const SceneSmallSdf& sc, const PathRegs& p, v3 fhp, v3 ffnormal, MarchRegs& m)
{
    m.hit = false;
    m.t = 0.0f;
    if (sc.n_lights == 0) return false;
    Rng rng = p.rng;
    v3 scatter_pos;
    float light_area;
    LightSample ls;
    if (!nee_sample(sc, fhp, ffnormal, rng, scatter_pos, light_area, ls)) return false;
    const float max_dist = ls.dist - sc.eps;
    const RayD shadow{scatter_pos, ls.direction};
    if (any_hit_analytic(sc, shadow, max_dist)) return false;      // occluded whatever the march says
    pool_submit(pool, threadIdx.x, scatter_pos, ls.direction, sdf_shadow_t_useful(sc, max_dist));
    return true;
}

}  // namespace rptdev
