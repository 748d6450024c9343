// dev_prof.h — block profiler for the megakernels (development tool, OFF in the product build).
//
// With -DRPT_PROFILE_BLOCKS (tools/block_profile.py builds librpt_hip_prof.so that way) every RPT_PROF(id)
// scope adds, once per wave and per execution of the scope: 1 to execs[id], the number of active lanes to
// lanes[id] and the s_memtime cycles spent inside to cycles[id].  lanes/(64*execs) is the block's lane
// utilisation, cycles its share of the wave's time (wall cycles of the wave: other waves of the SIMD issue in
// between, so shares are meaningful, absolute values are not).  Scopes go around call sites, so the closing
// stamp sits after the lanes reconverge.  Without the define the macro expands to nothing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rptdev {

enum ProfBlock : uint32_t {
    PB_TRACE = 0,      // the whole TRACE block of a scheduling pass
    PB_CLOSEST,        //   closest_hit + sample_lights
    PB_BACKGROUND,     //   miss: background()
    PB_FINALIZE,       //   State::finalize + emitter exit
    PB_FINISH,         //   blend + next camera path
    PB_SHADE,          // the whole SHADE block
    PB_FRAME,          //   make_frame
    PB_NEE_SAMPLE,     //   light pick + sample_light
    PB_ANYHIT,         //   shadow ray
    PB_EVAL,           //   disney_eval + MIS
    PB_SAMPLE_HEAD,    //   disney_sample, whole (the lobes below are inside it)
    PB_LOBE_DIFFUSE,
    PB_LOBE_CLEARCOAT,
    PB_LOBE_SPEC,
    PB_SAMPLE_TAIL,    //   to_world, throughput, next ray
    PB_PASS,           // one pass of the scheduling loop (votes included)
    PB_GRID_BEGIN,     // large scenes: DDA set-up of one grid walk
    PB_GRID_CELL,      // large scenes: one cell of a closest-hit walk (record tests + step + prefetch)
    PB_GRID_CELL_ANY,  // large scenes: one cell of a SHADOW walk (PB_GRID_CELL: of a closest-hit walk)
    PB_GRID_EXTRA,     //   a trip of a cell's list beyond its first batch (either walk)
    PB_GRID_RESOLVE,   //   the square-root half of a candidate (either walk)
    PB_WALK_HEAD,      // large scenes: sphere 0, oversize spheres, the grid-or-brute decision before a closest-hit walk
    PB_LIGHTS,         // large scenes: sample_lights' loop over the light spheres
    PB_ALIVE,          // one pass again, `lanes` = the lanes that still have samples to render, WEIGHTED by the pass's cycles (RPT_PROF_ALIVE)
    PB_COUNT
};

#ifdef RPT_PROFILE_BLOCKS
static __device__ unsigned long long g_prof[PB_COUNT * 3];
// Counters are kept per wave in LDS while the kernel runs (an LDS add by one lane costs a few cycles; global
// atomics per scope slow the kernel 30x) and flushed once per wave at the end.
__shared__ uint32_t s_prof[4 * PB_COUNT * 3];

__device__ __forceinline__ void prof_init()
{
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < PB_COUNT * 3; i += 64u) s_prof[wave * PB_COUNT * 3 + i] = 0u;
}
__device__ __forceinline__ void prof_flush()
{
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t ex = __ballot(1);
    const uint32_t first = (uint32_t)(__ffsll((unsigned long long)ex) - 1);
    if (__lane_id() == first)
        for (uint32_t i = 0; i < PB_COUNT * 3; ++i) atomicAdd(&g_prof[i], (unsigned long long)s_prof[wave * PB_COUNT * 3 + i]);
}

struct ProfScope {
    uint32_t id;
    uint32_t lanes;
    bool leader;
    uint64_t t0;
    bool weighted = false;
    __device__ __forceinline__ ProfScope(uint32_t i, uint32_t alive) : ProfScope(i) { lanes = alive; weighted = true; }
    __device__ __forceinline__ explicit ProfScope(uint32_t i) : id(i)
    {
        const uint64_t ex = __ballot(1);
        lanes = (uint32_t)__popcll(ex);
        leader = (__lane_id() == (uint32_t)(__ffsll((unsigned long long)ex) - 1));
        t0 = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ ~ProfScope()
    {
        const uint64_t t1 = __builtin_amdgcn_s_memtime();
        if (leader) {
            uint32_t* c = &s_prof[(threadIdx.x >> 6) * PB_COUNT * 3 + id * 3];
            atomicAdd(c + 0, 1u);
            atomicAdd(c + 1, weighted ? (lanes * (uint32_t)(t1 - t0)) >> 6 : lanes);      // weighted: lanes/64 x cycles, against cycles[id]
            atomicAdd(c + 2, (uint32_t)(t1 - t0));
        }
    }
};
// host: copy out and clear this translation unit's counters
inline hipError_t prof_read(unsigned long long* out)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * PB_COUNT * 3);
    if (e != hipSuccess) return e;
    static const unsigned long long zeros[PB_COUNT * 3] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zeros, sizeof(zeros));
}
#define RPT_PROF(id) ::rptdev::ProfScope rpt_prof_scope_##id(::rptdev::id)
#define RPT_PROF_ALIVE(alive) ::rptdev::ProfScope rpt_prof_scope_alive(::rptdev::PB_ALIVE, (alive))
#define RPT_PROF_INIT() ::rptdev::prof_init()
#define RPT_PROF_FLUSH() ::rptdev::prof_flush()
#else
#define RPT_PROF(id) do { } while (0)
#define RPT_PROF_ALIVE(alive) do { } while (0)
#define RPT_PROF_INIT() do { } while (0)
#define RPT_PROF_FLUSH() do { } while (0)
#endif

}  // namespace rptdev
