// knobs.h — every environment variable the library reads, in ONE struct read ONCE per process (the first time it is asked for).
// None of them changes a pixel: they decide which kernel instantiation runs, when and where work runs, or how the host side moves
// data — tuning runs and tests set them, callers need none.  (The library's own path comes from the loader; RPT_LIB is the Python
// package's, not the library's.)
#pragma once

#include <stdint.h>
#include <stdlib.h>

#include <string>

namespace rpthost {

struct Knobs {
    // -- scheduling inside a wave (lanes that must be waiting before the wave runs a block; 1..64)
    uint32_t shade_threshold = 56;       // RPT_SHADE_THRESHOLD      megakernels: SHADE runs when this many lanes wait for it
    uint32_t finish_threshold = 24;      // RPT_FINISH_THRESHOLD     small scenes' megakernel: ... FINISH (background, blend, next camera path)
    uint32_t sdf_march_min_lanes = 8;    // RPT_SDF_MARCH_MIN_LANES  SDF scenes: a wave keeps marching while this many lanes march
    uint32_t sdf_shade_room = 40;        // RPT_SDF_SHADE_ROOM       SDF scenes: surface hits wait in a room of their own until this many lanes do
    // -- dispatch of a launch (rpt_set_dispatch overrides per context)
    uint32_t dispatch_order = 1;         // RPT_DISPATCH_ORDER       0 bottom rows first; 1 most expensive tile first; 2 costs recorded, order kept (development)
    uint32_t unit_rounds = 12;           // RPT_UNIT_ROUNDS          launches of fewer rounds of workgroups are cut into chunks of samples
    uint32_t unit_min_spp = 64;          // RPT_UNIT_MIN_SPP         ... of at least this many samples
    bool dispatch_timeline = false;      // RPT_DISPATCH_TIMELINE    development: every wave leaves its start stamp (tools/dispatch_timeline.py)
    // -- which instantiation
    uint32_t compact_max_spp = 1;        // RPT_COMPACT_MAX_SPP      small scenes: launches of at most this many samples take the compacting kernel
    bool no_sized_kernels = false;       // RPT_NO_SIZED_KERNELS     never the kernels that know table sizes at compile time
    bool no_material_table = false;      // RPT_NO_MATERIAL_TABLE    never a hit's material from the workgroup's table
    uint32_t debug_extra_lds = 0;        // RPT_DEBUG_EXTRA_LDS      development: pad the headline kernel's LDS by this many bytes (occupancy experiments)
    // -- the grid of large scenes (host_grid.h)
    bool no_grid = false;                // RPT_NO_GRID              brute-force loops instead of the grid
    float grid_near_reach = 1.5f;        // RPT_GRID_NEAR_REACH      near tier's reach in half-diagonals of the grid box (0: no near tier)
    float grid_spheres_per_cell = 1.0f;  // RPT_GRID_SPHERES_PER_CELL
    bool grid_box_lists = false;         // RPT_GRID_BOX_LISTS       cell lists by box-cell overlap instead of ball-cell overlap (A/B)
    // -- host side
    std::string gather;                  // RPT_GATHER               "p2p": single-process multi-device contexts gather with peer copies instead of RCCL
    bool pin_host = true;                // RPT_PIN_HOST             0: rpt_render does not page-lock the caller's buffer for the call
    std::string rccl_lib;                // RPT_RCCL_LIB             load this instead of librccl.so.1 (tests: a name that cannot be loaded)
};

inline uint32_t knob_u32(const char* name, uint32_t dflt) { const char* e = getenv(name); return e ? (uint32_t)strtoul(e, nullptr, 10) : dflt; }
inline uint32_t knob_lanes(const char* name, uint32_t dflt) { const uint32_t v = knob_u32(name, dflt); return v < 1u ? 1u : (v > 64u ? 64u : v); }
inline bool knob_flag(const char* name, bool dflt) { const char* e = getenv(name); return e ? atoi(e) != 0 : dflt; }
inline float knob_f32(const char* name, float dflt) { const char* e = getenv(name); return e ? (float)atof(e) : dflt; }
inline std::string knob_str(const char* name) { const char* e = getenv(name); return e ? std::string(e) : std::string(); }

inline Knobs read_knobs()
{
    Knobs v;
    v.shade_threshold = knob_lanes("RPT_SHADE_THRESHOLD", v.shade_threshold);
    v.finish_threshold = knob_lanes("RPT_FINISH_THRESHOLD", v.finish_threshold);
    v.sdf_march_min_lanes = knob_lanes("RPT_SDF_MARCH_MIN_LANES", v.sdf_march_min_lanes);
    v.sdf_shade_room = knob_lanes("RPT_SDF_SHADE_ROOM", v.sdf_shade_room);
    v.dispatch_order = knob_u32("RPT_DISPATCH_ORDER", v.dispatch_order);
    v.unit_rounds = knob_u32("RPT_UNIT_ROUNDS", v.unit_rounds);
    v.unit_min_spp = knob_u32("RPT_UNIT_MIN_SPP", v.unit_min_spp);
    v.dispatch_timeline = getenv("RPT_DISPATCH_TIMELINE") != nullptr;
    v.compact_max_spp = knob_u32("RPT_COMPACT_MAX_SPP", v.compact_max_spp);
    v.no_sized_kernels = knob_flag("RPT_NO_SIZED_KERNELS", false);
    v.no_material_table = knob_flag("RPT_NO_MATERIAL_TABLE", false);
    v.debug_extra_lds = knob_u32("RPT_DEBUG_EXTRA_LDS", 0u);
    v.no_grid = getenv("RPT_NO_GRID") != nullptr;
    v.grid_near_reach = knob_f32("RPT_GRID_NEAR_REACH", v.grid_near_reach);
    v.grid_spheres_per_cell = knob_f32("RPT_GRID_SPHERES_PER_CELL", v.grid_spheres_per_cell);
    v.grid_box_lists = knob_flag("RPT_GRID_BOX_LISTS", false);
    v.gather = knob_str("RPT_GATHER");
    v.pin_host = knob_flag("RPT_PIN_HOST", true);
    v.rccl_lib = knob_str("RPT_RCCL_LIB");
    return v;
}

// The process's knobs: read from the environment the first time they are asked for.  (reload_knobs: the test build's
// rpt_debug_reload_knobs, for tests that change the environment between two scenes — never called by the product.)
inline Knobs& knobs_storage() { static Knobs k = read_knobs(); return k; }
inline const Knobs& knobs() { return knobs_storage(); }
inline void reload_knobs() { knobs_storage() = read_knobs(); }

}  // namespace rpthost
