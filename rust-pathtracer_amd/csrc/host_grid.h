// host_grid.h — the uniform grid of large scenes, built on the host at upload.  Plain C++ with no HIP type in it (compiled with
// -ffp-contract=off like everything else): capi.hip includes it through host_scene.h, and tests/host_harness.cpp compiles it
// alone with g++ -fsanitize=address,undefined (oracle/Makefile, `make asan`).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rpt.h"
#include "knobs.h"

namespace rpthost {

// Uniform grid over the spheres of a large scene (dev_scene_large.h).  Cell size targets ~1 sphere
// per cell.  Every sphere is listed in each cell its PADDED ball reaches; the padding is the distance
// outside the sphere at which the reference's f32 ray/sphere test (d2 = l.l - tca^2 <= r^2: eleven
// roundings of magnitude |l|^2, < 6.6e-7 |l|^2; 1.2e-6 |l|^2 is used) can still report a hit, for the
// farthest ray origin the lists serve, plus 1e-3 cell sizes for the DDA's own rounding.
struct HostGrid {
    uint32_t n[3];
    float gmin[3], gmax[3], cs[3], inv_cs[3];
    float center[3], safe_r2;
    // Two tiers of lists over the same cells: [0, ncell] for ray origins within sqrt(safe_r2) of the centre, and — the padding grows
    // with the square of the farthest origin served — [near_off, near_off + ncell] with far shorter lists for origins within
    // sqrt(near_r2) (every bounce ray of a camera inside the scene).  near_r2 < 0: no near tier.
    float near_r2 = -1.0f;
    uint32_t near_off = 0;
    std::vector<uint32_t> cell_start, items;
    // Spheres far larger than the rest (the classic r = 1000 "ground sphere") would stretch the grid's box and be listed
    // in every cell: they stay out of the grid and every walk tests them up front, like sphere 0.  Ascending indices.
    std::vector<uint32_t> oversize;
};

// Which spheres stay out of the grid: radius beyond 8 x the median radius, at most kMaxOversize of them (more than that
// is a scene of generally mixed sizes, which the grid takes as it is).
constexpr size_t kMaxOversize = 32;
inline std::vector<uint32_t> pick_oversize(const rpt_sphere* sph, uint32_t count)
{
    std::vector<float> radii(count);
    for (uint32_t i = 0; i < count; ++i) radii[i] = sph[i].radius;
    std::nth_element(radii.begin(), radii.begin() + count / 2, radii.end());
    const double limit = 8.0 * (double)radii[count / 2];
    std::vector<uint32_t> big;
    for (uint32_t i = 0; i < count; ++i)
        if ((double)sph[i].radius > limit) big.push_back(i);
    if (big.size() > kMaxOversize || big.size() == count) big.clear();
    return big;
}

// false (with `why`): the cell lists would not fit 32-bit offsets (e.g. tens of thousands of large overlapping
// spheres, each listed in most cells).
inline bool build_grid(const rpt_sphere* sph, uint32_t count, double spheres_per_cell, HostGrid& g, std::string& why)
{
    g.oversize = pick_oversize(sph, count);
    std::vector<bool> skip(count, false);
    for (uint32_t i : g.oversize) skip[i] = true;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i = 0; i < count; ++i) {
        if (skip[i]) continue;
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], (double)sph[i].center[a] - sph[i].radius);
            hi[a] = std::max(hi[a], (double)sph[i].center[a] + sph[i].radius);
        }
    }
    // The box also holds every point at which the reference's test can still report a hit of one of these spheres (a line that
    // passes sqrt(r^2 + d2_err) from the centre, see below): a ray that never enters the box has nothing to find.
    {
        double hd = 0.0;
        for (int a = 0; a < 3; ++a) hd += 0.25 * (hi[a] - lo[a]) * (hi[a] - lo[a]);
        const double max_l = 7.0 * std::sqrt(hd) * 1.05 + 2.0;      // (the padded box's own half-diagonal is a little larger)
        for (uint32_t i = 0; i < count; ++i) {
            if (skip[i]) continue;
            const double r = sph[i].radius, reach = std::sqrt(r * r + 1.2e-6 * max_l * max_l);
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], (double)sph[i].center[a] - reach);
                hi[a] = std::max(hi[a], (double)sph[i].center[a] + reach);
            }
        }
    }
    double ext[3], vol = 1.0;
    for (int a = 0; a < 3; ++a) {
        double pad = 1e-3 * (hi[a] - lo[a]) + 1e-3;
        lo[a] -= pad; hi[a] += pad;
        ext[a] = hi[a] - lo[a];
        vol *= ext[a];
    }
    const double target = std::cbrt(vol / ((count - g.oversize.size()) / spheres_per_cell + 1.0));   // cell edge for ~spheres_per_cell spheres per cell
    for (int a = 0; a < 3; ++a) {
        double n = std::ceil(ext[a] / target);
        n = n < 1 ? 1 : (n > 128 ? 128 : n);
        g.n[a] = (uint32_t)n;
        g.gmin[a] = (float)lo[a];
        g.cs[a] = (float)(ext[a] / n);
        g.inv_cs[a] = 1.0f / g.cs[a];
        g.gmax[a] = g.gmin[a] + (float)g.n[a] * g.cs[a];
    }
    const size_t ncell = (size_t)g.n[0] * g.n[1] * g.n[2];
    double half_diag = 0.0;
    for (int a = 0; a < 3; ++a) {
        g.center[a] = (float)(0.5 * (lo[a] + hi[a]));
        half_diag += 0.25 * ext[a] * ext[a];
    }
    half_diag = std::sqrt(half_diag);
    const double safe_r = 6.0 * half_diag;
    g.safe_r2 = (float)(safe_r * safe_r);
    const double near_r = (double)knobs().grid_near_reach * half_diag;      // near tier's reach in half-diagonals (0: none)
    const int n_tiers = near_r > 0.0 && near_r < safe_r ? 2 : 1;
    g.cell_start.clear();
    g.items.clear();
    for (int tier = 0; tier < n_tiers; ++tier) {
        const double max_l = (tier == 0 ? safe_r : near_r) + half_diag;     // |sphere centre - ray origin| for the rays this tier serves
        const double d2_err = 1.2e-6 * max_l * max_l;                       // bound on the f32 error of l.l - tca*tca
        auto range = [&](const rpt_sphere& s, int a, int& c0, int& c1) {
            const double r = s.radius;
            const double pad = (std::sqrt(r * r + d2_err) - r) + 1e-3 * g.cs[a];
            c0 = (int)std::floor(((double)s.center[a] - s.radius - pad - g.gmin[a]) / g.cs[a]);
            c1 = (int)std::floor(((double)s.center[a] + s.radius + pad - g.gmin[a]) / g.cs[a]);
            c0 = std::max(0, std::min((int)g.n[a] - 1, c0));
            c1 = std::max(0, std::min((int)g.n[a] - 1, c1));
        };
        // Within that box of cells the sphere is listed where the padded BALL reaches the cell (grown by the DDA's allowance): a
        // reported hit point lies within sqrt(r^2 + d2_err) of the centre, so its cell is one of these.  (Border cells stand for
        // everything outside the grid on their side — the box above is clamped — so they are kept as the box has them.)
        auto touches = [&](const rpt_sphere& s, int x, int y, int z) {
            if (knobs().grid_box_lists) return true;
            const int c[3] = {x, y, z};
            double d2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                if (c[a] == 0 || c[a] == (int)g.n[a] - 1) continue;
                const double lo_a = (double)g.gmin[a] + (c[a] - 1e-3) * (double)g.cs[a], hi_a = (double)g.gmin[a] + (c[a] + 1.0 + 1e-3) * (double)g.cs[a];
                const double v = (double)s.center[a];
                const double d = v < lo_a ? lo_a - v : (v > hi_a ? v - hi_a : 0.0);
                d2 += d * d;
            }
            const double r = s.radius;
            return d2 <= (r * r + d2_err) * (1.0 + 1e-9);
        };
        std::vector<size_t> counts(ncell + 1, 0);                   // size_t: the total is checked before it becomes an offset
        for (uint32_t i = 0; i < count; ++i) {
            if (skip[i]) continue;
            int x0, x1, y0, y1, z0, z1;
            range(sph[i], 0, x0, x1); range(sph[i], 1, y0, y1); range(sph[i], 2, z0, z1);
            for (int z = z0; z <= z1; ++z)
                for (int y = y0; y <= y1; ++y)
                    for (int x = x0; x <= x1; ++x)
                        if (touches(sph[i], x, y, z)) counts[((size_t)z * g.n[1] + y) * g.n[0] + x + 1] += 1;
        }
        const size_t first = g.items.size();                        // this tier's entries follow the previous tier's
        counts[0] = first;
        for (size_t c = 0; c < ncell; ++c) counts[c + 1] += counts[c];      // counts -> exclusive prefix sums (absolute offsets)
        if (counts[ncell] > 0x7FFFFFFFull) {
            why = "the scene's spheres overlap too many grid cells (" + std::to_string(counts[ncell]) + " list entries; the limit is 2^31)";
            return false;
        }
        const size_t cs0 = g.cell_start.size();
        g.cell_start.insert(g.cell_start.end(), counts.begin(), counts.end());
        std::vector<uint32_t> cursor(g.cell_start.begin() + cs0, g.cell_start.end() - 1);
        g.items.resize(counts[ncell], 0);
        for (uint32_t i = 0; i < count; ++i) {                      // ascending sphere index within a cell
            if (skip[i]) continue;
            int x0, x1, y0, y1, z0, z1;
            range(sph[i], 0, x0, x1); range(sph[i], 1, y0, y1); range(sph[i], 2, z0, z1);
            for (int z = z0; z <= z1; ++z)
                for (int y = y0; y <= y1; ++y)
                    for (int x = x0; x <= x1; ++x)
                        if (touches(sph[i], x, y, z)) g.items[cursor[((size_t)z * g.n[1] + y) * g.n[0] + x]++] = i;
        }
        if (tier == 1) { g.near_r2 = (float)(near_r * near_r); g.near_off = (uint32_t)cs0; }
    }
    return true;
}

// The acceleration structure of a large scene, serialised: built once on the host, copied into the device table allocation
// after the plain tables (host_scene.h binds the pointers into each device's SceneLarge).
constexpr size_t kSpareListEntries = 8;

struct HostAccelData {
    HostGrid grid;
    std::vector<float> cell_spheres;          // {cx, cy, cz, r * r} of items[k] at k: a cell's spheres are one load away from its bounds
    size_t sz_cstart = 0, sz_items = 0, sz_cell_sph = 0, sz_oversize = 0;

    size_t bytes() const { return sz_cstart + sz_items + sz_cell_sph + sz_oversize; }
    void write(unsigned char* dst) const
    {
        // (an empty vector's data() may be null, which memcpy must not be handed even for 0 bytes: found by UBSan)
        auto put = [](unsigned char* to, const void* from, size_t n) { if (n) std::memcpy(to, from, n); };
        put(dst, grid.cell_start.data(), sizeof(uint32_t) * grid.cell_start.size());
        put(dst + sz_cstart, grid.items.data(), sizeof(uint32_t) * grid.items.size());
        put(dst + sz_cstart + sz_items, cell_spheres.data(), sizeof(float) * cell_spheres.size());
        put(dst + sz_cstart + sz_items + sz_cell_sph, grid.oversize.data(), sizeof(uint32_t) * grid.oversize.size());
    }
};

inline bool build_accel(const rpt_sphere* sph, uint32_t count, HostAccelData& a, std::string& why)
{
    if (!build_grid(sph, count, (double)knobs().grid_spheres_per_cell, a.grid, why)) return false;
    // (kSpare more entries than the lists hold: the walks read a list RPT_GRID_BATCH entries per trip without asking whether the
    // last trip's entries are all there — the answer is in the candidate test anyway — so the array must be readable a little past its end)
    a.cell_spheres.assign((a.grid.items.size() + kSpareListEntries) * 4, 0.0f);
    for (size_t k = 0; k < a.grid.items.size(); ++k) {
        const rpt_sphere& s = sph[a.grid.items[k]];
        a.cell_spheres[4 * k + 0] = s.center[0]; a.cell_spheres[4 * k + 1] = s.center[1];
        a.cell_spheres[4 * k + 2] = s.center[2]; a.cell_spheres[4 * k + 3] = s.radius * s.radius;   // (f32: the test's own radius2, analytical.rs:173)
    }
    a.sz_cstart = (sizeof(uint32_t) * a.grid.cell_start.size() + 15) & ~(size_t)15;
    a.sz_items = (sizeof(uint32_t) * a.grid.items.size() + 15) & ~(size_t)15;
    a.sz_cell_sph = sizeof(float) * a.cell_spheres.size();
    a.sz_oversize = (sizeof(uint32_t) * a.grid.oversize.size() + 15) & ~(size_t)15;
    return true;
}

}  // namespace rpthost
