// host_scene.h — host-side preparation of the scene tables: the frame-invariant camera basis and the
// uniform grid of large scenes.  Plain C++ (compiled with -ffp-contract=off like everything else).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rpt.h"
#include "../../include/rpt_strict_math.h"
#include "dev_scene.h"
#include "dev_scene_large.h"
#include "host_grid.h"

namespace rpthost {

using rptdev::DevCamera;

// Frame-invariant part of Pinhole::gen_ray (camera/pinhole.rs:38-54), evaluated on the
// host with the reference's f32 operation order (this file is compiled with
// -ffp-contract=off) and the strict tan.
inline DevCamera make_camera(const rpt_camera& c, float width, float height)
{
    struct h3 { float x, y, z; };
    auto sub = [](h3 a, h3 b) { return h3{a.x - b.x, a.y - b.y, a.z - b.z}; };
    auto mulf = [](h3 a, float f) { return h3{a.x * f, a.y * f, a.z * f}; };
    auto cross = [](h3 a, h3 b) { return h3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };

    const float ratio = width / height;
    const float half_width = rpt_tanf((c.fov_deg * (3.14159265358979323846f / 180.0f)) * 0.5f);   // f32::to_radians
    const float half_height = half_width / ratio;
    const h3 origin{c.origin[0], c.origin[1], c.origin[2]};
    const h3 center{c.center[0], c.center[1], c.center[2]};
    const h3 up{0.0f, 1.0f, 0.0f};
    h3 w = sub(origin, center);
    const float wl = __builtin_sqrtf(w.x * w.x + w.y * w.y + w.z * w.z);
    w = h3{w.x / wl, w.y / wl, w.z / wl};
    const h3 u = cross(up, w);
    const h3 v = cross(w, u);
    const h3 lower_left = sub(sub(sub(origin, mulf(u, half_width)), mulf(v, half_height)), w);
    const h3 horizontal = mulf(u, half_width * 2.0f);
    const h3 vertical = mulf(v, half_height * 2.0f);
    const h3 rd = sub(lower_left, origin);

    DevCamera d;
    d.ox = origin.x; d.oy = origin.y; d.oz = origin.z;
    d.rdx = rd.x; d.rdy = rd.y; d.rdz = rd.z;
    d.hx = horizontal.x; d.hy = horizontal.y; d.hz = horizontal.z;
    d.vx = vertical.x; d.vy = vertical.y; d.vz = vertical.z;
    d.psx = 1.0f / width;
    d.psy = 1.0f / height;
    return d;
}

// The acceleration structure as capi.hip sees it: host_grid.h's data + the binding of its device copy into a SceneLarge.
struct HostAccel : HostAccelData {
    void bind(rptdev::SceneLarge& L, const unsigned char* base) const
    {
        for (int a = 0; a < 3; ++a) {
            L.gn[a] = grid.n[a]; L.gmin[a] = grid.gmin[a]; L.gmax[a] = grid.gmax[a];
            L.cell_size[a] = grid.cs[a]; L.inv_cell_size[a] = grid.inv_cs[a];
            L.gcenter[a] = grid.center[a];
        }
        L.safe_r2 = grid.safe_r2;
        L.near_r2 = grid.near_r2;
        L.near_cell_off = grid.near_off;
        L.cell_start = reinterpret_cast<const uint32_t*>(base);
        L.cell_items = reinterpret_cast<const uint32_t*>(base + sz_cstart);
        L.cell_spheres = reinterpret_cast<const float4*>(base + sz_cstart + sz_items);
        L.n_oversize = (uint32_t)grid.oversize.size();
        L.oversize = reinterpret_cast<const uint32_t*>(base + sz_cstart + sz_items + sz_cell_sph);
    }
};

}  // namespace rpthost
