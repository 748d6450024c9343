// host_harness.cpp — the host half of the product under sanitizers (test infrastructure; oracle/Makefile, `make asan`):
// csrc/host_grid.h's build_accel over fuzzed sphere sets, and csrc/tile_plan.h's tiling arithmetic against its row-by-row
// definition.  Both headers are plain C++; this file is compiled with g++ -fsanitize=address,undefined and must exit 0.
//     host_harness [histogram]      `histogram`: also print the cell-list length distribution of the 10 000-sphere layout
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../rust-pathtracer_amd/csrc/host_grid.h"
#include "../rust-pathtracer_amd/csrc/tile_plan.h"

static uint32_t g_state = 0x1234567u;
static uint32_t next_u32() { g_state = g_state * 747796405u + 2891336453u; uint32_t w = ((g_state >> ((g_state >> 28) + 4u)) ^ g_state) * 277803737u; return (w >> 22) ^ w; }
static float uni(float lo, float hi) { return lo + (hi - lo) * (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "host_harness: %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } } while (0)

static std::vector<rpt_sphere> make_spheres(uint32_t n, float ex, float ey, float ez, float rlo, float rhi, int ground)
{
    std::vector<rpt_sphere> s(n);
    for (uint32_t i = 0; i < n; ++i) {
        s[i].center[0] = uni(-ex, ex); s[i].center[1] = uni(0.0f, ey); s[i].center[2] = uni(-ez, 0.0f);
        s[i].radius = uni(rlo, rhi);
        s[i].material = 0;
    }
    if (ground >= 0 && n > 0) {                                      // the classic giant ground sphere, first or last
        rpt_sphere& g = s[ground ? n - 1 : 0];
        g.center[0] = 0.0f; g.center[1] = -1000.0f; g.center[2] = 0.0f; g.radius = 1000.0f;
    }
    return s;
}

static void check_accel(const std::vector<rpt_sphere>& sph, const rpthost::HostAccelData& a, const char* what)
{
    const rpthost::HostGrid& g = a.grid;
    const size_t ncell = (size_t)g.n[0] * g.n[1] * g.n[2];
    const int tiers = g.near_r2 >= 0.0f ? 2 : 1;
    CHECK(g.cell_start.size() == (ncell + 1) * (size_t)tiers, "%s: cell_start has %zu entries for %zu cells x %d tiers", what, g.cell_start.size(), ncell, tiers);
    CHECK(a.cell_spheres.size() == (g.items.size() + rpthost::kSpareListEntries) * 4, "%s: cell_spheres / items size", what);
    CHECK(a.sz_cell_sph == a.cell_spheres.size() * sizeof(float), "%s: the spare list entries are part of the device table", what);
    std::vector<char> oversize(sph.size(), 0);
    for (uint32_t i : g.oversize) { CHECK(i < sph.size(), "%s: oversize index", what); oversize[i] = 1; }
    for (int t = 0; t < tiers; ++t) {
        const size_t off = t == 0 ? 0 : g.near_off;
        std::vector<char> seen(sph.size(), 0);
        CHECK(off + ncell < g.cell_start.size(), "%s: tier offset", what);
        for (size_t c = 0; c < ncell; ++c) {
            const uint32_t k0 = g.cell_start[off + c], k1 = g.cell_start[off + c + 1];
            CHECK(k0 <= k1 && k1 <= g.items.size(), "%s: tier %d cell %zu bounds %u..%u of %zu", what, t, c, k0, k1, g.items.size());
            for (uint32_t k = k0; k < k1; ++k) {
                const uint32_t i = g.items[k];
                CHECK(i < sph.size() && !oversize[i], "%s: item %u", what, i);
                CHECK(k == k0 || g.items[k - 1] < i, "%s: cell lists ascend", what);
                seen[i] = 1;
                CHECK(a.cell_spheres[4 * k + 3] == sph[i].radius * sph[i].radius && a.cell_spheres[4 * k] == sph[i].center[0], "%s: cell_spheres[%u]", what, k);
                // the sphere's own centre cell must list it when the centre is inside the grid
            }
        }
        for (size_t i = 0; i < sph.size(); ++i) CHECK(seen[i] || oversize[i], "%s: tier %d: sphere %zu is in no cell", what, t, i);
    }
    // every sphere is listed in the cell that holds its centre (both tiers)
    for (size_t i = 0; i < sph.size(); ++i) {
        if (oversize[i]) continue;
        int c[3];
        for (int ax = 0; ax < 3; ++ax) {
            c[ax] = (int)std::floor((sph[i].center[ax] - g.gmin[ax]) * g.inv_cs[ax]);
            c[ax] = c[ax] < 0 ? 0 : (c[ax] > (int)g.n[ax] - 1 ? (int)g.n[ax] - 1 : c[ax]);
        }
        const size_t cell = ((size_t)c[2] * g.n[1] + c[1]) * g.n[0] + c[0];
        for (int t = 0; t < tiers; ++t) {
            const size_t off = t == 0 ? 0 : g.near_off;
            bool found = false;
            for (uint32_t k = g.cell_start[off + cell]; k < g.cell_start[off + cell + 1]; ++k) found = found || g.items[k] == i;
            CHECK(found, "%s: tier %d: sphere %zu is not in its centre's cell", what, t, i);
        }
    }
    std::vector<unsigned char> blob(a.bytes());
    a.write(blob.data());                                           // (ASan checks the serialisation's bounds)
}

int main(int argc, char** argv)
{
    // ---- grid builder over fuzzed layouts
    struct Case { uint32_t n; float ex, ey, ez, rlo, rhi; int ground; };
    const Case cases[] = {{64, 5, 3, 10, 0.1f, 0.5f, -1}, {200, 10, 4, 20, 0.2f, 0.8f, -1}, {300, 8, 2, 8, 0.05f, 1.5f, 0}, {1000, 30, 12, 60, 0.3f, 1.2f, 1},
                          {777, 1, 1, 1, 0.5f, 0.9f, -1}, {5000, 60, 12, 120, 0.3f, 1.2f, -1}, {128, 100, 0.1f, 100, 0.01f, 0.02f, -1}, {65, 0.001f, 0.001f, 0.001f, 0.0f, 0.0f, -1}};
    for (const Case& c : cases) {
        const std::vector<rpt_sphere> sph = make_spheres(c.n, c.ex, c.ey, c.ez, c.rlo, c.rhi, c.ground);
        rpthost::HostAccelData a;
        std::string why;
        CHECK(rpthost::build_accel(sph.data(), (uint32_t)sph.size(), a, why), "build_accel refused %u spheres: %s", c.n, why.c_str());
        char what[96];
        snprintf(what, sizeof(what), "%u spheres (ground %d)", c.n, c.ground);
        check_accel(sph, a, what);
    }
    if (argc > 1 && std::string(argv[1]) == "histogram") {
        g_state = 0x5EED0005u;
        const std::vector<rpt_sphere> sph = make_spheres(10000, 60, 12, 120, 0.3f, 1.2f, -1);
        rpthost::HostAccelData a;
        std::string why;
        CHECK(rpthost::build_accel(sph.data(), 10000, a, why), "%s", why.c_str());
        const rpthost::HostGrid& g = a.grid;
        const size_t ncell = (size_t)g.n[0] * g.n[1] * g.n[2];
        printf("grid %u x %u x %u = %zu cells, cell size %.2f %.2f %.2f\n", g.n[0], g.n[1], g.n[2], ncell, g.cs[0], g.cs[1], g.cs[2]);
        for (int t = 0; t < 2; ++t) {
            const size_t off = t == 0 ? 0 : g.near_off;
            std::vector<size_t> hist(24, 0);
            size_t total = 0;
            for (size_t c = 0; c < ncell; ++c) { const uint32_t n = g.cell_start[off + c + 1] - g.cell_start[off + c]; hist[n < 23 ? n : 23] += 1; total += n; }
            printf("tier %d: mean %.2f entries per cell; cells by list length:", t, (double)total / ncell);
            for (size_t n = 0; n < hist.size(); ++n) if (hist[n]) printf(" %zu:%.1f%%", n, 100.0 * hist[n] / ncell);
            printf("\n");
        }
    }
    // ---- tiling arithmetic against the row-by-row definition
    for (uint32_t world = 1; world <= 9; ++world)
        for (uint32_t tile_rows = 1; tile_rows <= 17; tile_rows += (tile_rows < 5 ? 1 : 4))
            for (uint32_t height = 1; height <= 130; height += (height < 40 ? 1 : 9)) {
                std::vector<int> owner(height, -1);
                uint32_t most = 0;
                for (uint32_t rank = 0; rank < world; ++rank) {
                    const uint32_t n = rptdev::tile_row_count(height, tile_rows, rank, world);
                    most = n > most ? n : most;
                    std::vector<int> local_of(height, -1);
                    for (uint32_t l = 0; l < n; ++l) {
                        const uint32_t grow = rptdev::tile_global_row(l, tile_rows, rank, world);
                        CHECK(grow < height && owner[grow] == -1, "tiling: height %u rows %u world %u: row %u of rank %u", height, tile_rows, world, grow, rank);
                        owner[grow] = (int)rank;
                        local_of[grow] = (int)l;
                    }
                    rpt_tile_plan p;
                    CHECK(rptdev::tile_copy_plan(height, tile_rows, rank, world, &p) == RPT_OK, "tile_copy_plan");
                    std::vector<int> planned(height, -1);
                    for (uint32_t b = 0; b < p.full_blocks; ++b)
                        for (uint32_t r = 0; r < p.block_rows; ++r) {
                            const uint32_t host_row = p.host_row0 + b * p.host_row_stride + r;
                            CHECK(host_row < height, "plan: host row %u of %u", host_row, height);
                            planned[host_row] = (int)(b * p.block_rows + r);
                        }
                    for (uint32_t r = 0; r < p.ragged_rows; ++r) { CHECK(p.ragged_host_row0 + r < height, "plan: ragged row"); planned[p.ragged_host_row0 + r] = (int)(p.ragged_tile_row0 + r); }
                    for (uint32_t g = 0; g < height; ++g) CHECK(planned[g] == local_of[g], "plan: height %u rows %u rank %u/%u: host row %u -> tile row %d, want %d", height, tile_rows, rank, world, g, planned[g], local_of[g]);
                }
                for (uint32_t g = 0; g < height; ++g) CHECK(owner[g] >= 0, "tiling: row %u has no owner", g);
                CHECK(rptdev::tile_rows_padded(height, tile_rows, world) == most, "tile_rows_padded");
            }
    rpt_tile_plan p;
    CHECK(rptdev::tile_copy_plan(0, 2, 0, 1, &p) == RPT_ERR_INVALID_ARG && rptdev::tile_copy_plan(8, 0, 0, 1, &p) == RPT_ERR_INVALID_ARG &&
          rptdev::tile_copy_plan(8, 2, 3, 3, &p) == RPT_ERR_INVALID_ARG && rptdev::tile_copy_plan(8, 2, 0, 1, nullptr) == RPT_ERR_INVALID_ARG, "argument checks");
    printf("host_harness: ok\n");
    return 0;
}
