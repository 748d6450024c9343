// examples/render_cpp.cpp — the reference's main loop (renderer/src/main.rs:36-42, :118-122) in C++
// over the C ABI: ColorBuffer 800x600, AnalyticalScene, Tracer, N x { render; convert_to_u8 }.
// Writes the raw f32 buffer and the u8 frame so a test can compare them with the oracle.
//   g++ -std=c++17 -I include examples/render_cpp.cpp -L rust-pathtracer_amd -lrpt_hip -o render_cpp
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rpt.hpp"

int main(int argc, char** argv)
{
    const size_t width = argc > 1 ? (size_t)atoi(argv[1]) : 800;
    const size_t height = argc > 2 ? (size_t)atoi(argv[2]) : 600;
    const int frames = argc > 3 ? atoi(argv[3]) : 4;
    const char* out = argc > 4 ? argv[4] : "render_cpp";
    try {
        rpt::ColorBuffer buffer(width, height);
        rpt::AnalyticalScene scene;
        rpt::Tracer pt(&scene);
        std::vector<uint8_t> frame(width * height * 4);
        for (int i = 0; i < frames; ++i) {
            pt.render(buffer);                       // main.rs:118
            pt.convert_to_u8(buffer, frame.data());  // main.rs:122
        }
        std::string base(out);
        FILE* f = fopen((base + ".f32").c_str(), "wb");
        fwrite(buffer.pixels.data(), sizeof(float), buffer.pixels.size(), f);
        fclose(f);
        f = fopen((base + ".u8").c_str(), "wb");
        fwrite(frame.data(), 1, frame.size(), f);
        fclose(f);
        printf("rendered %zux%zu, %zu frames\n", width, height, buffer.frames);
    } catch (const rpt::Error& e) {
        fprintf(stderr, "%s\n", e.what());
        return 2;
    }
    return 0;
}
