// pcr_render — headless twin of the reference's src/main.cpp for the Huffman methods and the 10-10-10 method: create
// the renderer, the resource and its methods, select one by name, then run update()/render() frames.
//
//   pcr_render <file.huffman> [--method huffman_mem_iter_cuda|huffman_hqs|huffman_cuda] [--size WxH]
//   pcr_render <file.las>      --method loop_las_cuda                      [--size WxH]
//              [--camera yaw pitch radius tx ty tz] [--lod 0.1] [--cull 0|1] [--frames N]
//              [--async-load]   (.huffman: copies on the loader stream, frames draw what has arrived)
//              [--dump-fb fb.u64] [--dump-rgba out.ppm] [--dump-depth depth.exr]   (depth: huffman_hqs only, huffman_hqs.h:217-237)
// Prints one JSON line: batches, frames needed to load, ms of the last frame, FNV-1a of the u64 framebuffer.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "pcr_methods.hpp"

using namespace pcr_host;

static uint64_t fnv1a(const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: pcr_render <file.huffman> [options]\n"); return 2; }
    std::string path = argv[1], method = "huffman_mem_iter_cuda", dump_fb, dump_rgba, dump_depth;
    int w = 1920, h = 1080, frames = 0;
    bool async_load = false;
    // src/main.cpp:192-218 default setting ("morrobay" overview)
    double cam[6] = {-0.15, -0.57, 3166.32, 2239.05, 1713.63, -202.02};
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { std::fprintf(stderr, "%s needs %d value(s)\n", a.c_str(), n); std::exit(2); } };
        if (a == "--method") { need(1); method = argv[++i]; }
        else if (a == "--size") { need(1); if (std::sscanf(argv[++i], "%dx%d", &w, &h) != 2) return 2; }
        else if (a == "--camera") { need(6); for (int k = 0; k < 6; ++k) cam[k] = std::atof(argv[++i]); }
        else if (a == "--lod") { need(1); Debug::LOD = (float)std::atof(argv[++i]); }
        else if (a == "--cull") { need(1); Debug::frustumCullingEnabled = std::atoi(argv[++i]) != 0; }
        else if (a == "--frames") { need(1); frames = std::atoi(argv[++i]); }
        else if (a == "--async-load") { async_load = true; }
        else if (a == "--dump-fb") { need(1); dump_fb = argv[++i]; }
        else if (a == "--dump-rgba") { need(1); dump_rgba = argv[++i]; }
        else if (a == "--dump-depth") { need(1); dump_depth = argv[++i]; }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    try {
        Renderer renderer(w, h, 0);
        renderer.yaw = cam[0]; renderer.pitch = cam[1]; renderer.radius = cam[2];
        renderer.target[0] = cam[3]; renderer.target[1] = cam[4]; renderer.target[2] = cam[5];

        std::shared_ptr<HuffmanLasData> las_huffman;
        std::shared_ptr<ComputeLasData> las_compute;
        std::unique_ptr<Method> m0, m1, m2;
        if (method == "loop_las_cuda") {
            las_compute = ComputeLasData::create(path);                 // main.cpp:241 (commented out there)
            m0 = std::make_unique<ComputeLoopLasCUDA>(&renderer, las_compute);   // main.cpp:251 (commented out there)
            Runtime::addMethod(m0.get());
        } else {
            las_huffman = HuffmanLasData::create(path);                 // main.cpp:244
            las_huffman->asyncUpload = async_load;
            m0 = std::make_unique<HuffmanMemIter>(&renderer, las_huffman);   // main.cpp:266-267
            m1 = std::make_unique<HuffmanHQS>(&renderer, las_huffman);
            m2 = std::make_unique<ComputeHuffman>(&renderer, las_huffman);   // main.cpp:265 (commented out there)
            Runtime::addMethod(m0.get());                               // main.cpp:272-273
            Runtime::addMethod(m1.get());
            Runtime::addMethod(m2.get());
        }
        Runtime::setSelectedMethod(method);
        Method *selected = Runtime::getSelectedMethod();
        if (!selected) { std::fprintf(stderr, "no method named %s\n", method.c_str()); return 2; }

        int n = 0;
        double ms = 0;
        // the frame loop of Renderer::loop: update() then render() (main.cpp:316-330, 433-436)
        while (true) {
            auto t0 = std::chrono::steady_clock::now();
            selected->update(&renderer);
            selected->render(&renderer);
            renderer.check(pcr_synchronize(renderer.ctx), "pcr_synchronize");
            ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            ++n;
            bool loaded = las_huffman ? las_huffman->fullyLoaded() && las_huffman->fullyResident(&renderer) : las_compute->fullyLoaded();
            if (frames > 0 ? n >= frames && loaded : loaded) break;
            if (n > 100000) throw std::runtime_error("loader made no progress");
        }
        if (!dump_depth.empty()) { Debug::saveDepthMap = true; Debug::depthMapPath = dump_depth; }
        selected->render(&renderer);   // one steady-state frame with everything resident
        std::vector<uint64_t> fb((size_t)w * h);
        renderer.check(pcr_read_framebuffer(renderer.ctx, fb.data(), fb.size()), "pcr_read_framebuffer");
        size_t covered = 0;
        for (uint64_t v : fb) covered += v != ~0ull;
        pcr_render_stats st;
        renderer.check(pcr_get_stats(renderer.ctx, &st), "pcr_get_stats");
        if (!dump_fb.empty()) std::ofstream(dump_fb, std::ios::binary).write((const char *)fb.data(), (std::streamsize)(fb.size() * 8));
        std::vector<uint32_t> rgba((size_t)w * h);
        renderer.check(pcr_read_rgba(renderer.ctx, rgba.data(), rgba.size()), "pcr_read_rgba");
        if (!dump_rgba.empty()) {
            std::ofstream o(dump_rgba, std::ios::binary);
            o << "P6\n" << w << " " << h << "\n255\n";
            for (int y = h - 1; y >= 0; --y)          // GL convention: row 0 is the bottom
                for (int x = 0; x < w; ++x) { uint32_t c = rgba[(size_t)y * w + x]; char px[3] = {(char)(c & 255), (char)((c >> 8) & 255), (char)((c >> 16) & 255)}; o.write(px, 3); }
        }
        std::printf("{\"method\": \"%s\", \"batches\": %lld, \"frames_to_load\": %d, \"last_frame_ms\": %.3f, \"points_iterated\": %lld, "
                    "\"batches_culled\": %lld, \"covered_pixels\": %zu, \"fb_fnv1a\": \"%016llx\", \"rgba_fnv1a\": \"%016llx\"}\n",
                    selected->name.c_str(), (long long)(las_huffman ? las_huffman->numBatches : las_compute->numBatchesLoaded), n, ms, (long long)st.points_iterated,
                    (long long)st.batches_culled, covered, (unsigned long long)fnv1a(fb.data(), fb.size() * 8), (unsigned long long)fnv1a(rgba.data(), rgba.size() * 4));
        Runtime::resource->unload(&renderer);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pcr_render: %s\n", e.what());
        return 1;
    }
    return 0;
}
