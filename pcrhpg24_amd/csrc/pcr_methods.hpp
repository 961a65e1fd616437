// pcr_methods.hpp — C++ host-side adapters with the reference's plugin surface, over the C ABI (pcr_hip.h).
//
// A maintainer of the reference swaps the CUDA-driver bodies of these classes for the pcr_* calls below; the
// names, members and call order are the reference's:
//   Method / Runtime / Debug              include/Method.h:10-23, include/Runtime.h:15-55, include/Debug.h:14-31
//   Resource, HuffmanLasData              modules/compute/Resources.h:20-35, modules/compute/HuffmanLasLoader.{h,cpp}
//   HuffmanMemIter ("huffman_mem_iter_cuda")   modules/huffman_mem_iter_cuda/huffman_mem_iter_cuda.h
//   HuffmanHQS     ("huffman_hqs")              modules/huffman_hqs/huffman_hqs.h
//   ComputeLasData, ComputeLoopLasCUDA ("loop_las_cuda")   modules/compute/ComputeLasLoader.{h,cpp},
//                                                          modules/compute_loop_las_cuda/compute_loop_las_cuda.h
// Headless: `Renderer` carries the window size and the orbit camera only (no GLFW/GL/ImGui); the resolve target is
// a device RGBA8 buffer instead of a GL texture.
#pragma once

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pcr_encode.h"
#include "pcr_hip.h"
#include "pcr_las_reader.hpp"

namespace pcr_host {

struct Debug {                                  // include/Debug.h:21-30 (flags the Huffman methods read)
    inline static float LOD = 0.1f;
    inline static bool frustumCullingEnabled = true;
    inline static bool colorizeChunks = false;
    inline static bool showNumPoints = false;
    inline static bool saveDepthMap = false;            // include/Debug.h: one-shot, cleared after the dump
    inline static std::string depthMapPath = "depth.exr";   // the reference writes "out/depth.exr" (huffman_hqs.h:235)
};

// huffman_hqs.h:71-113 saves the depth map through tinyexr: one FLOAT channel "Z". Same file kind, written directly:
// OpenEXR 2 scanline file, no compression, increasing-Y line order.
inline bool saveSingleChannelEXR(const char *filename, const float *depthData, int width, int height)
{
    std::vector<uint8_t> f;
    auto raw = [&](const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; f.insert(f.end(), b, b + n); };
    auto i32 = [&](int32_t v) { raw(&v, 4); };
    auto str = [&](const char *z) { raw(z, std::strlen(z) + 1); };
    auto attr = [&](const char *name, const char *type, int32_t size) { str(name); str(type); i32(size); };
    const uint8_t magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
    raw(magic, 8);
    attr("channels", "chlist", 2 + 16 + 1);
    str("Z"); i32(2 /* FLOAT */); { const uint8_t plin[4] = {0, 0, 0, 0}; raw(plin, 4); } i32(1); i32(1);
    { const uint8_t end = 0; raw(&end, 1); }
    attr("compression", "compression", 1); { const uint8_t none = 0; raw(&none, 1); }
    const int32_t box[4] = {0, 0, width - 1, height - 1};
    attr("dataWindow", "box2i", 16); raw(box, 16);
    attr("displayWindow", "box2i", 16); raw(box, 16);
    attr("lineOrder", "lineOrder", 1); { const uint8_t incy = 0; raw(&incy, 1); }
    const float one = 1.0f, zero2[2] = {0.0f, 0.0f};
    attr("pixelAspectRatio", "float", 4); raw(&one, 4);
    attr("screenWindowCenter", "v2f", 8); raw(zero2, 8);
    attr("screenWindowWidth", "float", 4); raw(&one, 4);
    { const uint8_t end = 0; raw(&end, 1); }
    const uint64_t table = f.size(), line = 8 + (uint64_t)width * 4;
    for (int y = 0; y < height; ++y) { const uint64_t off = table + 8ull * height + line * y; raw(&off, 8); }
    for (int y = 0; y < height; ++y) { i32(y); i32(width * 4); raw(depthData + (size_t)y * width, (size_t)width * 4); }
    std::ofstream o(filename, std::ios::binary);
    o.write((const char *)f.data(), (std::streamsize)f.size());
    return (bool)o;
}


struct Renderer {
    int width = 1920, height = 1080;            // src/Renderer.cpp:142-143
    pcr_ctx *ctx = nullptr;
    double yaw = 0, pitch = 0, radius = 10, target[3] = {0, 0, 0};   // include/OrbitControls.h

    explicit Renderer(int w = 1920, int h = 1080, int device = 0) : width(w), height(h)
    {
        if (pcr_create(device, &ctx) != PCR_OK) throw std::runtime_error(std::string("pcr_create: ") + pcr_last_error(nullptr));
        check(pcr_set_image_size(ctx, w, h), "pcr_set_image_size");
    }
    ~Renderer() { pcr_destroy(ctx); }
    Renderer(const Renderer &) = delete;

    void check(int rc, const char *what) const
    {
        if (rc != PCR_OK) throw std::runtime_error(std::string(what) + ": " + pcr_last_error(ctx));
    }

    // ChangingRenderData as HuffmanHQS::render fills it (huffman_hqs.h:157-183)
    pcr_render_params params() const
    {
        pcr_render_params p;
        if (pcr_camera_orbit(yaw, pitch, radius, target, width, height, 60.0, 0.1, 200000.0, &p))
            throw std::runtime_error(std::string("pcr_camera_orbit: ") + pcr_host_last_error());
        p.lod_percent = (int)(Debug::LOD * 100);
        p.enable_frustum_culling = Debug::frustumCullingEnabled;
        p.colorize_chunks = Debug::colorizeChunks;
        p.show_num_points = Debug::showNumPoints;
        return p;
    }
};

// huffman_hqs.h:217-237: depth of every covered pixel as float, image flipped vertically, empty pixels 0
inline bool dumpDepthMap(Renderer *r, pcr_ctx *ctx, int width, int height, const std::string &path)
{
    std::vector<uint64_t> fb_host((size_t)width * height);
    r->check(pcr_read_framebuffer(ctx, fb_host.data(), fb_host.size()), "pcr_read_framebuffer");
    std::vector<float> depthmap((size_t)width * height, 0.0f);
    for (int i = 0; i < height; ++i)
        for (int j = 0; j < width; ++j) {
            const uint32_t value = (uint32_t)(fb_host[(size_t)i * width + j] >> 32);
            if (value == 0xFFFFFFFFu) continue;
            std::memcpy(&depthmap[(size_t)(height - i - 1) * width + j], &value, 4);
        }
    return saveSingleChannelEXR(path.c_str(), depthmap.data(), width, height);
}

enum ResourceState { UNLOADED, LOADING, LOADED, UNLOADING };   // Resources.h:20-25

struct Resource {
    ResourceState state = UNLOADED;
    virtual ~Resource() = default;
    virtual void load(Renderer *renderer) = 0;
    virtual void unload(Renderer *renderer) = 0;
    virtual void process(Renderer *renderer) = 0;
};

struct Method {
    std::string name = "no name", description = "", group = "no group";
    virtual ~Method() = default;
    virtual void update(Renderer *renderer) = 0;
    virtual void render(Renderer *renderer) = 0;
};

struct Runtime {
    inline static std::vector<Method *> methods;
    inline static Method *selectedMethod = nullptr;
    inline static Resource *resource = nullptr;
    static void addMethod(Method *m) { methods.push_back(m); }
    static void setSelectedMethod(const std::string &name)
    {
        for (Method *m : methods) if (m->name == name) selectedMethod = m;
    }
    static Method *getSelectedMethod() { return selectedMethod; }
};

// modules/compute/HuffmanLasLoader.{h,cpp}: header parse, progressive loading through a reader thread that hands
// over tasks of <= 100 batch records (cpp:81-149), uploadBatch on the render thread (cpp:176-313).
struct HuffmanLasData : Resource {
    struct LoaderTask {
        std::vector<std::vector<char>> buffers;
        std::vector<int64_t> batchIndices;
    };

    std::string path;
    int64_t numBatches = 0, numPoints = 0, encodedBytes = 0, separateBytes = 0, clusterBytes = 0;
    std::vector<int64_t> batch_data_sizes, batch_data_sizes_prefix;
    int64_t numBatchesLoaded = 0, numPointsLoaded = 0, offsetToBatchData = 0;
    int64_t numBatchesResident = 0;     // what the next frame draws; lags numBatchesLoaded only with asyncUpload
    bool asyncUpload = false;           // copies + transcode on the context's loader stream (pcr_set_async_upload)

    std::shared_ptr<LoaderTask> task;
    std::mutex mtx_state, mtx_tasks;
    std::thread reader;
    std::string readerError;            // set by the reader thread (under mtx_tasks) when it had to give up; process() throws it

    ~HuffmanLasData() override { stopReader(); }

    void loadHeader()                                       // HuffmanLasLoader.h:57-85
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("cannot open " + path);
        int64_t h[5];
        f.read((char *)h, sizeof h);
        if (!f) throw std::runtime_error(path + ": shorter than its header");
        numPoints = h[0]; numBatches = h[1]; encodedBytes = h[2]; separateBytes = h[3]; clusterBytes = h[4];
        if (numBatches <= 0 || numPoints != numBatches * PCR_POINTS_PER_BATCH) throw std::runtime_error(path + ": bad header");
        batch_data_sizes.resize((size_t)numBatches);
        f.read((char *)batch_data_sizes.data(), 8 * numBatches);
        if (!f) throw std::runtime_error(path + ": shorter than its batch size table");
        offsetToBatchData = 40 + 8 * numBatches;
        // every record at least its fixed part, and all of them inside the file (the reference trusts these, HuffmanLasLoader.h:
        // 72-84; a negative or huge entry would otherwise end in an allocation failure inside the reader thread)
        f.seekg(0, std::ios::end);
        const int64_t fileBytes = (int64_t)f.tellg();
        const int64_t minRecord = PCR_BATCH_FIXED_HEADER + 4 * (3072 + 1024 + 4096 + 4096 + 32) + PCR_COLOR_BYTES_PER_BATCH;
        batch_data_sizes_prefix = batch_data_sizes;
        int64_t running = 0;
        for (int64_t i = 0; i < numBatches; ++i) {
            const int64_t sz = batch_data_sizes[(size_t)i];
            if (sz < minRecord || sz > fileBytes) throw std::runtime_error(path + ": batch " + std::to_string(i) + " has record size " + std::to_string(sz));
            running += sz;
            if (offsetToBatchData + running > fileBytes) throw std::runtime_error(path + ": batch records exceed the file");
            batch_data_sizes_prefix[(size_t)i] = running;
        }
    }

    static std::shared_ptr<HuffmanLasData> create(const std::string &path)   // HuffmanLasLoader.h:87-92
    {
        auto d = std::make_shared<HuffmanLasData>();
        d->path = path;
        d->loadHeader();
        return d;
    }

    void load(Renderer *renderer) override                  // HuffmanLasLoader.cpp:22-150
    {
        {
            std::lock_guard<std::mutex> lock(mtx_state);
            if (state != UNLOADED) return;
            state = LOADING;
        }
        pcr_file_header hdr{numPoints, numBatches, encodedBytes, separateBytes, clusterBytes};
        renderer->check(pcr_stream_begin(renderer->ctx, &hdr, 0), "pcr_stream_begin");
        renderer->check(pcr_set_async_upload(renderer->ctx, asyncUpload ? 1 : 0), "pcr_set_async_upload");
        numBatchesLoaded = numPointsLoaded = numBatchesResident = 0;
        readerError.clear();
        reader = std::thread([this] {
          try {
            std::ifstream f(path, std::ios::binary);
            if (!f) throw std::runtime_error("cannot open " + path);
            int64_t remaining = numBatches, read = 0;
            while (remaining > 0) {
                {
                    std::lock_guard<std::mutex> lock(mtx_state);
                    if (state == UNLOADING) { state = UNLOADED; return; }
                }
                {
                    std::lock_guard<std::mutex> lock(mtx_tasks);
                    if (task) { std::this_thread::sleep_for(std::chrono::microseconds(100)); continue; }
                }
                auto t = std::make_shared<LoaderTask>();
                int64_t n = remaining < 100 ? remaining : 100;
                for (int64_t i = 0; i < n; ++i) {
                    int64_t b = read + i;
                    int64_t start = offsetToBatchData + (b ? batch_data_sizes_prefix[(size_t)b - 1] : 0);
                    std::vector<char> buf((size_t)batch_data_sizes[(size_t)b]);
                    f.seekg(start);
                    f.read(buf.data(), (std::streamsize)buf.size());
                    if (!f) throw std::runtime_error(path + ": short read on batch " + std::to_string(b));
                    t->buffers.push_back(std::move(buf));
                    t->batchIndices.push_back(b);
                }
                read += n; remaining -= n;
                std::lock_guard<std::mutex> lock(mtx_tasks);
                task = t;
            }
            std::lock_guard<std::mutex> lock(mtx_state);
            if (state == UNLOADING) state = UNLOADED;
            else if (state == LOADING) state = LOADED;
          } catch (const std::exception &e) {              // never let an exception leave the thread (std::terminate)
            { std::lock_guard<std::mutex> lock(mtx_tasks); readerError = e.what(); }
            std::lock_guard<std::mutex> lock(mtx_state);
            state = UNLOADED;
          }
        });
    }

    void process(Renderer *renderer) override               // HuffmanLasLoader.cpp:301-313
    {
        std::lock_guard<std::mutex> lock(mtx_tasks);
        if (!readerError.empty()) { const std::string e = readerError; readerError.clear(); throw std::runtime_error("loader: " + e); }
        if (!task) return;
        // the task is dropped whatever happens to it: a record the library rejects must not be offered again every frame
        struct Drop { std::shared_ptr<LoaderTask> &t; ~Drop() { t = nullptr; } } drop{task};
        std::vector<const void *> blobs;
        std::vector<size_t> sizes;
        for (size_t i = 0; i < task->batchIndices.size(); ++i) { blobs.push_back(task->buffers[i].data()); sizes.push_back(task->buffers[i].size()); }
        renderer->check(pcr_upload_batches(renderer->ctx, task->batchIndices.front(), (int64_t)blobs.size(), blobs.data(), sizes.data()),
                        "pcr_upload_batches");
        numBatchesLoaded = pcr_batches_loaded(renderer->ctx);
        numPointsLoaded = pcr_points_loaded(renderer->ctx);
    }

    void unload(Renderer *renderer) override                // HuffmanLasLoader.cpp:152-174
    {
        stopReader();
        numBatchesLoaded = 0;
        pcr_stream_unload(renderer->ctx);
        std::lock_guard<std::mutex> lock(mtx_state);
        state = UNLOADED;
    }

    bool fullyLoaded() const { return numBatchesLoaded == numBatches; }
    bool fullyResident(Renderer *renderer)
    {
        numBatchesResident = pcr_batches_resident(renderer->ctx);
        return numBatchesResident == numBatches;
    }

private:
    void stopReader()
    {
        {
            std::lock_guard<std::mutex> lock(mtx_state);
            if (state == LOADING) state = UNLOADING;
        }
        {
            std::lock_guard<std::mutex> lock(mtx_tasks);
            task = nullptr;
        }
        if (reader.joinable()) reader.join();
    }
};

struct HuffmanMethodBase : Method {
    std::shared_ptr<HuffmanLasData> las;
    Renderer *renderer;
    pcr_render_params lastParams{};
    HuffmanMethodBase(Renderer *r, std::shared_ptr<HuffmanLasData> l) : las(std::move(l)), renderer(r) { group = "none"; }

    void update(Renderer *r) override                      // huffman_hqs.h:116-124
    {
        if (Runtime::resource != (Resource *)las.get()) {
            if (Runtime::resource != nullptr) Runtime::resource->unload(r);
            las->load(r);
            Runtime::resource = (Resource *)las.get();
        }
    }
};

// One frame = CLEAR (of the previous frame) + RENDER + RESOLVE. The reference clears at the end of render()
// (huffman_mem_iter_cuda.h:250-252); headless callers read the result, so the clear opens the next frame instead.
struct HuffmanMemIter : HuffmanMethodBase {
    HuffmanMemIter(Renderer *r, std::shared_ptr<HuffmanLasData> l) : HuffmanMethodBase(r, std::move(l))
    {
        name = "huffman_mem_iter_cuda";
        description = "- Decodes Huffman Encoded values on the GPU";
    }
    void render(Renderer *r) override                      // huffman_mem_iter_cuda.h:122-254
    {
        las->process(r);
        if (las->numPointsLoaded == 0) return;
        lastParams = r->params();
        r->check(pcr_frame_begin(r->ctx, &lastParams, PCR_METHOD_BASIC), "pcr_frame_begin");   // CLEAR + cull/LOD prepass
        r->check(pcr_render_basic(r->ctx, &lastParams), "pcr_render_basic");
        r->check(pcr_resolve_basic(r->ctx, &lastParams), "pcr_resolve_basic");
    }
};

// modules/huffman_cuda/huffman_cuda.h:60-75: the reference's first Huffman method (class ComputeHuffman, registered as
// "huffman_cuda"; commented out in its main.cpp:19, 265 in favour of huffman_mem_iter_cuda, whose kernels are the same
// decode + {depth, BC1 colour} atomicMin with the loader's memory iteration added). The name north_star lists: the same frame
// as HuffmanMemIter under the reference's original method name.
struct ComputeHuffman : HuffmanMemIter {
    ComputeHuffman(Renderer *r, std::shared_ptr<HuffmanLasData> l) : HuffmanMemIter(r, std::move(l)) { name = "huffman_cuda"; }
};

struct HuffmanHQS : HuffmanMethodBase {
    HuffmanHQS(Renderer *r, std::shared_ptr<HuffmanLasData> l) : HuffmanMethodBase(r, std::move(l))
    {
        name = "huffman_hqs";
        description = "- Decodes Huffman Encoded values on the GPU";
    }
    void render(Renderer *r) override                      // huffman_hqs.h:126-273
    {
        las->process(r);
        if (las->numPointsLoaded == 0) return;
        lastParams = r->params();
        r->check(pcr_frame_begin(r->ctx, &lastParams, PCR_METHOD_HQS), "pcr_frame_begin");
        r->check(pcr_render_hqs_depth(r->ctx, &lastParams), "pcr_render_hqs_depth");
        r->check(pcr_render_hqs_color(r->ctx, &lastParams), "pcr_render_hqs_color");
        if (Debug::saveDepthMap) {                          // huffman_hqs.h:217-237
            dumpDepthMap(r, r->ctx, r->width, r->height, Debug::depthMapPath);
            Debug::saveDepthMap = false;
        }
        r->check(pcr_resolve_hqs(r->ctx, &lastParams), "pcr_resolve_hqs");
    }
};

// modules/compute/ComputeLasLoader.{h,cpp}: the LAS file is quantised task by task (<= 100 batches per frame, the
// reference's MAX_POINTS_PER_BATCH load buffer) into the three 10-10-10 levels. The reference quantises in a compute
// shader on upload (computeLasLoader.cs); here libpcr_host.so does it (pcr_las_quantize).
struct ComputeLasData : Resource {
    std::string path;
    LasPoints pts;
    int64_t numPoints = 0, numPointsLoaded = 0, numBatchesLoaded = 0;

    static std::shared_ptr<ComputeLasData> create(const std::string &path)      // ComputeLasLoader.h:97-103
    {
        auto d = std::make_shared<ComputeLasData>();
        d->path = path;
        std::string err;
        if (!read_las(path, d->pts, err)) throw std::runtime_error(err);
        d->numPoints = d->pts.numPoints;
        return d;
    }

    void load(Renderer *renderer) override                                       // ComputeLasLoader.cpp:14-38
    {
        if (state != UNLOADED) return;
        state = LOADING;
        renderer->check(pcr_las_begin(renderer->ctx, numPoints), "pcr_las_begin");
        numPointsLoaded = numBatchesLoaded = 0;
    }

    void process(Renderer *renderer) override                                    // ComputeLasLoader.cpp:140-262
    {
        if (state != LOADING) return;
        const int64_t first = numPointsLoaded, n = std::min<int64_t>(numPoints - first, PCR_DEFAULT_CHUNK_POINTS);
        const int64_t nb = (n + PCR_POINTS_PER_BATCH - 1) / PCR_POINTS_PER_BATCH;
        const size_t slots = (size_t)nb * PCR_POINTS_PER_BATCH;
        std::vector<pcr_xyz_batch> batches((size_t)nb);
        std::vector<uint32_t> xyz12(slots), xyz8(slots), xyz4(slots), rgba(slots);
        if (pcr_las_quantize(pts.x.data() + first, pts.y.data() + first, pts.z.data() + first, pts.color.data() + first, n,
                             &pts.las, batches.data(), xyz12.data(), xyz8.data(), xyz4.data(), rgba.data(), 0))
            throw std::runtime_error(std::string("pcr_las_quantize: ") + pcr_host_last_error());
        renderer->check(pcr_las_upload(renderer->ctx, numBatchesLoaded, nb, batches.data(), xyz12.data(), xyz8.data(),
                                       xyz4.data(), rgba.data()), "pcr_las_upload");
        numPointsLoaded += n;
        numBatchesLoaded = pcr_las_batches_loaded(renderer->ctx);
        if (numPointsLoaded == numPoints) state = LOADED;
    }

    void unload(Renderer *renderer) override                                     // ComputeLasLoader.cpp:114-131
    {
        numPointsLoaded = numBatchesLoaded = 0;
        pcr_las_unload(renderer->ctx);
        state = UNLOADED;
    }

    bool fullyLoaded() const { return numPointsLoaded == numPoints; }
};

struct ComputeLoopLasCUDA : Method {                                             // compute_loop_las_cuda.h:52-222
    std::shared_ptr<ComputeLasData> las;
    Renderer *renderer;
    pcr_render_params lastParams{};
    ComputeLoopLasCUDA(Renderer *r, std::shared_ptr<ComputeLasData> l) : las(std::move(l)), renderer(r)
    {
        name = "loop_las_cuda";
        description = "- Each thread renders X points.\n- Loads points from LAS file\n- Workgroup picks 4, 8, or 12 byte precision\n  depending on screen size of bounding box";
        group = "10-10-10 bit encoded";
    }
    void update(Renderer *r) override        // empty in the reference (:92-93); resource switch as huffman_hqs.h:116-124
    {
        if (Runtime::resource != (Resource *)las.get()) {
            if (Runtime::resource != nullptr) Runtime::resource->unload(r);
            las->load(r);
            Runtime::resource = (Resource *)las.get();
        }
    }
    void render(Renderer *r) override                                            // compute_loop_las_cuda.h:99-222
    {
        las->process(r);
        if (las->numPointsLoaded == 0) return;
        lastParams = r->params();
        r->check(pcr_clear(r->ctx), "pcr_clear");
        r->check(pcr_render_las(r->ctx, &lastParams), "pcr_render_las");
        r->check(pcr_resolve_las(r->ctx, &lastParams), "pcr_resolve_las");
    }
};

} // namespace pcr_host
