// pcr_render_dist — the headless frame loop of pcr_render over N GPUs of one node, in C++ over include/pcr_dist.h: one
// process, one pcr_ctx per GPU, the file's batches split into N contiguous shards (pcr_dist_shard_range), every frame =
// shard renders + ONE exchange step (RCCL ncclReduce / ncclAllReduce with ncclUint64 + ncclMin, or min -> colour -> sum for
// HQS) + resolve on rank 0. There is no reference counterpart (single-GPU viewer, src/main.cpp:61).
//
//   pcr_render_dist <file.huffman> --ranks N [--method huffman_mem_iter_cuda|huffman_hqs] [--size WxH]
//                   [--camera yaw pitch radius tx ty tz] [--lod 0.1] [--cull 0|1] [--frames K] [--allreduce]
//                   [--merge auto|reduce|sliced|sliced_p2p]
// --merge sliced: the frame is cut into N slices, reduce-scattered (sliced_p2p: all-to-all + local min), every rank resolves
// the slice it owns and the RGBA8 image is gathered on rank 0 (--allreduce: on every rank); auto = sliced from 64 MB frames on.
// Prints one JSON line: ranks, batches per rank, ms per frame, FNV-1a of rank 0's image (rgba_fnv1a) and -- for the
// whole-frame merges, after which rank 0 holds it -- of the merged u64 framebuffer (fb_fnv1a): the hashes pcr_render prints
// for the same file and camera on one GPU (min and + are associative: the merge is bit-exact).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "pcr_dist.h"
#include "pcr_encode.h"

static uint64_t fnv1a(const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

struct File {                                       // HuffmanLasData::loadHeader (HuffmanLasLoader.h:57-85), whole file in memory
    std::vector<uint8_t> bytes;
    int64_t num_batches = 0;
    std::vector<int64_t> offset;                    // record offsets, num_batches + 1
    static constexpr size_t FIXED = PCR_BATCH_FIXED_HEADER;
    // the part of a record in front of its two streams: header + start values + escape prefix + tables + cluster prefix
    static constexpr size_t SIDE = PCR_BATCH_FIXED_HEADER + 4 * (3072 + 1024 + 4096 + 4096 + 32);
    void stream_lengths(int64_t b, int32_t *ne, int32_t *ns) const       // (read_file checked that the record holds its side data)
    {
        const uint8_t *r = bytes.data() + offset[(size_t)b] + FIXED;
        std::memcpy(ns, r + 4 * 3072 + 4 * 1023, 4);
        std::memcpy(ne, r + 4 * (3072 + 1024 + 4096 + 4096) + 4 * 31, 4);
    }
    const uint8_t *streams(int64_t b) const { return bytes.data() + offset[(size_t)b] + FIXED + 4 * (3072 + 1024 + 4096 + 4096 + 32); }
};

static File read_file(const std::string &path)
{
    File f;
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) throw std::runtime_error("cannot open " + path);
    const std::streamsize n = in.tellg();
    in.seekg(0);
    f.bytes.resize((size_t)n);
    if (!in.read((char *)f.bytes.data(), n)) throw std::runtime_error("short read on " + path);
    if (n < 40) throw std::runtime_error("file shorter than its header");
    int64_t hdr[5];
    std::memcpy(hdr, f.bytes.data(), 40);
    f.num_batches = hdr[1];
    if (f.num_batches <= 0 || (size_t)n < 40 + 8 * (size_t)f.num_batches) throw std::runtime_error("bad batch count");
    f.offset.resize((size_t)f.num_batches + 1);
    int64_t off = 40 + 8 * f.num_batches;
    for (int64_t b = 0; b < f.num_batches; ++b) {
        int64_t sz;
        std::memcpy(&sz, f.bytes.data() + 40 + 8 * b, 8);
        // a record holds at least its side data and one colour array; its two stream lengths (read from inside it) have to
        // add up to its size -- checked here, before anything indexes into a record (the library validates the same again)
        if (sz < (int64_t)(File::SIDE + PCR_COLOR_BYTES_PER_BATCH) || off + sz > n)
            throw std::runtime_error("batch record " + std::to_string(b) + " is too short or exceeds the file");
        f.offset[(size_t)b] = off;
        int32_t ne, ns;
        f.stream_lengths(b, &ne, &ns);
        const int64_t streams = (int64_t)File::SIDE + 4 * ((int64_t)ne + (int64_t)ns);
        if (ne < 0 || ns < 0 || (sz != streams + PCR_COLOR_BYTES_PER_BATCH && sz != streams + PCR_COLOR_BYTES_PER_BATCH_BC7))
            throw std::runtime_error("batch record " + std::to_string(b) + ": size " + std::to_string(sz) + " does not match its stream lengths");
        off += sz;
    }
    f.offset[(size_t)f.num_batches] = off;
    return f;
}

#define CHECK(ctx, call) do { int rc_ = (call); if (rc_) throw std::runtime_error(std::string(#call) + ": " + pcr_last_error(ctx)); } while (0)
#define DCHECK(call) do { int rc_ = (call); if (rc_) throw std::runtime_error(std::string(#call) + ": " + pcr_dist_last_error()); } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: pcr_render_dist <file.huffman> --ranks N [options]\n"); return 2; }
    std::string path = argv[1], method = "huffman_mem_iter_cuda";
    int w = 1920, h = 1080, frames = 20, ranks = 1, cull = 1;
    bool allreduce = false;
    std::string merge = "auto";
    double lod = 0.1, cam[6] = {-0.15, -0.57, 3166.32, 2239.05, 1713.63, -202.02};
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { std::fprintf(stderr, "%s needs %d value(s)\n", a.c_str(), n); std::exit(2); } };
        if (a == "--method") { need(1); method = argv[++i]; }
        else if (a == "--ranks") { need(1); ranks = std::atoi(argv[++i]); }
        else if (a == "--size") { need(1); if (std::sscanf(argv[++i], "%dx%d", &w, &h) != 2) return 2; }
        else if (a == "--camera") { need(6); for (int k = 0; k < 6; ++k) cam[k] = std::atof(argv[++i]); }
        else if (a == "--lod") { need(1); lod = std::atof(argv[++i]); }
        else if (a == "--cull") { need(1); cull = std::atoi(argv[++i]); }
        else if (a == "--frames") { need(1); frames = std::atoi(argv[++i]); }
        else if (a == "--allreduce") { allreduce = true; }
        else if (a == "--merge") { need(1); merge = argv[++i]; }
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    const bool hqs = method == "huffman_hqs";
    if (!hqs && method != "huffman_mem_iter_cuda" && method != "huffman_cuda") { std::fprintf(stderr, "no method named %s\n", method.c_str()); return 2; }
    if (ranks < 1) return 2;
    const int exchange_arg = merge == "auto" ? PCR_DIST_EXCHANGE_AUTO : merge == "reduce" ? PCR_DIST_EXCHANGE_REDUCE :
                             merge == "sliced" ? PCR_DIST_EXCHANGE_SLICED : merge == "sliced_p2p" ? PCR_DIST_EXCHANGE_SLICED_P2P : -1;
    if (exchange_arg < 0) { std::fprintf(stderr, "--merge auto|reduce|sliced|sliced_p2p\n"); return 2; }
    std::vector<pcr_ctx *> ctx((size_t)ranks, nullptr);
    std::vector<pcr_dist *> dist((size_t)ranks, nullptr);
    int status = 0;
    try {
        const File f = read_file(path);
        pcr_render_params p;
        const double target[3] = {cam[3], cam[4], cam[5]};
        if (pcr_camera_orbit(cam[0], cam[1], cam[2], target, w, h, 60.0, 0.1, 200000.0, &p)) throw std::runtime_error(pcr_host_last_error());
        p.lod_percent = (int)(lod * 100.0);                                 // huffman_hqs.h:177
        p.enable_frustum_culling = cull;
        std::vector<int64_t> first((size_t)ranks), count((size_t)ranks);
        for (int r = 0; r < ranks; ++r) {
            pcr_dist_shard_range(f.num_batches, ranks, r, &first[(size_t)r], &count[(size_t)r]);
            if (count[(size_t)r] == 0) throw std::runtime_error("more ranks than batches");
            if (pcr_create(r, &ctx[(size_t)r])) throw std::runtime_error(std::string("pcr_create(") + std::to_string(r) + "): " + pcr_last_error(nullptr));
            pcr_ctx *c = ctx[(size_t)r];
            CHECK(c, pcr_set_image_size(c, w, h));
            pcr_file_header hd{};
            hd.num_batches = count[(size_t)r]; hd.num_points = count[(size_t)r] * PCR_POINTS_PER_BATCH; hd.cluster_bytes = 128 * count[(size_t)r];
            for (int64_t b = first[(size_t)r]; b < first[(size_t)r] + count[(size_t)r]; ++b) {
                int32_t ne, ns; f.stream_lengths(b, &ne, &ns);
                hd.encoded_bytes += 4 * (int64_t)ne; hd.separate_bytes += 4 * (int64_t)ns;
            }
            CHECK(c, pcr_stream_begin(c, &hd, first[(size_t)r]));
            for (int64_t b0 = 0; b0 < count[(size_t)r]; b0 += 100) {        // loader tasks of <= 100 records (HuffmanLasLoader.cpp:106)
                const int64_t n = std::min<int64_t>(100, count[(size_t)r] - b0);
                std::vector<const void *> blobs((size_t)n); std::vector<size_t> sizes((size_t)n);
                for (int64_t k = 0; k < n; ++k) {
                    const int64_t b = first[(size_t)r] + b0 + k;
                    blobs[(size_t)k] = f.bytes.data() + f.offset[(size_t)b]; sizes[(size_t)k] = (size_t)(f.offset[(size_t)b + 1] - f.offset[(size_t)b]);
                }
                CHECK(c, pcr_upload_batches(c, b0, n, blobs.data(), sizes.data()));
            }
            const int64_t nxt = first[(size_t)r] + count[(size_t)r];       // shard boundary: the words that follow in the file
            if (nxt < f.num_batches) {
                int32_t ne, ns; f.stream_lengths(nxt, &ne, &ns);
                const uint8_t *s = f.streams(nxt);
                CHECK(c, pcr_upload_tail(c, (const uint32_t *)s, (size_t)std::min<int32_t>(ne, PCR_ENCODED_PAD_WORDS),
                                         (const int32_t *)(s + 4 * (size_t)ne), (size_t)std::min<int32_t>(ns, PCR_SEPARATE_PAD_WORDS)));
            }
        }
        if (ranks == 1) {
            unsigned char id[PCR_DIST_ID_BYTES];
            DCHECK(pcr_dist_unique_id(id));
            DCHECK(pcr_dist_create(ctx[0], id, 0, 1, &dist[0]));
        } else {
            DCHECK(pcr_dist_create_local(ctx.data(), ranks, dist.data()));
        }
        const int root = allreduce ? PCR_DIST_ALL : 0;
        for (int r = 0; r < ranks; ++r) DCHECK(pcr_dist_set_exchange(dist[(size_t)r], exchange_arg));
        const int exchange = pcr_dist_exchange(dist[0]);
        const bool sliced = exchange != PCR_DIST_EXCHANGE_REDUCE;
        const size_t pixels = (size_t)w * h;
        // the image assembled from the slices the ranks resolved (sliced exchange): one group of gathers
        auto gather = [&]() {
            DCHECK(pcr_dist_group_begin());
            for (int r = 0; r < ranks; ++r) DCHECK(pcr_dist_gather_image(dist[(size_t)r], root));
            DCHECK(pcr_dist_group_end());
        };
        auto my_slice = [&](int r, size_t *first, size_t *px) {
            size_t cnt;
            pcr_dist_slice_range(pcr_framebuffer_elems(ctx[(size_t)r]), ranks, r, first, &cnt);
            *px = *first < pixels ? std::min(cnt, pixels - *first) : 0;
        };
        auto frame = [&]() {
            // one thread drives every rank: kernels are asynchronous, the collectives of the N ranks go into one group
            if (!hqs) {
                for (int r = 0; r < ranks; ++r) { CHECK(ctx[(size_t)r], pcr_frame_begin(ctx[(size_t)r], &p, PCR_METHOD_BASIC)); CHECK(ctx[(size_t)r], pcr_render_basic(ctx[(size_t)r], &p)); }
                DCHECK(pcr_dist_group_begin());
                for (int r = 0; r < ranks; ++r) DCHECK(sliced ? pcr_dist_merge_min_sliced(dist[(size_t)r]) : pcr_dist_merge_min(dist[(size_t)r], root));
                DCHECK(pcr_dist_group_end());
                if (sliced) {
                    for (int r = 0; r < ranks; ++r) {
                        size_t first, px; my_slice(r, &first, &px);
                        const void *merged = pcr_dist_merged_slice(dist[(size_t)r]);
                        if (!merged) throw std::runtime_error(std::string("pcr_dist_merged_slice: ") + pcr_dist_last_error());
                        CHECK(ctx[(size_t)r], pcr_resolve_basic_range(ctx[(size_t)r], &p, merged, px, (uint32_t *)pcr_device_rgba(ctx[(size_t)r]) + first));
                    }
                    gather();
                } else
                for (int r = 0; r < ranks; ++r) if (allreduce || r == 0) CHECK(ctx[(size_t)r], pcr_resolve_basic(ctx[(size_t)r], &p));
            } else if (sliced) {
                for (int r = 0; r < ranks; ++r) { CHECK(ctx[(size_t)r], pcr_frame_begin(ctx[(size_t)r], &p, PCR_METHOD_HQS)); CHECK(ctx[(size_t)r], pcr_render_hqs_depth(ctx[(size_t)r], &p)); }
                DCHECK(pcr_dist_group_begin());
                for (int r = 0; r < ranks; ++r) DCHECK(pcr_dist_merge_min(dist[(size_t)r], PCR_DIST_ALL));
                DCHECK(pcr_dist_group_end());
                for (int r = 0; r < ranks; ++r) CHECK(ctx[(size_t)r], pcr_render_hqs_color(ctx[(size_t)r], &p));
                DCHECK(pcr_dist_group_begin());
                for (int r = 0; r < ranks; ++r) DCHECK(pcr_dist_merge_sum_sliced(dist[(size_t)r]));
                DCHECK(pcr_dist_group_end());
                for (int r = 0; r < ranks; ++r) {
                    size_t first, px; my_slice(r, &first, &px);
                    pcr_ctx *c = ctx[(size_t)r];
                    CHECK(c, pcr_resolve_hqs_range(c, &p, (const uint64_t *)pcr_device_framebuffer(c) + first, (const uint64_t *)pcr_device_rg(c) + first,
                                                   (const uint64_t *)pcr_device_ba(c) + first, px, (uint32_t *)pcr_device_rgba(c) + first));
                }
                gather();
            } else {
                for (int r = 0; r < ranks; ++r) { CHECK(ctx[(size_t)r], pcr_frame_begin(ctx[(size_t)r], &p, PCR_METHOD_HQS)); CHECK(ctx[(size_t)r], pcr_render_hqs_depth(ctx[(size_t)r], &p)); }
                DCHECK(pcr_dist_group_begin());
                for (int r = 0; r < ranks; ++r) DCHECK(pcr_dist_merge_min(dist[(size_t)r], PCR_DIST_ALL));
                DCHECK(pcr_dist_group_end());
                for (int r = 0; r < ranks; ++r) CHECK(ctx[(size_t)r], pcr_render_hqs_color(ctx[(size_t)r], &p));
                DCHECK(pcr_dist_group_begin());
                for (int r = 0; r < ranks; ++r) DCHECK(pcr_dist_merge_sum(dist[(size_t)r], root));
                DCHECK(pcr_dist_group_end());
                for (int r = 0; r < ranks; ++r) if (allreduce || r == 0) CHECK(ctx[(size_t)r], pcr_resolve_hqs(ctx[(size_t)r], &p));
            }
        };
        auto sync_all = [&]() { for (int r = 0; r < ranks; ++r) CHECK(ctx[(size_t)r], pcr_synchronize(ctx[(size_t)r])); };
        frame(); frame(); sync_all();                                       // warm-up (first frame also releases the load-time buffers)
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < frames; ++k) frame();
        sync_all();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / std::max(1, frames);
        std::vector<uint64_t> fb((size_t)w * h);
        std::vector<uint32_t> rgba((size_t)w * h);
        CHECK(ctx[0], pcr_read_framebuffer(ctx[0], fb.data(), fb.size()));
        CHECK(ctx[0], pcr_read_rgba(ctx[0], rgba.data(), rgba.size()));
        size_t covered = 0;
        // (after a sliced exchange rank 0 holds only its slice of the merged u64 frame: covered pixels are counted in the image,
        // a pixel no point reached is the background colour there: resolve.cu:166)
        if (sliced && !hqs) for (uint32_t v : rgba) covered += v != PCR_BACKGROUND_COLOR;
        else for (uint64_t v : fb) covered += v != ~0ull;
        int64_t points = 0;
        for (int r = 0; r < ranks; ++r) { pcr_render_stats st; CHECK(ctx[(size_t)r], pcr_get_stats(ctx[(size_t)r], &st)); points += st.points_iterated; }
        const char *merge_name = exchange == PCR_DIST_EXCHANGE_SLICED ? (allreduce ? "reduce-scatter + all-gather of the image" : "reduce-scatter + gather of the image")
                               : exchange == PCR_DIST_EXCHANGE_SLICED_P2P ? (allreduce ? "all-to-all + local min + all-gather of the image" : "all-to-all + local min + gather of the image")
                               : allreduce ? "allreduce" : "reduce to rank 0";
        std::printf("{\"method\": \"%s\", \"ranks\": %d, \"batches\": %lld, \"batches_rank0\": %lld, \"merge\": \"%s\", \"ms_per_frame\": %.4f, "
                    "\"points_iterated\": %lld, \"covered_pixels\": %zu, \"fb_fnv1a\": ",
                    method.c_str(), ranks, (long long)f.num_batches, (long long)count[0], merge_name, ms, (long long)points, covered);
        if (sliced) std::printf("null");
        else std::printf("\"%016llx\"", (unsigned long long)fnv1a(fb.data(), fb.size() * 8));
        std::printf(", \"rgba_fnv1a\": \"%016llx\"}\n", (unsigned long long)fnv1a(rgba.data(), rgba.size() * 4));
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pcr_render_dist: %s\n", e.what());
        status = 1;
    }
    for (pcr_dist *d : dist) pcr_dist_destroy(d);
    for (pcr_ctx *c : ctx) if (c) pcr_destroy(c);
    return status;
}
