// pcr_preprocess — native twin of the reference's offline encoder CLI (src/preprocess.cpp:1167-1279):
//     pcr_preprocess <in.las> <out.huffman> <sort 0|1>
// LAS parsing: pcr_las_reader.hpp (LasLoader::loadSync, src/preprocess.cpp:74-171).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "pcr_gpu_encode.h"
#include "pcr_las_reader.hpp"

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: pcr_preprocess <in.las> <out.huffman> <sort 0|1> [threads] [--pad-tails] [--bc7] [--gpu]\n"); return 2; }   // preprocess.cpp:1169-1180
    const std::string in = argv[1], out = argv[2];
    int flags = std::atoi(argv[3]) ? PCR_ENCODE_MORTON_SORT : 0;
    int threads = 0;
    bool gpu = false;
    for (int i = 4; i < argc; ++i) {
        if (std::string(argv[i]) == "--pad-tails") flags |= PCR_ENCODE_PAD_TAILS;   // not in the reference, see pcr_encode.h
        else if (std::string(argv[i]) == "--bc7") flags |= PCR_ENCODE_BC7;         // the reference built with COLOR_COMPRESSION == 7
        else if (std::string(argv[i]) == "--gpu") gpu = true;                       // pcr_gpu_encode_points: same bytes, encoded on the MI355X
        else threads = std::atoi(argv[i]);
    }
    pcr_host::LasPoints pts;
    std::string err;
    if (!pcr_host::read_las(in, pts, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    const std::vector<int32_t> &x = pts.x, &y = pts.y, &z = pts.z;
    const std::vector<uint32_t> &c = pts.color;
    const pcr_las_info &las = pts.las;
    const int64_t numPoints = pts.numPoints;
    void *bytes = nullptr; size_t len = 0; pcr_encode_stats st;
    if (gpu) {
        pcr_ctx *ctx = nullptr;
        if (pcr_create(0, &ctx)) { std::fprintf(stderr, "pcr_create: %s\n", pcr_last_error(nullptr)); return 1; }
        if (pcr_gpu_encode_points(ctx, x.data(), y.data(), z.data(), c.data(), numPoints, &las, flags, 0, &bytes, &len, &st)) {
            std::fprintf(stderr, "GPU encode failed: %s\n", pcr_last_error(ctx));
            pcr_destroy(ctx);
            return 1;
        }
        pcr_destroy(ctx);
    } else if (pcr_encode_points(x.data(), y.data(), z.data(), c.data(), numPoints, &las, flags, 0, threads, &bytes, &len, &st)) {
        std::fprintf(stderr, "encode failed: %s\n", pcr_host_last_error());
        return 1;
    }
    std::ofstream o(out, std::ios::binary);
    o.write((const char *)bytes, (std::streamsize)len);
    pcr_host_free(bytes);
    // the reference prints compression ratios at this point (preprocess.cpp:1238-1263)
    const double raw = 16.0 * (double)st.num_points;
    std::printf("points %lld (padded %lld) batches %lld  file %lld bytes  %.3f bits/point encoded  %.2f %% escaped symbols  ratio %.2f\n",
                (long long)st.num_points_in, (long long)st.num_points, (long long)st.num_batches, (long long)st.file_bytes,
                8.0 * (double)st.encoded_bytes / (double)st.num_points, 100.0 * (double)st.escaped_symbols / (double)st.total_symbols,
                raw / (double)st.file_bytes);
    return o ? 0 : 1;
}
