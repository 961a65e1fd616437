// pcr_preprocess — native twin of the reference's offline encoder CLI (src/preprocess.cpp:1167-1279):
//     pcr_preprocess <in.las> <out.huffman> <sort 0|1>
// LAS parsing follows LasLoader::loadSync (src/preprocess.cpp:74-171): header offsets 24/25 version, 96 offset to
// point data, 104 format, 105 record length, 107 (<=1.3) or 247 (1.4) point count, 131 scale, 155 offset,
// 179..219 max/min; records: int32 X,Y,Z at 0,4,8 and uint16 R,G,B at 20 / 28 / 30 for formats 2 / 3 / 7-8,
// colour components above 255 divided by 256.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "pcr_encode.h"

template <class T> static T rd(const std::vector<char> &b, size_t off) { T v; std::memcpy(&v, b.data() + off, sizeof v); return v; }

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: pcr_preprocess <in.las> <out.huffman> <sort 0|1> [threads]\n"); return 2; }   // preprocess.cpp:1169-1180
    const std::string in = argv[1], out = argv[2];
    const int sort = std::atoi(argv[3]);
    const int threads = argc > 4 ? std::atoi(argv[4]) : 0;
    std::ifstream f(in, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", in.c_str()); return 1; }
    const size_t size = (size_t)f.tellg();
    if (size < 227) { std::fprintf(stderr, "%s: not a LAS file\n", in.c_str()); return 1; }
    std::vector<char> hdr(size < 2048 ? size : 2048);
    f.seekg(0); f.read(hdr.data(), (std::streamsize)hdr.size());
    if (std::memcmp(hdr.data(), "LASF", 4) != 0) { std::fprintf(stderr, "%s: missing LASF signature\n", in.c_str()); return 1; }
    const uint64_t offsetToPointData = rd<uint32_t>(hdr, 96);
    const int format = rd<uint8_t>(hdr, 104), recordLength = rd<uint16_t>(hdr, 105);
    const int vMajor = rd<uint8_t>(hdr, 24), vMinor = rd<uint8_t>(hdr, 25);
    int64_t numPoints = (vMajor == 1 && vMinor <= 3) ? (int64_t)rd<uint32_t>(hdr, 107) : (hdr.size() >= 255 ? rd<int64_t>(hdr, 247) : 0);
    pcr_las_info las;
    for (int k = 0; k < 3; ++k) {
        las.scale[k] = rd<double>(hdr, 131 + 8 * (size_t)k);
        las.offset[k] = rd<double>(hdr, 155 + 8 * (size_t)k);
        las.max[k] = rd<double>(hdr, 179 + 16 * (size_t)k);
        las.min[k] = rd<double>(hdr, 187 + 16 * (size_t)k);
    }
    int offset_rgb = 0;
    if (format == 2) offset_rgb = 20; else if (format == 3) offset_rgb = 28; else if (format == 7 || format == 8) offset_rgb = 30;
    if (recordLength < 12 || numPoints <= 0 || offsetToPointData + (uint64_t)numPoints * recordLength > size) {
        std::fprintf(stderr, "%s: inconsistent header (points %lld, record %d)\n", in.c_str(), (long long)numPoints, recordLength);
        return 1;
    }
    std::vector<int32_t> x((size_t)numPoints), y((size_t)numPoints), z((size_t)numPoints);
    std::vector<uint32_t> c((size_t)numPoints);
    std::vector<char> rec((size_t)recordLength * 65536);
    f.seekg((std::streamoff)offsetToPointData);
    for (int64_t done = 0; done < numPoints;) {
        int64_t n = std::min<int64_t>(65536, numPoints - done);
        f.read(rec.data(), (std::streamsize)(n * recordLength));
        for (int64_t i = 0; i < n; ++i) {
            const char *r = rec.data() + i * recordLength;
            std::memcpy(&x[(size_t)(done + i)], r + 0, 4); std::memcpy(&y[(size_t)(done + i)], r + 4, 4); std::memcpy(&z[(size_t)(done + i)], r + 8, 4);
            uint16_t R = 0, G = 0, B = 0;
            if (offset_rgb + 6 <= recordLength) { std::memcpy(&R, r + offset_rgb, 2); std::memcpy(&G, r + offset_rgb + 2, 2); std::memcpy(&B, r + offset_rgb + 4, 2); }
            uint32_t UR = R > 255 ? R / 256 : R, UG = G > 255 ? G / 256 : G, UB = B > 255 ? B / 256 : B;   // preprocess.cpp:150-152
            c[(size_t)(done + i)] = UR | (UG << 8) | (UB << 16);
        }
        done += n;
    }
    void *bytes = nullptr; size_t len = 0; pcr_encode_stats st;
    if (pcr_encode_points(x.data(), y.data(), z.data(), c.data(), numPoints, &las, sort, 0, threads, &bytes, &len, &st)) {
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
