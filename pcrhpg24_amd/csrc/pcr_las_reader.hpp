// pcr_las_reader.hpp — LAS 1.x point reader shared by the host tools.
// Follows LasLoader::loadSync (src/preprocess.cpp:74-171) and ComputeLasData::loadHeader
// (modules/compute/ComputeLasLoader.h:55-95): header offsets 24/25 version, 96 offset to point data, 104 format,
// 105 record length, 107 (<=1.3) or 247 (1.4) point count, 131 scale, 155 offset, 179..219 max/min; records: int32
// X,Y,Z at 0,4,8 and uint16 R,G,B at 20 / 28 / 30 for formats 2 / 3 / 7-8, colour components above 255 divided by 256.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "pcr_encode.h"

namespace pcr_host {

struct LasPoints {
    std::vector<int32_t> x, y, z;
    std::vector<uint32_t> color;       // 0x00BBGGRR
    pcr_las_info las{};
    int64_t numPoints = 0;
    int format = 0, recordLength = 0;
};

inline bool read_las(const std::string &path, LasPoints &out, std::string &err)
{
    auto rd = [](const std::vector<char> &b, size_t off, auto &v) { std::memcpy(&v, b.data() + off, sizeof v); };
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { err = "cannot open " + path; return false; }
    const size_t size = (size_t)f.tellg();
    if (size < 227) { err = path + ": not a LAS file"; return false; }
    std::vector<char> hdr(size < 2048 ? size : 2048);
    f.seekg(0); f.read(hdr.data(), (std::streamsize)hdr.size());
    if (std::memcmp(hdr.data(), "LASF", 4) != 0) { err = path + ": missing LASF signature"; return false; }
    uint32_t offsetToPointData = 0, legacyCount = 0; uint8_t format = 0, vMajor = 0, vMinor = 0; uint16_t recordLength = 0;
    rd(hdr, 96, offsetToPointData); rd(hdr, 104, format); rd(hdr, 105, recordLength); rd(hdr, 24, vMajor); rd(hdr, 25, vMinor);
    rd(hdr, 107, legacyCount);
    int64_t numPoints = legacyCount;
    if (!(vMajor == 1 && vMinor <= 3)) { numPoints = 0; if (hdr.size() >= 255) rd(hdr, 247, numPoints); }
    for (int k = 0; k < 3; ++k) {
        rd(hdr, 131 + 8 * (size_t)k, out.las.scale[k]);
        rd(hdr, 155 + 8 * (size_t)k, out.las.offset[k]);
        rd(hdr, 179 + 16 * (size_t)k, out.las.max[k]);
        rd(hdr, 187 + 16 * (size_t)k, out.las.min[k]);
    }
    int offset_rgb = 0;
    if (format == 2) offset_rgb = 20; else if (format == 3) offset_rgb = 28; else if (format == 7 || format == 8) offset_rgb = 30;
    if (recordLength < 12 || numPoints <= 0 || (uint64_t)offsetToPointData + (uint64_t)numPoints * recordLength > size) {
        err = path + ": inconsistent header (points " + std::to_string(numPoints) + ", record " + std::to_string(recordLength) + ")";
        return false;
    }
    out.numPoints = numPoints; out.format = format; out.recordLength = recordLength;
    out.x.resize((size_t)numPoints); out.y.resize((size_t)numPoints); out.z.resize((size_t)numPoints); out.color.resize((size_t)numPoints);
    std::vector<char> rec((size_t)recordLength * 65536);
    f.seekg((std::streamoff)offsetToPointData);
    for (int64_t done = 0; done < numPoints;) {
        const int64_t n = std::min<int64_t>(65536, numPoints - done);
        f.read(rec.data(), (std::streamsize)(n * recordLength));
        if (!f) { err = path + ": truncated point data"; return false; }
        for (int64_t i = 0; i < n; ++i) {
            const char *r = rec.data() + i * recordLength;
            const size_t k = (size_t)(done + i);
            std::memcpy(&out.x[k], r + 0, 4); std::memcpy(&out.y[k], r + 4, 4); std::memcpy(&out.z[k], r + 8, 4);
            uint16_t R = 0, G = 0, B = 0;
            if (offset_rgb + 6 <= recordLength) { std::memcpy(&R, r + offset_rgb, 2); std::memcpy(&G, r + offset_rgb + 2, 2); std::memcpy(&B, r + offset_rgb + 4, 2); }
            const uint32_t UR = R > 255 ? R / 256 : R, UG = G > 255 ? G / 256 : G, UB = B > 255 ? B / 256 : B;   // preprocess.cpp:150-152
            out.color[k] = UR | (UG << 8) | (UB << 16);
        }
        done += n;
    }
    return true;
}

} // namespace pcr_host
