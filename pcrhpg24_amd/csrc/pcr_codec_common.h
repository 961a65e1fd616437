// pcr_codec_common.h — pieces of the encoder that the CPU encoder (pcr_encoder.cpp, g++) and the GPU encoder
// (pcr_gpu_encoder.hip, hipcc) must compute identically, written once: Morton key, BC1 block encoder.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define PCR_HD __host__ __device__ inline
#else
#define PCR_HD inline
#endif

namespace pcr_codec {

// ---------------------------------------------------------------------------------------------
// Morton key (src/mymorton.h:12-37)
// ---------------------------------------------------------------------------------------------
struct MortonKey {
    uint32_t hi; uint64_t lo;
};
PCR_HD bool operator<(const MortonKey &a, const MortonKey &b) { return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo; }
PCR_HD bool operator==(const MortonKey &a, const MortonKey &b) { return a.hi == b.hi && a.lo == b.lo; }

PCR_HD uint64_t spread3_21(uint32_t v) // bits 0..20 of v to positions 3i
{
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8)  & 0x100F00F00F00F00Full;
    x = (x | x << 4)  & 0x10C30C30C30C30C3ull;
    x = (x | x << 2)  & 0x1249249249249249ull;
    return x;
}

PCR_HD MortonKey morton_key(uint32_t X, uint32_t Y, uint32_t Z)
{
    MortonKey k;
    k.lo = spread3_21(X) | (spread3_21(Y) << 1) | (spread3_21(Z) << 2);     // mymorton.h:16-20
    k.lo |= (uint64_t)((X >> 21) & 1u) << 63;                                // :23
    // The reference accumulates these in 64-bit arithmetic and stores them into a uint32_t (mymorton.h:10,30-34):
    // bit 31 of X would land on bit 32 and is dropped by that truncation. Reproduced, not fixed.
    uint64_t hi = ((Y >> 21) & 1u) | (((Z >> 21) & 1u) << 1);                // :26-27
    for (int i = 22; i < 32; ++i) {                                          // :30-34
        hi |= (uint64_t)((X >> i) & 1u) << (3 * (i - 21) + 2);
        hi |= (uint64_t)((Y >> i) & 1u) << (3 * (i - 21) + 0);
        hi |= (uint64_t)((Z >> i) & 1u) << (3 * (i - 21) + 1);
    }
    k.hi = (uint32_t)hi;
    return k;
}

PCR_HD uint32_t shift_coord(int32_t v) { return (uint32_t)((int64_t)v - (int64_t)INT32_MIN); } // mymorton.h:47-49

// ---------------------------------------------------------------------------------------------
// BC1 (4-colour mode only; decoder = modules/huffman_mem_iter_cuda/render.cu:23-65)
// ---------------------------------------------------------------------------------------------
PCR_HD void expand565(uint32_t c, int &r, int &g, int &b)
{
    int cr = (c >> 11) & 31, cg = (c >> 5) & 63, cb = c & 31;
    r = (cr << 3) | (cr >> 2); g = (cg << 2) | (cg >> 4); b = (cb << 3) | (cb >> 2);
}

PCR_HD uint32_t to565(const int *c)
{
    int r = (c[0] * 31 + 127) / 255, g = (c[1] * 63 + 127) / 255, b = (c[2] * 31 + 127) / 255;
    return (uint32_t)((r << 11) | (g << 5) | b);
}

// 16 colours (0x00BBGGRR) -> 8 bytes: bounding-box endpoints, nearest palette entry per pixel
PCR_HD void bc1_encode(const uint32_t *px, uint8_t *out)
{
    int mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
    for (int i = 0; i < 16; ++i)
        for (int c = 0; c < 3; ++c) {
            int v = (px[i] >> (8 * c)) & 255;
            mn[c] = v < mn[c] ? v : mn[c]; mx[c] = v > mx[c] ? v : mx[c];
        }
    uint32_t c0 = to565(mx), c1 = to565(mn);
    if (c0 < c1) { uint32_t t = c0; c0 = c1; c1 = t; }
    int pal[4][3];
    expand565(c0, pal[0][0], pal[0][1], pal[0][2]);
    expand565(c1, pal[1][0], pal[1][1], pal[1][2]);
    for (int c = 0; c < 3; ++c) {
        pal[2][c] = (pal[0][c] * 2 + pal[1][c]) / 3;
        pal[3][c] = (pal[0][c] + pal[1][c] * 2) / 3;
    }
    out[0] = (uint8_t)(c0 & 255); out[1] = (uint8_t)(c0 >> 8); out[2] = (uint8_t)(c1 & 255); out[3] = (uint8_t)(c1 >> 8);
    out[4] = out[5] = out[6] = out[7] = 0;
    for (int i = 0; i < 16; ++i) {
        int best = 0, bestd = 1 << 30;
        for (int k = 0; k < 4; ++k) {
            int d = 0;
            for (int c = 0; c < 3; ++c) { int e = (int)((px[i] >> (8 * c)) & 255) - pal[k][c]; d += e * e; }
            if (d < bestd) { bestd = d; best = k; }
        }
        out[4 + i / 4] |= (uint8_t)(best << (2 * (i % 4)));
    }
}

} // namespace pcr_codec
