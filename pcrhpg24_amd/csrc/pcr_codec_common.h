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

// 8-bit channel value -> the 5- or 6-bit endpoint whose expansion is nearest
PCR_HD int quant_channel(int v, int bits)
{
    const int top = (1 << bits) - 1;
    int q = (v * top + 127) / 255;
    int best = q, bestd = 1 << 30;
    for (int c = q - 1; c <= q + 1; ++c) {
        if (c < 0 || c > top) continue;
        const int e = bits == 5 ? ((c << 3) | (c >> 2)) : ((c << 2) | (c >> 4));
        const int d = e > v ? e - v : v - e;
        if (d < bestd) { bestd = d; best = c; }
    }
    return best;
}
PCR_HD uint32_t to565_nearest(const int *c)
{
    int v[3];
    for (int k = 0; k < 3; ++k) v[k] = c[k] < 0 ? 0 : c[k] > 255 ? 255 : c[k];
    return (uint32_t)((quant_channel(v[0], 5) << 11) | (quant_channel(v[1], 6) << 5) | quant_channel(v[2], 5));
}

struct Bc1Trial { uint32_t c0, c1, selectors; int err; };

// Selectors (nearest of the four palette colours, as the kernel decodes them: always 4-colour mode) and squared error of
// one endpoint pair.
PCR_HD Bc1Trial bc1_try(const int px[16][3], uint32_t c0, uint32_t c1)
{
    int pal[4][3];
    expand565(c0, pal[0][0], pal[0][1], pal[0][2]);
    expand565(c1, pal[1][0], pal[1][1], pal[1][2]);
    for (int c = 0; c < 3; ++c) {
        pal[2][c] = (pal[0][c] * 2 + pal[1][c]) / 3;
        pal[3][c] = (pal[0][c] + pal[1][c] * 2) / 3;
    }
    Bc1Trial t; t.c0 = c0; t.c1 = c1; t.selectors = 0; t.err = 0;
    for (int i = 0; i < 16; ++i) {
        int best = 0, bestd = 1 << 30;
        for (int k = 0; k < 4; ++k) {
            int d = 0;
            for (int c = 0; c < 3; ++c) { const int e = px[i][c] - pal[k][c]; d += e * e; }
            if (d < bestd) { bestd = d; best = k; }
        }
        t.selectors |= (uint32_t)best << (2 * i);
        t.err += bestd;
    }
    return t;
}

// 16 colours (0x00BBGGRR) -> 8 bytes. Own encoder (the reference uses rgbcx::encode_bc1(level 8), src/preprocess.cpp:282-297;
// any valid block decodes the same way in the kernel, what differs is how close the decoded colours come): integer
// arithmetic only, so the CPU and the GPU encoder produce the same bytes.
//   solid block      per channel the endpoint pair (e, e-4 .. e+4) whose 2/3-1/3 interpolant comes closest
//   otherwise        candidates = bounding-box diagonal, principal axis (integer power iteration on the covariance), then
//                    three rounds of {nearest selectors -> least-squares endpoints for those selectors}; least error wins
// tools/bc1_quality.py: PSNR against rgbcx's blocks on tests/golden/bc1_ref_blocks.npz.
PCR_HD void bc1_encode(const uint32_t *pix, uint8_t *out)
{
    int px[16][3];
    int mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0}, sum[3] = {0, 0, 0};
    for (int i = 0; i < 16; ++i)
        for (int c = 0; c < 3; ++c) {
            const int v = (int)((pix[i] >> (8 * c)) & 255);
            px[i][c] = v; sum[c] += v;
            mn[c] = v < mn[c] ? v : mn[c]; mx[c] = v > mx[c] ? v : mx[c];
        }
    Bc1Trial best;
    if (mn[0] == mx[0] && mn[1] == mx[1] && mn[2] == mx[2]) {
        // every pixel the same: selector 2 everywhere, (2 e0 + e1) / 3 per channel as close to the colour as 5/6 bits allow
        int e0[3], e1[3];
        for (int c = 0; c < 3; ++c) {
            const int bits = c == 1 ? 6 : 5, top = (1 << bits) - 1;
            int bd = 1 << 30;
            e0[c] = e1[c] = 0;
            for (int a = 0; a <= top; ++a)
                for (int b2 = a - 4; b2 <= a + 4; ++b2) {
                    if (b2 < 0 || b2 > top) continue;
                    const int xa = bits == 5 ? ((a << 3) | (a >> 2)) : ((a << 2) | (a >> 4));
                    const int xb = bits == 5 ? ((b2 << 3) | (b2 >> 2)) : ((b2 << 2) | (b2 >> 4));
                    const int v = (2 * xa + xb) / 3, d = v > mn[c] ? v - mn[c] : mn[c] - v;
                    if (d < bd) { bd = d; e0[c] = a; e1[c] = b2; }
                }
        }
        best.c0 = (uint32_t)((e0[0] << 11) | (e0[1] << 5) | e0[2]);
        best.c1 = (uint32_t)((e1[0] << 11) | (e1[1] << 5) | e1[2]);
        best.selectors = 0xAAAAAAAAu;
        best.err = 0;
    } else {
        best = bc1_try(px, to565(mx), to565(mn));                           // bounding-box diagonal (round 1's encoder)
        // principal axis of the 16 colours: covariance x 16 in integers, four power iterations from the box diagonal
        long long cov[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b2 = 0; b2 < 3; ++b2) {
                long long acc = 0;
                for (int i = 0; i < 16; ++i) acc += (long long)(16 * px[i][a] - sum[a]) * (16 * px[i][b2] - sum[b2]);
                cov[a][b2] = acc;
            }
        long long ax[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
        for (int it = 0; it < 4; ++it) {
            long long n[3], big = 0;
            for (int a = 0; a < 3; ++a) {
                n[a] = cov[a][0] * ax[0] + cov[a][1] * ax[1] + cov[a][2] * ax[2];
                const long long m = n[a] < 0 ? -n[a] : n[a];
                big = m > big ? m : big;
            }
            if (big == 0) break;
            int sh = 0;
            while ((big >> sh) > (1 << 12)) ++sh;                           // keep the vector within 13 bits
            for (int a = 0; a < 3; ++a) ax[a] = n[a] >= 0 ? n[a] >> sh : -((-n[a]) >> sh);
        }
        const long long len2 = ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2];
        if (len2 > 0) {
            // extreme projections onto the axis through the mean -> endpoints
            long long lo = 0, hi = 0;
            for (int i = 0; i < 16; ++i) {
                long long d = 0;
                for (int c = 0; c < 3; ++c) d += (long long)(16 * px[i][c] - sum[c]) * ax[c];
                lo = (i == 0 || d < lo) ? d : lo; hi = (i == 0 || d > hi) ? d : hi;
            }
            int p0[3], p1[3];
            for (int c = 0; c < 3; ++c) {       // mean + axis * t / |axis|^2, everything still x 16
                const long long a0 = sum[c] * len2 + hi * ax[c], a1 = sum[c] * len2 + lo * ax[c], den = 16 * len2;
                p0[c] = (int)((a0 >= 0 ? a0 + den / 2 : a0 - den / 2) / den);
                p1[c] = (int)((a1 >= 0 ? a1 + den / 2 : a1 - den / 2) / den);
            }
            const Bc1Trial t = bc1_try(px, to565_nearest(p0), to565_nearest(p1));
            if (t.err < best.err) best = t;
        }
        // refinement: with the selectors fixed, the endpoints that minimise the squared error solve a 2x2 system per channel
        // (weights in thirds: selector 0 -> 3/3 of c0, 1 -> 0, 2 -> 2/3, 3 -> 1/3)
        Bc1Trial cur = best;
        for (int round = 0; round < 3; ++round) {
            int A = 0, B = 0, Cc = 0;
            long long X[3] = {0, 0, 0}, Y[3] = {0, 0, 0};
            for (int i = 0; i < 16; ++i) {
                const int sel = (int)((cur.selectors >> (2 * i)) & 3u);
                const int w = sel == 0 ? 3 : sel == 1 ? 0 : sel == 2 ? 2 : 1;
                A += w * w; B += w * (3 - w); Cc += (3 - w) * (3 - w);
                for (int c = 0; c < 3; ++c) { X[c] += (long long)w * px[i][c]; Y[c] += (long long)(3 - w) * px[i][c]; }
            }
            const long long det = (long long)A * Cc - (long long)B * B;
            if (det == 0) break;                                            // all pixels on one palette entry
            int p0[3], p1[3];
            for (int c = 0; c < 3; ++c) {
                const long long n0 = 3 * (Cc * X[c] - B * Y[c]), n1 = 3 * (A * Y[c] - B * X[c]);
                p0[c] = (int)((n0 >= 0 ? n0 + det / 2 : n0 - det / 2) / det);
                p1[c] = (int)((n1 >= 0 ? n1 + det / 2 : n1 - det / 2) / det);
            }
            const Bc1Trial t = bc1_try(px, to565_nearest(p0), to565_nearest(p1));
            if (t.err < best.err) best = t;
            if (t.c0 == cur.c0 && t.c1 == cur.c1) break;
            cur = t;
        }
    }
    // a valid 4-colour block has c0 > c1: swap the endpoints (selectors 0 <-> 1, 2 <-> 3) if need be
    if (best.c0 < best.c1) {
        const uint32_t t = best.c0; best.c0 = best.c1; best.c1 = t;
        best.selectors ^= 0x55555555u;
    }
    out[0] = (uint8_t)(best.c0 & 255); out[1] = (uint8_t)(best.c0 >> 8); out[2] = (uint8_t)(best.c1 & 255); out[3] = (uint8_t)(best.c1 >> 8);
    out[4] = (uint8_t)best.selectors; out[5] = (uint8_t)(best.selectors >> 8); out[6] = (uint8_t)(best.selectors >> 16); out[7] = (uint8_t)(best.selectors >> 24);
}

// BC7 mode-6 block of 16 colours (0x00BBGGRR), the colour format of the reference built with COLOR_COMPRESSION == 7
// (src/preprocess.cpp:299-316 calls bc7enc with m_mode_mask = 1 << 6). Own encoder, integer arithmetic: endpoints = the ends
// of the bounding box along the signs of the channels' covariance with the widest channel, each rounded to 7 bits + the
// p-bit (shared by the endpoint's channels) with the smaller error, alpha 255; 4-bit indices = nearest of BC7's sixteen
// weights on the segment; endpoints swapped when pixel 0's index would need its fourth bit (the anchor has three).
// Layout (render.cu:66-110, struct bc7_mode_6): low quadword mode:7 (= 0x40) r0:7 r1:7 g0:7 g1:7 b0:7 b1:7 a0:7 a1:7 p0:1,
// high quadword p1:1, index of pixel 0 :3, indices of pixels 1..15 :4 each.
PCR_HD void bc7_mode6_encode(const uint32_t *pix, uint8_t *out)
{
    const int weights[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};
    int px[16][3], mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0}, sum[3] = {0, 0, 0};
    for (int i = 0; i < 16; ++i)
        for (int c = 0; c < 3; ++c) {
            px[i][c] = (int)((pix[i] >> (8 * c)) & 255u);
            mn[c] = px[i][c] < mn[c] ? px[i][c] : mn[c]; mx[c] = px[i][c] > mx[c] ? px[i][c] : mx[c]; sum[c] += px[i][c];
        }
    int main_c = 0;
    for (int c = 1; c < 3; ++c) if (mx[c] - mn[c] > mx[main_c] - mn[main_c]) main_c = c;
    int e[2][3];
    for (int c = 0; c < 3; ++c) {
        long long cov = 0;
        for (int i = 0; i < 16; ++i) cov += (long long)(16 * px[i][c] - sum[c]) * (16 * px[i][main_c] - sum[main_c]);
        e[0][c] = cov >= 0 ? mn[c] : mx[c]; e[1][c] = cov >= 0 ? mx[c] : mn[c];
    }
    int q[2][3], pbit[2];
    for (int k = 0; k < 2; ++k) {
        int best_err = 1 << 30;
        for (int pb = 0; pb < 2; ++pb) {
            int err = 0, qq[3];
            for (int c = 0; c < 3; ++c) {
                int v = (e[k][c] - pb + 1) >> 1;
                v = v < 0 ? 0 : v > 127 ? 127 : v;
                qq[c] = v;
                const int d = ((v << 1) | pb) - e[k][c];
                err += d * d;
            }
            if (err < best_err) { best_err = err; pbit[k] = pb; for (int c = 0; c < 3; ++c) q[k][c] = qq[c]; }
        }
    }
    int idx[16];
    {
        int r0[3], d[3], dd = 0;
        for (int c = 0; c < 3; ++c) { r0[c] = (q[0][c] << 1) | pbit[0]; d[c] = ((q[1][c] << 1) | pbit[1]) - r0[c]; dd += d[c] * d[c]; }
        for (int i = 0; i < 16; ++i) {
            int best = 0, best_err = 1 << 30;
            if (dd) {
                for (int k = 0; k < 16; ++k) {
                    int err = 0;
                    for (int c = 0; c < 3; ++c) {
                        const int v = ((r0[c] * (64 - weights[k]) + (r0[c] + d[c]) * weights[k] + 32) >> 6) - px[i][c];
                        err += v * v;
                    }
                    if (err < best_err) { best_err = err; best = k; }
                }
            }
            idx[i] = best;
        }
    }
    if (idx[0] >= 8) {                                      // the anchor index has three bits: swap the endpoints
        for (int c = 0; c < 3; ++c) { const int t = q[0][c]; q[0][c] = q[1][c]; q[1][c] = t; }
        const int t = pbit[0]; pbit[0] = pbit[1]; pbit[1] = t;
        for (int i = 0; i < 16; ++i) idx[i] = 15 - idx[i];
    }
    unsigned long long lo = 0x40ull, hi = 0;
    lo |= (unsigned long long)q[0][0] << 7;  lo |= (unsigned long long)q[1][0] << 14;
    lo |= (unsigned long long)q[0][1] << 21; lo |= (unsigned long long)q[1][1] << 28;
    lo |= (unsigned long long)q[0][2] << 35; lo |= (unsigned long long)q[1][2] << 42;
    lo |= 127ull << 49; lo |= 127ull << 56; lo |= (unsigned long long)pbit[0] << 63;
    hi |= (unsigned long long)pbit[1];
    hi |= (unsigned long long)idx[0] << 1;
    for (int i = 1; i < 16; ++i) hi |= (unsigned long long)idx[i] << (4 * i);
    for (int k = 0; k < 8; ++k) { out[k] = (uint8_t)(lo >> (8 * k)); out[8 + k] = (uint8_t)(hi >> (8 * k)); }
}

} // namespace pcr_codec
