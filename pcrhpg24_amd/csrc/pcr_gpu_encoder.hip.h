// pcr_gpu_encoder.hip.h — GPU encoder (include/pcr_gpu_encode.h). Included at the end of pcr_api.hip (one translation
// unit: it uses pcr_ctx, set_err and HIP_TRY from there).
//
// One chunk (<= 100 batches, preprocess.cpp:925-1165) at a time, every stage a kernel over all batches of the chunk:
//   k_enc_keys + 2 x hipcub radix sort      stable sort by the 96-bit Morton key        preprocess.cpp:959-977
//   k_enc_gather                            pad by repeating the last point, permute   :945-955
//   k_enc_deltas                            start values, deltas, batch min/max          :211-227, 318-343
//   hipcub segmented sort + k_enc_unique    symbol alphabet and frequencies per batch    huffman.h:94-113
//   hipcub segmented sort (by frequency)    stable ascending order of the leaves
//   k_enc_tree                              two-queue Huffman + 12-bit clipped codes      huffman.h:58-69, 180-217
//   k_enc_table                             4096-entry decoder table                      huffman.h:220-240
//   k_enc_pack                              MSB-first packing, escapes, completion times  huffman.h:242-300
//   k_enc_interleave + k_enc_compact        (time, lane) order per 32-lane cluster       preprocess.cpp:540-587
//   k_enc_bc1                               colour blocks                                 preprocess.cpp:282-297
// The host then lays the batch records out exactly as pcr_encoder.cpp does (BatchDumpData.h:151-202).
#pragma once

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>

#include "pcr_codec_common.h"
#include "pcr_gpu_encode.h"

namespace pcr {
namespace enc {

constexpr int NT = PCR_WORKGROUP_SIZE, PPT = PCR_POINTS_PER_THREAD, NPB = PCR_POINTS_PER_BATCH;
constexpr int SYMS = NPB * 3;                   // symbols per batch (the first delta of every chain is a stored 0)
constexpr int CHAIN_SYMS = PPT * 3;             // 192
constexpr int MAXW = 72;                        // 192 symbols x 12 bits / 32
constexpr int MAXSLOT = MAXW + 2;               // + the two zero words of PCR_ENCODE_PAD_TAILS
constexpr int CL_WORDS = PCR_CLUSTER_LANES * MAXSLOT;
constexpr int TIMES = CHAIN_SYMS + 2;           // slot times -1 .. 192

// ---- sort --------------------------------------------------------------------------------------------------
__global__ void k_enc_keys(const int32_t *x, const int32_t *y, const int32_t *z, int64_t n_real, int64_t n, uint64_t *lo,
                           uint32_t *hi, uint32_t *idx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t s = i < n_real ? i : n_real - 1;                           // padding repeats the last point
    const pcr_codec::MortonKey k = pcr_codec::morton_key(pcr_codec::shift_coord(x[s]), pcr_codec::shift_coord(y[s]), pcr_codec::shift_coord(z[s]));
    lo[i] = k.lo; hi[i] = k.hi; idx[i] = (uint32_t)i;
}

__global__ void k_enc_gather_u32(const uint32_t *src, const uint32_t *idx, int64_t n, uint32_t *dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ void k_enc_gather_points(const int32_t *x, const int32_t *y, const int32_t *z, const uint32_t *c, const uint32_t *idx,
                                    int64_t n_real, int64_t n, int32_t *ox, int32_t *oy, int32_t *oz, uint32_t *oc)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t s = idx ? (int64_t)idx[i] : i;
    if (s >= n_real) s = n_real - 1;
    ox[i] = x[s]; oy[i] = y[s]; oz[i] = z[s]; oc[i] = c[s];
}

// ---- deltas --------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT) k_enc_deltas(const int32_t *x, const int32_t *y, const int32_t *z, int32_t *start,
                                                   int32_t *deltas, int32_t *mnmx)
{
    const int b = blockIdx.x, c = threadIdx.x;
    __shared__ int s_mn[3], s_mx[3];
    if (c < 3) { s_mn[c] = INT32_MAX; s_mx[c] = INT32_MIN; }
    __syncthreads();
    const size_t p0 = (size_t)b * NPB + (size_t)c * PPT;
    int32_t px = x[p0], py = y[p0], pz = z[p0];
    int32_t *st = start + ((size_t)b * NT + c) * 3;
    st[0] = px; st[1] = py; st[2] = pz;
    int32_t *d = deltas + (size_t)b * SYMS + (size_t)c * CHAIN_SYMS;
    d[0] = d[1] = d[2] = 0;                                                   // preprocess.cpp:328
    int mn[3] = {px, py, pz}, mx[3] = {px, py, pz};
    for (int i = 1; i < PPT; ++i) {                                           // :323-327 (int32 wrap)
        const int32_t qx = x[p0 + i], qy = y[p0 + i], qz = z[p0 + i];
        d[i * 3 + 0] = (int32_t)((uint32_t)qx - (uint32_t)px);
        d[i * 3 + 1] = (int32_t)((uint32_t)qy - (uint32_t)py);
        d[i * 3 + 2] = (int32_t)((uint32_t)qz - (uint32_t)pz);
        px = qx; py = qy; pz = qz;
        mn[0] = min(mn[0], qx); mx[0] = max(mx[0], qx);
        mn[1] = min(mn[1], qy); mx[1] = max(mx[1], qy);
        mn[2] = min(mn[2], qz); mx[2] = max(mx[2], qz);
    }
    for (int k = 0; k < 3; ++k) { atomicMin(&s_mn[k], mn[k]); atomicMax(&s_mx[k], mx[k]); }
    __syncthreads();
    if (c < 3) { mnmx[b * 6 + c] = s_mn[c]; mnmx[b * 6 + 3 + c] = s_mx[c]; }
}

// ---- alphabet ------------------------------------------------------------------------------------------------
// sorted: the batch's 196 608 symbols in ascending order. Writes the distinct symbols, their frequencies and count.
__global__ void __launch_bounds__(NT) k_enc_unique(const int32_t *sorted, int32_t *syms, uint32_t *freqs, uint32_t *leaf_id,
                                                   int32_t *num_syms, int32_t *seg_begin, int32_t *seg_end)
{
    const int b = blockIdx.x, t = threadIdx.x;
    const int32_t *s = sorted + (size_t)b * SYMS;
    constexpr int PER = SYMS / NT;                                            // 192
    typedef hipcub::BlockScan<int, NT> Scan;
    __shared__ typename Scan::TempStorage tmp;
    __shared__ int s_total;
    const int i0 = t * PER;
    int heads = 0;
    for (int i = i0; i < i0 + PER; ++i) heads += (i == 0 || s[i] != s[i - 1]) ? 1 : 0;
    int base, total;
    Scan(tmp).ExclusiveSum(heads, base, total);
    if (t == 0) { s_total = total; num_syms[b] = total; seg_begin[b] = b * SYMS; seg_end[b] = b * SYMS + total; }
    int32_t *sy = syms + (size_t)b * SYMS;
    uint32_t *fr = freqs + (size_t)b * SYMS;
    uint32_t *lid = leaf_id + (size_t)b * SYMS;     // first holds the position of each symbol's first occurrence
    int u = base;
    for (int i = i0; i < i0 + PER; ++i)
        if (i == 0 || s[i] != s[i - 1]) { sy[u] = s[i]; lid[u] = (uint32_t)i; ++u; }
    __syncthreads();
    const int k = s_total;
    for (int v = t; v < k; v += NT) fr[v] = (v + 1 < k ? lid[v + 1] : (uint32_t)SYMS) - lid[v];
    __syncthreads();
    for (int v = t; v < k; v += NT) lid[v] = (uint32_t)v;
}

// ---- Huffman code --------------------------------------------------------------------------------------------
// sfreq/order: leaves sorted by frequency (stable => ties keep ascending symbol order). One lane per batch walks the
// classic two-queue construction; the work per batch is O(distinct symbols) and batches run side by side.
struct TreeScratch {
    uint64_t *ifreq;      // [nB][SYMS] internal node frequencies
    int32_t *left, *right;// [nB][SYMS] children of internal node j (node id = k + j)
    uint32_t *prefix;     // [nB][2*SYMS]
    int32_t *depth;       // [nB][2*SYMS]
};

__global__ void __launch_bounds__(64) k_enc_tree(const uint32_t *sfreq, const uint32_t *order, const int32_t *num_syms, TreeScratch ts,
                                                 uint32_t *code_cw, int32_t *code_len)
{
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x;
    const int k = num_syms[b];
    const uint32_t *lf = sfreq + (size_t)b * SYMS;
    const uint32_t *ord = order + (size_t)b * SYMS;
    uint32_t *cw = code_cw + (size_t)b * SYMS;
    int32_t *cl = code_len + (size_t)b * SYMS;
    if (k == 1) { cw[0] = 0; cl[0] = 1; return; }       // degenerate batch: one bit (pcr_encoder.cpp build_code)
    uint64_t *inf = ts.ifreq + (size_t)b * SYMS;
    int32_t *L = ts.left + (size_t)b * SYMS, *R = ts.right + (size_t)b * SYMS;
    uint32_t *pf = ts.prefix + (size_t)b * 2 * SYMS;
    int32_t *dp = ts.depth + (size_t)b * 2 * SYMS;
    // leaves are nodes [0,k), internal nodes [k, 2k-1) in creation order (non-decreasing frequency)
    int qa = 0, qb = 0, nb = 0;                          // qb, nb count internal nodes
    uint64_t fa = lf[0], fb = 0;                         // cached heads of the two queues
    while ((k - qa) + (nb - qb) > 1) {
        int32_t pick[2];
        uint64_t f[2];
        for (int r = 0; r < 2; ++r) {
            if (qa < k && (qb >= nb || fa <= fb)) {      // huffman.h:58-61 comparator: leaves first on ties
                pick[r] = qa; f[r] = fa; ++qa;
                if (qa < k) fa = lf[qa];
            } else {
                pick[r] = k + qb; f[r] = fb; ++qb;
                if (qb < nb) fb = inf[qb];
            }
        }
        inf[nb] = f[0] + f[1];
        L[nb] = pick[1]; R[nb] = pick[0];               // huffman.h:62-69: left = second popped
        if (qb == nb) fb = inf[nb];                      // the new node is now the head of the internal queue
        ++nb;
    }
    // depths and 12-bit prefixes, parents before children: left = 0, right = 1 (huffman.h:183-195)
    const int root = k + nb - 1;
    pf[root] = 0; dp[root] = 0;
    for (int id = root; id >= k; --id) {
        const int d = dp[id];
        const uint32_t p = pf[id];
        const int l = L[id - k], r = R[id - k];
        pf[l] = d < PCR_MAX_CW_LEN ? (p << 1) : p;        dp[l] = d + 1;
        pf[r] = d < PCR_MAX_CW_LEN ? ((p << 1) | 1u) : p; dp[r] = d + 1;
    }
    for (int i = 0; i < k; ++i) {
        const uint32_t u = ord[i];
        cw[u] = pf[i];
        cl[u] = dp[i] <= PCR_MAX_CW_LEN ? dp[i] : -PCR_MAX_CW_LEN;             // huffman.h:206-209
    }
}

// huffman.h:220-240; symbols are visited in ascending order there, so the largest symbol index owns a shared escape key
__global__ void __launch_bounds__(NT) k_enc_table(const int32_t *syms, const uint32_t *code_cw, const int32_t *code_len,
                                                  const int32_t *num_syms, int32_t *tv, int32_t *tl, int *error)
{
    const int b = blockIdx.x, t = threadIdx.x;
    __shared__ int owner[PCR_HUFFMAN_TABLE_SIZE];
    for (int e = t; e < PCR_HUFFMAN_TABLE_SIZE; e += NT) owner[e] = -1;
    __syncthreads();
    const int k = num_syms[b];
    const uint32_t *cw = code_cw + (size_t)b * SYMS;
    const int32_t *cl = code_len + (size_t)b * SYMS;
    for (int u = t; u < k; u += NT) {
        const int alen = abs(cl[u]);
        if (alen < 1 || alen > PCR_MAX_CW_LEN) { atomicExch(error, 1); continue; }
        const int rem = PCR_MAX_CW_LEN - alen;
        const uint32_t base = cw[u] << rem;
        for (uint32_t m = 0; m < (1u << rem); ++m) {
            if (base + m >= PCR_HUFFMAN_TABLE_SIZE) { atomicExch(error, 1); break; }
            atomicMax(&owner[base + m], u);
        }
    }
    __syncthreads();
    const int32_t *sy = syms + (size_t)b * SYMS;
    for (int e = t; e < PCR_HUFFMAN_TABLE_SIZE; e += NT) {
        int o = owner[e];
        if (k == 1) o = 0;                                                    // degenerate: every key valid
        if (o < 0) { atomicExch(error, 2); continue; }                        // huffman.h:237
        tv[(size_t)b * PCR_HUFFMAN_TABLE_SIZE + e] = sy[o];
        tl[(size_t)b * PCR_HUFFMAN_TABLE_SIZE + e] = cl[o];
    }
}

// ---- packing -------------------------------------------------------------------------------------------------
// One thread per chain (huffman.h:242-300): words, completion times, escapes into fixed-stride scratch.
__global__ void __launch_bounds__(NT) k_enc_pack(const int32_t *deltas, const int32_t *syms, const uint32_t *code_cw,
                                                 const int32_t *code_len, const int32_t *num_syms, uint32_t *wbuf, uint8_t *tbuf,
                                                 int32_t *ebuf, int32_t *chain_nw, int32_t *chain_bits, int32_t *sep_sizes,
                                                 int32_t *batch_sep)
{
    const int b = blockIdx.x, c = threadIdx.x;
    const int k = num_syms[b];
    const int32_t *sy = syms + (size_t)b * SYMS;
    const uint32_t *cw = code_cw + (size_t)b * SYMS;
    const int32_t *cl = code_len + (size_t)b * SYMS;
    const int32_t *d = deltas + (size_t)b * SYMS + (size_t)c * CHAIN_SYMS;
    const size_t chain = (size_t)b * NT + c;
    uint32_t *w = wbuf + chain * MAXW;
    uint8_t *tm = tbuf + chain * MAXW;
    int32_t *e = ebuf + chain * CHAIN_SYMS;
    uint64_t acc = 0;
    int fill = 0, nw = 0, nsep = 0;
    for (int i = 0; i < CHAIN_SYMS; ++i) {
        const int32_t s = d[i];
        int lo = 0, hi = k - 1;                                               // the symbol is in the alphabet
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sy[mid] < s) lo = mid + 1; else hi = mid; }
        const int32_t len = cl[lo];
        const int alen = abs(len);
        if (len < 0) e[nsep++] = s;
        acc |= (uint64_t)cw[lo] << (64 - fill - alen);
        fill += alen;
        if (fill >= 32) {
            w[nw] = (uint32_t)(acc >> 32); tm[nw] = (uint8_t)(i + 1); ++nw;
            acc <<= 32; fill -= 32;
        }
    }
    const int bits = nw * 32 + fill;
    if (fill > 0) { w[nw] = (uint32_t)(acc >> 32); tm[nw] = (uint8_t)CHAIN_SYMS; ++nw; }
    chain_nw[chain] = nw; chain_bits[chain] = bits;
    typedef hipcub::BlockScan<int, NT> Scan;
    __shared__ typename Scan::TempStorage tmp;
    int incl, total;
    Scan(tmp).InclusiveSum(nsep, incl, total);
    sep_sizes[chain] = incl;                                                  // inclusive prefix, preprocess.cpp:1105-1111
    if (c == 0) batch_sep[b] = total;
}

// Four blocks of 8 clusters per batch. Slot (time, lane) -> position inside the cluster (preprocess.cpp:552-573).
__global__ void __launch_bounds__(256) k_enc_interleave(const uint32_t *wbuf, const uint8_t *tbuf, const int32_t *chain_nw,
                                                        const int32_t *chain_bits, int pad_tails, uint32_t *cbuf, int32_t *csize,
                                                        int *error)
{
    const int b = blockIdx.x >> 2, cl = (blockIdx.x & 3) * 8 + (threadIdx.x >> 5), lane = threadIdx.x & 31, lc = threadIdx.x >> 5;
    __shared__ uint8_t H[8][TIMES][PCR_CLUSTER_LANES];        // slots per (time + 1, lane): 0, 1 or 2
    __shared__ uint16_t Rw[8][TIMES + 1];                      // words before time row t
    for (int i = threadIdx.x; i < 8 * TIMES * PCR_CLUSTER_LANES; i += 256) (&H[0][0][0])[i] = 0;
    __syncthreads();
    const size_t chain = (size_t)b * NT + (size_t)cl * 32 + lane;
    const int nw = chain_nw[chain];
    if (nw < 2) atomicExch(error, 3);                          // a chain always has >= 6 words (192 symbols x >= 1 bit)
    const uint8_t *tm = tbuf + chain * MAXW;
    const uint32_t *w = wbuf + chain * MAXW;
    int nslots = nw;
    if (pad_tails) {
        const int full = chain_bits[chain] % 32 == 0 ? nw : nw - 1;          // words that run dry
        nslots = max(nw, full + 2);
    }
    auto slot_time = [&](int i) -> int { return i == 0 ? -1 : i == 1 ? 0 : (int)tm[i - 2]; };
    for (int i = 0; i < nslots; ++i) H[lc][slot_time(i) + 1][lane] += 1;
    __syncthreads();
    // row totals -> exclusive prefix over time
    for (int t = lane; t < TIMES; t += 32) {
        int s = 0;
        for (int l = 0; l < PCR_CLUSTER_LANES; ++l) s += H[lc][t][l];
        Rw[lc][t + 1] = (uint16_t)s;
    }
    __syncthreads();
    if (lane == 0) {
        int run = 0;
        Rw[lc][0] = 0;
        for (int t = 1; t <= TIMES; ++t) { run += Rw[lc][t]; Rw[lc][t] = (uint16_t)run; }
        csize[b * PCR_CLUSTERS_PER_BATCH + cl] = run;
    }
    __syncthreads();
    uint32_t *out = cbuf + ((size_t)b * PCR_CLUSTERS_PER_BATCH + cl) * CL_WORDS;
    int prev_t = -2, same = 0;
    for (int i = 0; i < nslots; ++i) {
        const int t = slot_time(i);
        same = t == prev_t ? same + 1 : 0;
        prev_t = t;
        int pos = Rw[lc][t + 1] + same;
        for (int l = 0; l < lane; ++l) pos += H[lc][t + 1][l];
        out[pos] = i < nw ? w[i] : 0u;
    }
}

// cluster_sizes (inclusive prefix inside the batch, preprocess.cpp:584-586) and the batch's words back to back
__global__ void __launch_bounds__(NT) k_enc_compact(const uint32_t *cbuf, const int32_t *csize, const int32_t *ebuf,
                                                    const int32_t *sep_sizes, int32_t *cluster_sizes, uint32_t *enc, int32_t *batch_enc,
                                                    int32_t *sep)
{
    const int b = blockIdx.x, t = threadIdx.x;
    __shared__ int s_off[PCR_CLUSTERS_PER_BATCH + 1];
    if (t == 0) {
        int run = 0;
        s_off[0] = 0;
        for (int c = 0; c < PCR_CLUSTERS_PER_BATCH; ++c) {
            run += csize[b * PCR_CLUSTERS_PER_BATCH + c];
            s_off[c + 1] = run;
            cluster_sizes[b * PCR_CLUSTERS_PER_BATCH + c] = run;
        }
        batch_enc[b] = run;
    }
    __syncthreads();
    uint32_t *eo = enc + (size_t)b * PCR_CLUSTERS_PER_BATCH * CL_WORDS;
    for (int c = 0; c < PCR_CLUSTERS_PER_BATCH; ++c) {
        const uint32_t *src = cbuf + ((size_t)b * PCR_CLUSTERS_PER_BATCH + c) * CL_WORDS;
        const int n = s_off[c + 1] - s_off[c];
        for (int i = t; i < n; i += NT) eo[s_off[c] + i] = src[i];
    }
    // escapes of chain t, after those of chains 0..t-1
    const size_t chain = (size_t)b * NT + t;
    const int end = sep_sizes[chain], begin = t ? sep_sizes[chain - 1] : 0;
    int32_t *so = sep + (size_t)b * SYMS;
    const int32_t *e = ebuf + chain * CHAIN_SYMS;
    for (int i = 0; i < end - begin; ++i) so[begin + i] = e[i];
}

__global__ void k_enc_bc1(const uint32_t *color, int64_t nblocks, uint8_t *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    uint32_t px[16];
    for (int k = 0; k < 16; ++k) px[k] = color[i * 16 + k];
    uint8_t o[8];
    pcr_codec::bc1_encode(px, o);
    for (int k = 0; k < 8; ++k) out[i * 8 + k] = o[k];
}

// ---- host ------------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <class T> T *as() const { return (T *)p; }
};

inline void put(std::vector<uint8_t> &buf, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    buf.insert(buf.end(), b, b + n);
}

} // namespace enc
} // namespace pcr

extern "C" void pcr_gpu_encode_free(void *p) { std::free(p); }

extern "C" int pcr_gpu_encode_points(pcr_ctx *c, const int32_t *x, const int32_t *y, const int32_t *z, const uint32_t *color,
                                     int64_t n, const pcr_las_info *las, int flags, int64_t chunk_points,
                                     void **out_bytes, size_t *out_len, pcr_encode_stats *stats)
{
    using namespace pcr::enc;
    if (!c) return PCR_E_ARG;
    if (!x || !y || !z || !color || !las || !out_bytes || !out_len) return set_err(c, PCR_E_ARG, "null argument");
    if (n <= 0) return set_err(c, PCR_E_ARG, "no points");
    if (flags & PCR_ENCODE_BC7) return set_err(c, PCR_E_ARG, "BC7 colours are written by the CPU encoder only (pcr_encode_points)");
    if (chunk_points <= 0) chunk_points = PCR_DEFAULT_CHUNK_POINTS;
    if (chunk_points % NPB) return set_err(c, PCR_E_ARG, "chunk_points must be a multiple of %d", NPB);
    if (chunk_points / NPB > 1024)      // scratch is sized per chunk (16 MB per batch) and the sorts index it with 32-bit ints
        return set_err(c, PCR_E_ARG, "chunk_points must not exceed %d (1024 batches)", 1024 * NPB);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const bool sort = (flags & PCR_ENCODE_MORTON_SORT) != 0;
    const int pad_tails = (flags & PCR_ENCODE_PAD_TAILS) ? 1 : 0;
    const int64_t maxB = chunk_points / NPB, maxN = chunk_points;

    // ---- device scratch for one chunk -------------------------------------------------------------------------
    DevBuf in_x, in_y, in_z, in_c, px, py, pz, pc, klo, klo2, khi, khi2, idx, idx2, idx3, start, deltas, sorted, syms, freqs, leaf,
        sfreq, order, nsyms, segb, sege, mnmx, ccw, clen, ifreq, left, right, prefix, depth, tv, tl, wbuf, tbuf, ebuf, cnw, cbits, sepsz,
        bsep, cbuf, csize, clsz, enc, benc, sep, bc1, err, tmp;
    auto alloc = [&](DevBuf &d, size_t bytes) -> int {
        HIP_TRY(c, hipMalloc(&d.p, bytes ? bytes : 1));
        return PCR_OK;
    };
    int rc;
#define A(buf, bytes) if ((rc = alloc(buf, (size_t)(bytes)))) return rc
    A(in_x, maxN * 4); A(in_y, maxN * 4); A(in_z, maxN * 4); A(in_c, maxN * 4);
    A(px, maxN * 4); A(py, maxN * 4); A(pz, maxN * 4); A(pc, maxN * 4);
    if (sort) { A(klo, maxN * 8); A(klo2, maxN * 8); A(khi, maxN * 4); A(khi2, maxN * 4); A(idx, maxN * 4); A(idx2, maxN * 4); A(idx3, maxN * 4); }
    A(start, maxB * NT * 3 * 4); A(deltas, maxB * SYMS * 4); A(sorted, maxB * SYMS * 4); A(syms, maxB * SYMS * 4);
    A(freqs, maxB * SYMS * 4); A(leaf, maxB * SYMS * 4); A(sfreq, maxB * SYMS * 4); A(order, maxB * SYMS * 4);
    A(nsyms, maxB * 4); A(segb, maxB * 4); A(sege, maxB * 4); A(mnmx, maxB * 6 * 4);
    A(ccw, maxB * SYMS * 4); A(clen, maxB * SYMS * 4);
    A(ifreq, maxB * SYMS * 8); A(left, maxB * SYMS * 4); A(right, maxB * SYMS * 4); A(prefix, maxB * SYMS * 8); A(depth, maxB * SYMS * 8);
    A(tv, maxB * PCR_HUFFMAN_TABLE_SIZE * 4); A(tl, maxB * PCR_HUFFMAN_TABLE_SIZE * 4);
    A(wbuf, maxB * NT * MAXW * 4); A(tbuf, maxB * NT * MAXW); A(ebuf, maxB * NT * CHAIN_SYMS * 4);
    A(cnw, maxB * NT * 4); A(cbits, maxB * NT * 4); A(sepsz, maxB * NT * 4); A(bsep, maxB * 4);
    A(cbuf, maxB * PCR_CLUSTERS_PER_BATCH * CL_WORDS * 4); A(csize, maxB * PCR_CLUSTERS_PER_BATCH * 4);
    A(clsz, maxB * PCR_CLUSTERS_PER_BATCH * 4); A(enc, maxB * PCR_CLUSTERS_PER_BATCH * CL_WORDS * 4); A(benc, maxB * 4);
    A(sep, maxB * SYMS * 4); A(bc1, maxB * PCR_COLOR_BYTES_PER_BATCH); A(err, 4);
    // hipcub temp storage: the largest of the four sorts
    size_t t1 = 0, t2 = 0, t3 = 0, t4 = 0;
    if (sort) {
        HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, t1, klo.as<uint64_t>(), klo2.as<uint64_t>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), (int)maxN, 0, 64, st));
        HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, t2, khi.as<uint32_t>(), khi2.as<uint32_t>(), idx2.as<uint32_t>(), idx3.as<uint32_t>(), (int)maxN, 0, 32, st));
    }
    HIP_TRY(c, hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, t3, deltas.as<int32_t>(), sorted.as<int32_t>(), (int)(maxB * SYMS), (int)maxB,
                                                           segb.as<int32_t>(), sege.as<int32_t>(), 0, 32, st));
    HIP_TRY(c, hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, t4, freqs.as<uint32_t>(), sfreq.as<uint32_t>(), leaf.as<uint32_t>(), order.as<uint32_t>(),
                                                            (int)(maxB * SYMS), (int)maxB, segb.as<int32_t>(), sege.as<int32_t>(), 0, 32, st));
    size_t tmp_bytes = std::max(std::max(t1, t2), std::max(t3, t4));
    A(tmp, tmp_bytes);
#undef A
    HIP_TRY(c, hipMemsetAsync(err.p, 0, 4, st));

    // ---- chunks -----------------------------------------------------------------------------------------------
    struct ChunkOut { std::vector<uint8_t> bytes; std::vector<int64_t> batch_sizes; int64_t points = 0, enc_words = 0, sep_words = 0; };
    const int64_t nchunks = (n + chunk_points - 1) / chunk_points;
    std::vector<ChunkOut> chunks((size_t)nchunks);
    std::vector<int32_t> h_start, h_sepsz, h_tv, h_tl, h_clsz, h_benc, h_bsep, h_mnmx, h_sep;
    std::vector<uint32_t> h_enc;
    std::vector<uint8_t> h_bc1;
    for (int64_t ch = 0; ch < nchunks; ++ch) {
        const int64_t a = ch * chunk_points, real = std::min(n, a + chunk_points) - a;
        const int64_t np = (real + NPB - 1) / NPB * NPB;                      // padded, preprocess.cpp:945-955
        const int nB = (int)(np / NPB);
        HIP_TRY(c, hipMemcpyAsync(in_x.p, x + a, (size_t)real * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(in_y.p, y + a, (size_t)real * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(in_z.p, z + a, (size_t)real * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemcpyAsync(in_c.p, color + a, (size_t)real * 4, hipMemcpyHostToDevice, st));
        const unsigned gp = (unsigned)((np + 255) / 256);
        const uint32_t *perm = nullptr;
        if (sort) {
            hipLaunchKernelGGL(k_enc_keys, dim3(gp), dim3(256), 0, st, in_x.as<int32_t>(), in_y.as<int32_t>(), in_z.as<int32_t>(), real, np,
                               klo.as<uint64_t>(), khi.as<uint32_t>(), idx.as<uint32_t>());
            // stable LSD: low 64 bits first, then the high 32 bits
            size_t tb = tmp_bytes;
            HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, klo.as<uint64_t>(), klo2.as<uint64_t>(), idx.as<uint32_t>(), idx2.as<uint32_t>(), (int)np, 0, 64, st));
            hipLaunchKernelGGL(k_enc_gather_u32, dim3(gp), dim3(256), 0, st, khi.as<uint32_t>(), idx2.as<uint32_t>(), np, khi2.as<uint32_t>());
            tb = tmp_bytes;
            HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, khi2.as<uint32_t>(), khi.as<uint32_t>(), idx2.as<uint32_t>(), idx3.as<uint32_t>(), (int)np, 0, 32, st));
            perm = idx3.as<uint32_t>();
        }
        hipLaunchKernelGGL(k_enc_gather_points, dim3(gp), dim3(256), 0, st, in_x.as<int32_t>(), in_y.as<int32_t>(), in_z.as<int32_t>(), in_c.as<uint32_t>(),
                           perm, real, np, px.as<int32_t>(), py.as<int32_t>(), pz.as<int32_t>(), pc.as<uint32_t>());
        hipLaunchKernelGGL(k_enc_deltas, dim3(nB), dim3(NT), 0, st, px.as<int32_t>(), py.as<int32_t>(), pz.as<int32_t>(), start.as<int32_t>(),
                           deltas.as<int32_t>(), mnmx.as<int32_t>());
        // alphabet: sort each batch's symbols, count runs
        {
            std::vector<int32_t> hb((size_t)nB), he((size_t)nB);
            for (int b = 0; b < nB; ++b) { hb[(size_t)b] = b * SYMS; he[(size_t)b] = (b + 1) * SYMS; }
            HIP_TRY(c, hipMemcpyAsync(segb.p, hb.data(), (size_t)nB * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(c, hipMemcpyAsync(sege.p, he.data(), (size_t)nB * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(c, hipStreamSynchronize(st));                             // hb/he are about to go out of scope
        }
        size_t tb = tmp_bytes;
        HIP_TRY(c, hipcub::DeviceSegmentedRadixSort::SortKeys(tmp.p, tb, deltas.as<int32_t>(), sorted.as<int32_t>(), nB * SYMS, nB,
                                                               segb.as<int32_t>(), sege.as<int32_t>(), 0, 32, st));
        hipLaunchKernelGGL(k_enc_unique, dim3(nB), dim3(NT), 0, st, sorted.as<int32_t>(), syms.as<int32_t>(), freqs.as<uint32_t>(), leaf.as<uint32_t>(),
                           nsyms.as<int32_t>(), segb.as<int32_t>(), sege.as<int32_t>());
        tb = tmp_bytes;
        HIP_TRY(c, hipcub::DeviceSegmentedRadixSort::SortPairs(tmp.p, tb, freqs.as<uint32_t>(), sfreq.as<uint32_t>(), leaf.as<uint32_t>(), order.as<uint32_t>(),
                                                                nB * SYMS, nB, segb.as<int32_t>(), sege.as<int32_t>(), 0, 32, st));
        TreeScratch ts{ifreq.as<uint64_t>(), left.as<int32_t>(), right.as<int32_t>(), prefix.as<uint32_t>(), depth.as<int32_t>()};
        hipLaunchKernelGGL(k_enc_tree, dim3(nB), dim3(64), 0, st, sfreq.as<uint32_t>(), order.as<uint32_t>(), nsyms.as<int32_t>(), ts,
                           ccw.as<uint32_t>(), clen.as<int32_t>());
        hipLaunchKernelGGL(k_enc_table, dim3(nB), dim3(NT), 0, st, syms.as<int32_t>(), ccw.as<uint32_t>(), clen.as<int32_t>(), nsyms.as<int32_t>(),
                           tv.as<int32_t>(), tl.as<int32_t>(), err.as<int>());
        hipLaunchKernelGGL(k_enc_pack, dim3(nB), dim3(NT), 0, st, deltas.as<int32_t>(), syms.as<int32_t>(), ccw.as<uint32_t>(), clen.as<int32_t>(),
                           nsyms.as<int32_t>(), wbuf.as<uint32_t>(), tbuf.as<uint8_t>(), ebuf.as<int32_t>(), cnw.as<int32_t>(), cbits.as<int32_t>(),
                           sepsz.as<int32_t>(), bsep.as<int32_t>());
        hipLaunchKernelGGL(k_enc_interleave, dim3(nB * 4), dim3(256), 0, st, wbuf.as<uint32_t>(), tbuf.as<uint8_t>(), cnw.as<int32_t>(),
                           cbits.as<int32_t>(), pad_tails, cbuf.as<uint32_t>(), csize.as<int32_t>(), err.as<int>());
        hipLaunchKernelGGL(k_enc_compact, dim3(nB), dim3(NT), 0, st, cbuf.as<uint32_t>(), csize.as<int32_t>(), ebuf.as<int32_t>(), sepsz.as<int32_t>(),
                           clsz.as<int32_t>(), enc.as<uint32_t>(), benc.as<int32_t>(), sep.as<int32_t>());
        hipLaunchKernelGGL(k_enc_bc1, dim3((unsigned)((np / 16 + 255) / 256)), dim3(256), 0, st, pc.as<uint32_t>(), np / 16, bc1.as<uint8_t>());
        HIP_TRY(c, hipGetLastError());

        // ---- back to the host: fixed-size parts whole, variable parts per batch --------------------------------
        h_start.resize((size_t)nB * NT * 3); h_sepsz.resize((size_t)nB * NT); h_tv.resize((size_t)nB * PCR_HUFFMAN_TABLE_SIZE);
        h_tl.resize((size_t)nB * PCR_HUFFMAN_TABLE_SIZE); h_clsz.resize((size_t)nB * PCR_CLUSTERS_PER_BATCH); h_benc.resize((size_t)nB);
        h_bsep.resize((size_t)nB); h_mnmx.resize((size_t)nB * 6); h_bc1.resize((size_t)nB * PCR_COLOR_BYTES_PER_BATCH);
        int h_err = 0;
        HIP_TRY(c, hipMemcpyAsync(h_start.data(), start.p, h_start.size() * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(h_sepsz.data(), sepsz.p, h_sepsz.size() * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(h_tv.data(), tv.p, h_tv.size() * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(h_tl.data(), tl.p, h_tl.size() * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(h_clsz.data(), clsz.p, h_clsz.size() * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(h_benc.data(), benc.p, h_benc.size() * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(h_bsep.data(), bsep.p, h_bsep.size() * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(h_mnmx.data(), mnmx.p, h_mnmx.size() * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(h_bc1.data(), bc1.p, h_bc1.size(), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(&h_err, err.p, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (h_err) return set_err(c, PCR_E_FORMAT, "internal: GPU encoder consistency check %d failed", h_err);
        size_t enc_total = 0, sep_total = 0;
        std::vector<size_t> enc_off((size_t)nB), sep_off((size_t)nB);
        for (int b = 0; b < nB; ++b) { enc_off[(size_t)b] = enc_total; enc_total += (size_t)h_benc[(size_t)b]; sep_off[(size_t)b] = sep_total; sep_total += (size_t)h_bsep[(size_t)b]; }
        h_enc.resize(enc_total); h_sep.resize(sep_total ? sep_total : 1);
        for (int b = 0; b < nB; ++b) {
            if (h_benc[(size_t)b])
                HIP_TRY(c, hipMemcpyAsync(h_enc.data() + enc_off[(size_t)b], enc.as<uint32_t>() + (size_t)b * PCR_CLUSTERS_PER_BATCH * CL_WORDS,
                                          (size_t)h_benc[(size_t)b] * 4, hipMemcpyDeviceToHost, st));
            if (h_bsep[(size_t)b])
                HIP_TRY(c, hipMemcpyAsync(h_sep.data() + sep_off[(size_t)b], sep.as<int32_t>() + (size_t)b * SYMS, (size_t)h_bsep[(size_t)b] * 4,
                                          hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(c, hipStreamSynchronize(st));

        // ---- batch records (include/BatchDumpData.h:151-202; same statements as pcr_encoder.cpp encode_batch) ---
        ChunkOut &co = chunks[(size_t)ch];
        co.points = np;
        co.bytes.reserve((size_t)np * 4);
        for (int b = 0; b < nB; ++b) {
            const size_t before = co.bytes.size();
            std::vector<uint8_t> &rec = co.bytes;
            int32_t hdr[5] = {(int32_t)((int64_t)b * NPB), NPB, NT, PPT, PCR_CLUSTERS_PER_THREAD};
            put(rec, hdr, sizeof hdr);
            put(rec, las->scale, 24);
            put(rec, las->offset, 24);
            float bmin[3], bmax[3], lmin[3], lmax[3];
            for (int k = 0; k < 3; ++k) {
                // preprocess.cpp:1082-1087: float(int) promoted to double, * scale + offset, narrowed to float
                volatile double va = (double)(float)h_mnmx[(size_t)b * 6 + k] * las->scale[k];
                volatile double vb = (double)(float)h_mnmx[(size_t)b * 6 + 3 + k] * las->scale[k];
                bmin[k] = (float)(va + las->offset[k]);
                bmax[k] = (float)(vb + las->offset[k]);
                lmin[k] = (float)las->min[k]; lmax[k] = (float)las->max[k];   // :1075-1080
            }
            put(rec, bmin, 12); put(rec, bmax, 12); put(rec, lmin, 12); put(rec, lmax, 12);
            int32_t dt_size = PCR_HUFFMAN_TABLE_SIZE, ncl = PCR_CLUSTERS_PER_BATCH;
            put(rec, &dt_size, 4); put(rec, &ncl, 4);
            put(rec, h_start.data() + (size_t)b * NT * 3, (size_t)NT * 3 * 4);
            put(rec, h_sepsz.data() + (size_t)b * NT, (size_t)NT * 4);
            put(rec, h_tv.data() + (size_t)b * PCR_HUFFMAN_TABLE_SIZE, (size_t)PCR_HUFFMAN_TABLE_SIZE * 4);
            put(rec, h_tl.data() + (size_t)b * PCR_HUFFMAN_TABLE_SIZE, (size_t)PCR_HUFFMAN_TABLE_SIZE * 4);
            put(rec, h_clsz.data() + (size_t)b * PCR_CLUSTERS_PER_BATCH, (size_t)PCR_CLUSTERS_PER_BATCH * 4);
            put(rec, h_enc.data() + enc_off[(size_t)b], (size_t)h_benc[(size_t)b] * 4);
            put(rec, h_sep.data() + sep_off[(size_t)b], (size_t)h_bsep[(size_t)b] * 4);
            put(rec, h_bc1.data() + (size_t)b * PCR_COLOR_BYTES_PER_BATCH, PCR_COLOR_BYTES_PER_BATCH);
            co.batch_sizes.push_back((int64_t)(rec.size() - before));
            co.enc_words += h_benc[(size_t)b]; co.sep_words += h_bsep[(size_t)b];
        }
    }

    // ---- file image (preprocess.cpp:1205-1234; same as pcr_encoder.cpp assemble) --------------------------------
    pcr_file_header h{};
    size_t body = 0;
    int64_t escaped = 0;
    for (auto &co : chunks) {
        h.num_points += co.points; h.num_batches += (int64_t)co.batch_sizes.size();
        h.encoded_bytes += 4 * co.enc_words; h.separate_bytes += 4 * co.sep_words;
        body += co.bytes.size(); escaped += co.sep_words;
    }
    h.cluster_bytes = 4 * PCR_CLUSTERS_PER_BATCH * h.num_batches;
    const size_t total = sizeof h + 8 * (size_t)h.num_batches + body;
    uint8_t *buf = (uint8_t *)std::malloc(total);
    if (!buf) return set_err(c, PCR_E_NOMEM, "out of memory assembling the file image");
    std::memcpy(buf, &h, sizeof h);
    size_t so = sizeof h, bo = sizeof h + 8 * (size_t)h.num_batches;
    for (auto &co : chunks) {
        std::memcpy(buf + so, co.batch_sizes.data(), 8 * co.batch_sizes.size()); so += 8 * co.batch_sizes.size();
        std::memcpy(buf + bo, co.bytes.data(), co.bytes.size()); bo += co.bytes.size();
        std::vector<uint8_t>().swap(co.bytes);
    }
    *out_bytes = buf; *out_len = total;
    if (stats) {
        stats->num_points_in = n; stats->num_points = h.num_points; stats->num_batches = h.num_batches;
        stats->encoded_bytes = h.encoded_bytes; stats->separate_bytes = h.separate_bytes;
        stats->cluster_bytes = h.cluster_bytes; stats->escaped_symbols = escaped;
        stats->total_symbols = h.num_points * 3; stats->file_bytes = (int64_t)total;
    }
    return PCR_OK;
}
