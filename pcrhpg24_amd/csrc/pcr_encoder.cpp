// pcr_encoder.cpp — host-side (CPU) .huffman encoder, synthetic scene generator and camera helper.
// Part of libpcr_host.so; C ABI in include/pcr_encode.h (which lists the reference lines each step follows).
//
// Design notes (not a transliteration of src/preprocess.cpp):
//   * one flat pass per batch over 196 608 delta symbols; frequencies via sort of a copy, not unordered_map
//   * Huffman tree from a two-queue merge over sorted leaves (O(k) after the sort) with a deterministic
//     tie-break, so streams are reproducible on any libstdc++ (the reference's depend on hash order)
//   * chunks are generated / sorted / encoded by a thread pool; batches inside a chunk are independent
#include "pcr_encode.h"
#include "pcr_codec_common.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(const char *msg) { g_err = msg; return -1; }

using pcr_codec::MortonKey; using pcr_codec::morton_key; using pcr_codec::shift_coord; using pcr_codec::bc1_encode;   // pcr_codec_common.h

// ---------------------------------------------------------------------------------------------
// Clipped Huffman code (include/huffman.h:58-69, 94-113, 180-240)
// ---------------------------------------------------------------------------------------------
struct Code { uint32_t cw; int32_t len; };   // len < 0: escape, cw = 12-bit prefix

struct HuffmanCode {
    std::vector<int32_t> symbols;      // sorted ascending
    std::vector<Code> codes;           // parallel to symbols
    const Code &lookup(int32_t s) const
    {
        size_t i = std::lower_bound(symbols.begin(), symbols.end(), s) - symbols.begin();
        return codes[i];
    }
};

// Build from (symbol, frequency) pairs sorted by symbol.
void build_code(const std::vector<int32_t> &syms, const std::vector<uint32_t> &freqs, HuffmanCode &out)
{
    const size_t k = syms.size();
    out.symbols = syms;
    out.codes.assign(k, Code{0, 0});
    if (k == 1) {             // degenerate batch (the reference asserts, huffman.h:265): give it one bit
        out.codes[0] = Code{0, 1};
        return;
    }
    struct Node { uint64_t freq; int32_t left, right; };
    std::vector<Node> nodes(2 * k - 1);
    std::vector<uint32_t> order(k);
    for (size_t i = 0; i < k; ++i) order[i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return freqs[a] < freqs[b]; });
    for (size_t i = 0; i < k; ++i) nodes[i] = Node{freqs[order[i]], -1, -1};
    // two-queue Huffman: leaves [0,k) sorted by freq, internal nodes appended in non-decreasing freq order
    size_t qa = 0, qb = k, nb = k;
    auto pop = [&]() -> int32_t {
        if (qa < k && (qb >= nb || nodes[qa].freq <= nodes[qb].freq)) return (int32_t)qa++;
        return (int32_t)qb++;
    };
    while ((k - qa) + (nb - qb) > 1) {
        int32_t a = pop(), b = pop();
        nodes[nb++] = Node{nodes[a].freq + nodes[b].freq, b, a};   // huffman.h:62-69: left = second popped
    }
    // iterative walk: left = 0, right = 1 (huffman.h:183-195)
    struct Item { int32_t node; uint32_t prefix; int depth; };
    std::vector<Item> stack;
    stack.push_back(Item{(int32_t)(nb - 1), 0, 0});
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        const Node &n = nodes[it.node];
        if (n.left < 0) {
            Code c;
            if (it.depth <= PCR_MAX_CW_LEN) c = Code{it.prefix, it.depth};
            else                            c = Code{it.prefix, -PCR_MAX_CW_LEN};       // huffman.h:206-209
            out.codes[order[it.node]] = c;
            continue;
        }
        // prefix keeps only the first 12 bits of the path
        auto ext = [&](int bit) { return it.depth < PCR_MAX_CW_LEN ? ((it.prefix << 1) | (uint32_t)bit) : it.prefix; };
        stack.push_back(Item{n.right, ext(1), it.depth + 1});
        stack.push_back(Item{n.left, ext(0), it.depth + 1});
    }
}

// huffman.h:220-240
int table_from_codes(const int32_t *syms, const Code *codes, size_t k, int32_t *tv, int32_t *tl)
{
    std::vector<char> touched(PCR_HUFFMAN_TABLE_SIZE, 0);
    for (size_t i = 0; i < k; ++i) {
        int alen = std::abs(codes[i].len);
        if (alen < 1 || alen > PCR_MAX_CW_LEN) return -1;
        int rem = PCR_MAX_CW_LEN - alen;
        uint32_t base = codes[i].cw << rem;
        for (uint32_t m = 0; m < (1u << rem); ++m) {
            if (base + m >= PCR_HUFFMAN_TABLE_SIZE) return -1;
            tv[base + m] = syms[i]; tl[base + m] = codes[i].len; touched[base + m] = 1;
        }
    }
    if (k == 1) {   // degenerate: make every key valid
        for (int i = 0; i < PCR_HUFFMAN_TABLE_SIZE; ++i) { tv[i] = syms[0]; tl[i] = codes[0].len; touched[i] = 1; }
    }
    for (char t : touched) if (!t) return -2;   // huffman.h:237
    return 0;
}

// huffman.h:242-300: MSB-first packing, escape list, per-word completion index
struct PackedChain {
    int total_bits = 0;
    std::vector<uint32_t> words;
    std::vector<int32_t> separate;
    std::vector<int32_t> num_cw;
};

template <class Lookup>
void pack_chain(const int32_t *symbols, int n, Lookup &&lookup, PackedChain &out)
{
    out.words.clear(); out.separate.clear(); out.num_cw.clear();
    uint64_t acc = 0;   // bits accumulate at the top of a 64-bit register
    int fill = 0;       // number of valid bits in acc (from the MSB side)
    for (int i = 0; i < n; ++i) {
        const Code &c = lookup(symbols[i]);
        int len = std::abs(c.len);
        if (c.len < 0) out.separate.push_back(symbols[i]);
        acc |= (uint64_t)c.cw << (64 - fill - len);
        fill += len;
        if (fill >= 32) {
            out.words.push_back((uint32_t)(acc >> 32));
            out.num_cw.push_back(i + 1);
            acc <<= 32; fill -= 32;
        }
    }
    out.total_bits = (int)out.words.size() * 32 + fill;
    if (fill > 0) {
        out.words.push_back((uint32_t)(acc >> 32));
        out.num_cw.push_back(n);
    }
}

// ---------------------------------------------------------------------------------------------
// Batch encoder
// ---------------------------------------------------------------------------------------------
struct BatchStats { int64_t enc_words = 0, sep_words = 0, escaped = 0; };

void put(std::vector<uint8_t> &buf, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    buf.insert(buf.end(), b, b + n);
}

// x,y,z,color: 65536 points of this batch in final order. Appends one batch record to `rec`.
int encode_batch(const int32_t *x, const int32_t *y, const int32_t *z, const uint32_t *color,
                 int32_t point_offset, const pcr_las_info &las, bool pad_tails, bool bc7, std::vector<uint8_t> &rec, BatchStats &st)
{
    const int NT = PCR_WORKGROUP_SIZE, PPT = PCR_POINTS_PER_THREAD, N = PCR_POINTS_PER_BATCH;
    std::vector<int32_t> deltas((size_t)N * 3);
    std::vector<int32_t> start(NT * 3);
    int32_t mn[3] = {x[0], y[0], z[0]}, mx[3] = {x[0], y[0], z[0]};
    for (int c = 0; c < NT; ++c) {
        const int b = c * PPT;
        start[c * 3 + 0] = x[b]; start[c * 3 + 1] = y[b]; start[c * 3 + 2] = z[b];
        int32_t *d = &deltas[(size_t)b * 3];
        d[0] = d[1] = d[2] = 0;                                   // preprocess.cpp:328
        for (int i = 1; i < PPT; ++i) {                           // :323-327 (int32 wrap as the GPU adds back)
            d[i * 3 + 0] = (int32_t)((uint32_t)x[b + i] - (uint32_t)x[b + i - 1]);
            d[i * 3 + 1] = (int32_t)((uint32_t)y[b + i] - (uint32_t)y[b + i - 1]);
            d[i * 3 + 2] = (int32_t)((uint32_t)z[b + i] - (uint32_t)z[b + i - 1]);
        }
    }
    for (int i = 0; i < N; ++i) {
        mn[0] = std::min(mn[0], x[i]); mx[0] = std::max(mx[0], x[i]);
        mn[1] = std::min(mn[1], y[i]); mx[1] = std::max(mx[1], y[i]);
        mn[2] = std::min(mn[2], z[i]); mx[2] = std::max(mx[2], z[i]);
    }

    // frequencies
    std::vector<int32_t> sorted(deltas);
    std::sort(sorted.begin(), sorted.end());
    std::vector<int32_t> syms; std::vector<uint32_t> freqs;
    for (size_t i = 0; i < sorted.size();) {
        size_t j = i;
        while (j < sorted.size() && sorted[j] == sorted[i]) ++j;
        syms.push_back(sorted[i]); freqs.push_back((uint32_t)(j - i));
        i = j;
    }
    HuffmanCode hc;
    build_code(syms, freqs, hc);
    std::vector<int32_t> tv(PCR_HUFFMAN_TABLE_SIZE), tl(PCR_HUFFMAN_TABLE_SIZE);
    if (table_from_codes(hc.symbols.data(), hc.codes.data(), hc.symbols.size(), tv.data(), tl.data()) != 0)
        return fail("internal: decoder table incomplete");

    // fast symbol -> code map: dense window around 0 plus binary search for the rest
    const int32_t DW = 1 << 12;
    std::vector<Code> dense(2 * DW, Code{0, 0});
    for (size_t i = 0; i < hc.symbols.size(); ++i) {
        int64_t s = hc.symbols[i];
        if (s >= -DW && s < DW) dense[(size_t)(s + DW)] = hc.codes[i];
    }
    auto lookup = [&](int32_t s) -> const Code & {
        if (s >= -DW && s < DW) return dense[(size_t)(s + DW)];
        return hc.lookup(s);
    };

    // per chain packing + (time, lane) interleave per 32-lane cluster (preprocess.cpp:540-587)
    std::vector<uint32_t> encoding;
    std::vector<int32_t> separate, separate_sizes(NT), cluster_sizes(PCR_CLUSTERS_PER_BATCH);
    std::vector<PackedChain> pc(PCR_CLUSTER_LANES);
    struct Slot { int32_t time; int32_t lane; int32_t word; };
    std::vector<Slot> slots;
    for (int cl = 0; cl < PCR_CLUSTERS_PER_BATCH; ++cl) {
        slots.clear();
        for (int l = 0; l < PCR_CLUSTER_LANES; ++l) {
            int chain = cl * PCR_CLUSTER_LANES + l;
            pack_chain(&deltas[(size_t)chain * PPT * 3], PPT * 3, lookup, pc[l]);
            separate.insert(separate.end(), pc[l].separate.begin(), pc[l].separate.end());
            separate_sizes[chain] = (int32_t)separate.size();              // inclusive prefix, :1105-1111
            st.escaped += (int64_t)pc[l].separate.size();
            const int nw = (int)pc[l].words.size();
            // :553-556 pushes words 0 and 1 unconditionally; a chain always has >= 2 words here
            // (192 symbols x >= 1 bit = 6 words minimum), checked below.
            if (nw < 2) return fail("internal: chain shorter than two words");
            slots.push_back(Slot{-1, l, 0});
            slots.push_back(Slot{0, l, 1});
            for (int i = 2; i < nw; ++i) slots.push_back(Slot{pc[l].num_cw[i - 2], l, i});   // :558-563
            if (pad_tails) {
                // PCR_ENCODE_PAD_TAILS (not in the reference): the decoder refills every time a word runs dry, also
                // for the last one or two words of a chain, for which the reference queues nothing. Queue
                // a zero word for each of those refills so every lane's fetch order stays the one the decoder follows.
                const int bits_total = pc[l].total_bits;
                const int full = bits_total % 32 == 0 ? nw : nw - 1;          // words that run dry
                for (int i = std::max(nw, 2); i < full + 2; ++i) slots.push_back(Slot{pc[l].num_cw[i - 2], l, i});
            }
        }
        std::sort(slots.begin(), slots.end(), [](const Slot &a, const Slot &b) {               // :564
            if (a.time != b.time) return a.time < b.time;
            if (a.lane != b.lane) return a.lane < b.lane;
            return a.word < b.word;
        });
        for (const Slot &s : slots) encoding.push_back(s.word < (int)pc[s.lane].words.size() ? pc[s.lane].words[s.word] : 0u);
        cluster_sizes[cl] = (int32_t)encoding.size();                       // inclusive prefix, :584-586
    }
    st.enc_words = (int64_t)encoding.size();
    st.sep_words = (int64_t)separate.size();

    // BC1 colours, chain-major == point order (preprocess.cpp:1123-1128); or BC7 mode 6 (:1129-1138, COLOR_COMPRESSION == 7)
    std::vector<uint8_t> bc1(bc7 ? PCR_COLOR_BYTES_PER_BATCH_BC7 : PCR_COLOR_BYTES_PER_BATCH);
    if (bc7) for (int blk = 0; blk < N / 16; ++blk) pcr_codec::bc7_mode6_encode(color + blk * 16, &bc1[(size_t)blk * 16]);
    else     for (int blk = 0; blk < N / 16; ++blk) bc1_encode(color + blk * 16, &bc1[(size_t)blk * 8]);

    // record (include/BatchDumpData.h:151-202)
    int32_t hdr[5] = {point_offset, N, NT, PPT, PCR_CLUSTERS_PER_THREAD};
    put(rec, hdr, sizeof hdr);
    put(rec, las.scale, 24);
    put(rec, las.offset, 24);
    float bmin[3], bmax[3], lmin[3], lmax[3];
    for (int k = 0; k < 3; ++k) {
        // preprocess.cpp:1082-1087: float(int) promoted to double, * scale + offset, narrowed to float
        volatile double a = (double)(float)mn[k] * las.scale[k];
        volatile double b = (double)(float)mx[k] * las.scale[k];
        bmin[k] = (float)(a + las.offset[k]);
        bmax[k] = (float)(b + las.offset[k]);
        lmin[k] = (float)las.min[k]; lmax[k] = (float)las.max[k];       // :1075-1080
    }
    put(rec, bmin, 12); put(rec, bmax, 12); put(rec, lmin, 12); put(rec, lmax, 12);
    int32_t dt_size = PCR_HUFFMAN_TABLE_SIZE, ncl = PCR_CLUSTERS_PER_BATCH;
    put(rec, &dt_size, 4); put(rec, &ncl, 4);
    put(rec, start.data(), start.size() * 4);
    put(rec, separate_sizes.data(), separate_sizes.size() * 4);
    put(rec, tv.data(), tv.size() * 4);
    put(rec, tl.data(), tl.size() * 4);
    put(rec, cluster_sizes.data(), cluster_sizes.size() * 4);
    put(rec, encoding.data(), encoding.size() * 4);
    put(rec, separate.data(), separate.size() * 4);
    put(rec, bc1.data(), bc1.size());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Chunk encoder (process_chunk, src/preprocess.cpp:925-1165)
// ---------------------------------------------------------------------------------------------
struct ChunkOut {
    std::vector<uint8_t> bytes;
    std::vector<int64_t> batch_sizes;
    int64_t points = 0, enc_words = 0, sep_words = 0, escaped = 0;
};

int encode_chunk(std::vector<int32_t> &x, std::vector<int32_t> &y, std::vector<int32_t> &z,
                 std::vector<uint32_t> &c, const pcr_las_info &las, int flags, ChunkOut &out)
{
    const bool sort = (flags & PCR_ENCODE_MORTON_SORT) != 0, pad_tails = (flags & PCR_ENCODE_PAD_TAILS) != 0, bc7 = (flags & PCR_ENCODE_BC7) != 0;
    if (x.empty()) return fail("empty chunk");
    size_t n = x.size();
    size_t extra = (n % PCR_POINTS_PER_BATCH) ? PCR_POINTS_PER_BATCH - (n % PCR_POINTS_PER_BATCH) : 0;
    x.resize(n + extra, x.back()); y.resize(n + extra, y.back());         // :945-955
    z.resize(n + extra, z.back()); c.resize(n + extra, c.back());
    n += extra;
    if (sort) {                                                           // :959-977
        struct KI { MortonKey k; uint32_t i; };
        std::vector<KI> ki(n);
        for (size_t i = 0; i < n; ++i)
            ki[i] = KI{morton_key(shift_coord(x[i]), shift_coord(y[i]), shift_coord(z[i])), (uint32_t)i};
        std::sort(ki.begin(), ki.end(), [](const KI &a, const KI &b) {   // == stable_sort on the key
            if (!(a.k == b.k)) return a.k < b.k;
            return a.i < b.i;
        });
        std::vector<int32_t> t(n); std::vector<uint32_t> tc(n);
        for (size_t i = 0; i < n; ++i) t[i] = x[ki[i].i];
        x.swap(t);
        for (size_t i = 0; i < n; ++i) t[i] = y[ki[i].i];
        y.swap(t);
        for (size_t i = 0; i < n; ++i) t[i] = z[ki[i].i];
        z.swap(t);
        for (size_t i = 0; i < n; ++i) tc[i] = c[ki[i].i];
        c.swap(tc);
    }
    out.points = (int64_t)n;
    out.bytes.reserve(n * 4);
    for (size_t b = 0; b * PCR_POINTS_PER_BATCH < n; ++b) {
        size_t o = b * PCR_POINTS_PER_BATCH, before = out.bytes.size();
        BatchStats st;
        if (encode_batch(&x[o], &y[o], &z[o], &c[o], (int32_t)o, las, pad_tails, bc7, out.bytes, st)) return -1;
        out.batch_sizes.push_back((int64_t)(out.bytes.size() - before));
        out.enc_words += st.enc_words; out.sep_words += st.sep_words; out.escaped += st.escaped;
    }
    return 0;
}

int assemble(std::vector<ChunkOut> &chunks, int64_t points_in, void **out_bytes, size_t *out_len,
             pcr_encode_stats *stats)
{
    pcr_file_header h{};
    size_t body = 0;
    int64_t escaped = 0;
    for (auto &c : chunks) {
        h.num_points += c.points; h.num_batches += (int64_t)c.batch_sizes.size();
        h.encoded_bytes += 4 * c.enc_words; h.separate_bytes += 4 * c.sep_words;
        body += c.bytes.size(); escaped += c.escaped;
    }
    h.cluster_bytes = 4 * PCR_CLUSTERS_PER_BATCH * h.num_batches;
    size_t total = sizeof h + 8 * (size_t)h.num_batches + body;
    uint8_t *buf = (uint8_t *)std::malloc(total);
    if (!buf) return fail("out of memory assembling file");
    std::memcpy(buf, &h, sizeof h);                                        // preprocess.cpp:1205-1234
    size_t so = sizeof h, bo = sizeof h + 8 * (size_t)h.num_batches;
    for (auto &c : chunks) {
        std::memcpy(buf + so, c.batch_sizes.data(), 8 * c.batch_sizes.size()); so += 8 * c.batch_sizes.size();
        std::memcpy(buf + bo, c.bytes.data(), c.bytes.size()); bo += c.bytes.size();
        std::vector<uint8_t>().swap(c.bytes);
    }
    *out_bytes = buf; *out_len = total;
    if (stats) {
        stats->num_points_in = points_in; stats->num_points = h.num_points; stats->num_batches = h.num_batches;
        stats->encoded_bytes = h.encoded_bytes; stats->separate_bytes = h.separate_bytes;
        stats->cluster_bytes = h.cluster_bytes; stats->escaped_symbols = escaped;
        stats->total_symbols = h.num_points * 3; stats->file_bytes = (int64_t)total;
    }
    return 0;
}

template <class F>
int parallel_chunks(int64_t nchunks, int nthreads, F &&f)
{
    if (nthreads <= 0) nthreads = (int)std::max(1u, std::thread::hardware_concurrency());
    nthreads = (int)std::min<int64_t>(nthreads, std::max<int64_t>(nchunks, 1));
    std::atomic<int64_t> next{0};
    std::atomic<int> rc{0};
    std::vector<std::string> errs((size_t)nthreads);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            for (;;) {
                int64_t c = next.fetch_add(1);
                if (c >= nchunks || rc.load()) break;
                if (f(c)) { errs[(size_t)t] = g_err; rc.store(-1); }
            }
        });
    for (auto &t : th) t.join();
    if (rc.load()) for (auto &e : errs) if (!e.empty()) { g_err = e; break; }
    return rc.load();
}

// ---------------------------------------------------------------------------------------------
// Synthetic scene
// ---------------------------------------------------------------------------------------------
inline uint64_t mix64(uint64_t v)
{
    v ^= v >> 30; v *= 0xbf58476d1ce4e5b9ull; v ^= v >> 27; v *= 0x94d049bb133111ebull; v ^= v >> 31;
    return v;
}

struct Scene {
    int64_t total; uint64_t seed; int64_t G; double cell;
    Scene(int64_t t, uint64_t s) : total(t), seed(s)
    {
        G = (int64_t)std::ceil(std::sqrt((double)std::max<int64_t>(t, 1)));
        cell = 1.0e6 / (double)G;     // int units per grid cell (tile = [0, 1e6]^2, LAS scale 0.001 -> 1 km)
    }
    // smooth terrain in metres
    static double height_m(double xm, double ym)
    {
        return 40.0 + 25.0 * std::sin(xm * 0.0061) * std::cos(ym * 0.0047)
                    + 9.0 * std::sin(xm * 0.031 + 1.3) * std::sin(ym * 0.027 + 0.4)
                    + 1.5 * std::sin(xm * 0.23 + ym * 0.19);
    }
    void point(int64_t i, int32_t &X, int32_t &Y, int32_t &Z, uint32_t &C) const
    {
        int64_t gx = i % G, gy = i / G;
        uint64_t h = mix64(seed ^ mix64((uint64_t)i + 0x9E3779B97F4A7C15ull));
        double jx = ((h & 0xFFFF) / 65536.0 - 0.5) * 0.08, jy = (((h >> 16) & 0xFFFF) / 65536.0 - 0.5) * 0.08;   // +-4 % of a cell
        double xu = ((double)gx + 0.5 + jx) * cell, yu = ((double)gy + 0.5 + jy) * cell;
        double xm = xu * 0.001, ym = yu * 0.001;
        double zm = height_m(xm, ym) + (((h >> 32) & 0xFF) / 255.0 - 0.5) * 0.004;   // +-2 mm sensor noise
        X = (int32_t)std::min(1.0e6, std::max(0.0, xu));
        Y = (int32_t)std::min(1.0e6, std::max(0.0, yu));
        Z = (int32_t)std::llround(zm * 1000.0);
        // smooth colour field from height and position + a little per-point noise
        double t = (zm - 5.0) / 75.0; t = std::min(1.0, std::max(0.0, t));
        double r = 60 + 150 * t + 20 * std::sin(xm * 0.05);
        double g = 110 + 80 * (1 - t) + 25 * std::cos(ym * 0.04);
        double b = 50 + 90 * t * t + 15 * std::sin((xm + ym) * 0.03);
        int n = (int)((h >> 40) & 7) - 3;
        auto cl = [](double v) { return (uint32_t)std::min(255.0, std::max(0.0, v)); };
        C = cl(r + n) | (cl(g + n) << 8) | (cl(b + n) << 16);
    }
};

// ---------------------------------------------------------------------------------------------
// 4x4 helpers for the camera (column-major like glm)
// ---------------------------------------------------------------------------------------------
struct DM { double m[4][4]; };   // m[col][row]
DM dm_identity() { DM r{}; for (int i = 0; i < 4; ++i) r.m[i][i] = 1; return r; }
DM dm_mul(const DM &a, const DM &b)
{
    DM r{};
    for (int c = 0; c < 4; ++c)
        for (int rr = 0; rr < 4; ++rr)
            r.m[c][rr] = a.m[0][rr] * b.m[c][0] + a.m[1][rr] * b.m[c][1] + a.m[2][rr] * b.m[c][2] + a.m[3][rr] * b.m[c][3];
    return r;
}
DM dm_translate(double x, double y, double z) { DM r = dm_identity(); r.m[3][0] = x; r.m[3][1] = y; r.m[3][2] = z; return r; }
DM dm_rotate(double angle, double ax, double ay, double az)   // glm::rotate(angle, axis)
{
    double c = std::cos(angle), s = std::sin(angle);
    double l = std::sqrt(ax * ax + ay * ay + az * az); ax /= l; ay /= l; az /= l;
    double tx = (1 - c) * ax, ty = (1 - c) * ay, tz = (1 - c) * az;
    DM r = dm_identity();
    r.m[0][0] = c + tx * ax;      r.m[0][1] = tx * ay + s * az; r.m[0][2] = tx * az - s * ay;
    r.m[1][0] = ty * ax - s * az; r.m[1][1] = c + ty * ay;      r.m[1][2] = ty * az + s * ax;
    r.m[2][0] = tz * ax + s * ay; r.m[2][1] = tz * ay - s * ax; r.m[2][2] = c + tz * az;
    return r;
}
bool dm_inverse(const DM &a, DM &out)
{
    double inv[16], m[16];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) m[c * 4 + r] = a.m[c][r];
    inv[0] = m[5]*m[10]*m[15] - m[5]*m[11]*m[14] - m[9]*m[6]*m[15] + m[9]*m[7]*m[14] + m[13]*m[6]*m[11] - m[13]*m[7]*m[10];
    inv[4] = -m[4]*m[10]*m[15] + m[4]*m[11]*m[14] + m[8]*m[6]*m[15] - m[8]*m[7]*m[14] - m[12]*m[6]*m[11] + m[12]*m[7]*m[10];
    inv[8] = m[4]*m[9]*m[15] - m[4]*m[11]*m[13] - m[8]*m[5]*m[15] + m[8]*m[7]*m[13] + m[12]*m[5]*m[11] - m[12]*m[7]*m[9];
    inv[12] = -m[4]*m[9]*m[14] + m[4]*m[10]*m[13] + m[8]*m[5]*m[14] - m[8]*m[6]*m[13] - m[12]*m[5]*m[10] + m[12]*m[6]*m[9];
    inv[1] = -m[1]*m[10]*m[15] + m[1]*m[11]*m[14] + m[9]*m[2]*m[15] - m[9]*m[3]*m[14] - m[13]*m[2]*m[11] + m[13]*m[3]*m[10];
    inv[5] = m[0]*m[10]*m[15] - m[0]*m[11]*m[14] - m[8]*m[2]*m[15] + m[8]*m[3]*m[14] + m[12]*m[2]*m[11] - m[12]*m[3]*m[10];
    inv[9] = -m[0]*m[9]*m[15] + m[0]*m[11]*m[13] + m[8]*m[1]*m[15] - m[8]*m[3]*m[13] - m[12]*m[1]*m[11] + m[12]*m[3]*m[9];
    inv[13] = m[0]*m[9]*m[14] - m[0]*m[10]*m[13] - m[8]*m[1]*m[14] + m[8]*m[2]*m[13] + m[12]*m[1]*m[10] - m[12]*m[2]*m[9];
    inv[2] = m[1]*m[6]*m[15] - m[1]*m[7]*m[14] - m[5]*m[2]*m[15] + m[5]*m[3]*m[14] + m[13]*m[2]*m[7] - m[13]*m[3]*m[6];
    inv[6] = -m[0]*m[6]*m[15] + m[0]*m[7]*m[14] + m[4]*m[2]*m[15] - m[4]*m[3]*m[14] - m[12]*m[2]*m[7] + m[12]*m[3]*m[6];
    inv[10] = m[0]*m[5]*m[15] - m[0]*m[7]*m[13] - m[4]*m[1]*m[15] + m[4]*m[3]*m[13] + m[12]*m[1]*m[7] - m[12]*m[3]*m[5];
    inv[14] = -m[0]*m[5]*m[14] + m[0]*m[6]*m[13] + m[4]*m[1]*m[14] - m[4]*m[2]*m[13] - m[12]*m[1]*m[6] + m[12]*m[2]*m[5];
    inv[3] = -m[1]*m[6]*m[11] + m[1]*m[7]*m[10] + m[5]*m[2]*m[11] - m[5]*m[3]*m[10] - m[9]*m[2]*m[7] + m[9]*m[3]*m[6];
    inv[7] = m[0]*m[6]*m[11] - m[0]*m[7]*m[10] - m[4]*m[2]*m[11] + m[4]*m[3]*m[10] + m[8]*m[2]*m[7] - m[8]*m[3]*m[6];
    inv[11] = -m[0]*m[5]*m[11] + m[0]*m[7]*m[9] + m[4]*m[1]*m[11] - m[4]*m[3]*m[9] - m[8]*m[1]*m[7] + m[8]*m[3]*m[5];
    inv[15] = m[0]*m[5]*m[10] - m[0]*m[6]*m[9] - m[4]*m[1]*m[10] + m[4]*m[2]*m[9] + m[8]*m[1]*m[6] - m[8]*m[2]*m[5];
    double det = m[0]*inv[0] + m[1]*inv[4] + m[2]*inv[8] + m[3]*inv[12];
    if (det == 0) return false;
    det = 1.0 / det;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) out.m[c][r] = inv[c * 4 + r] * det;
    return true;
}
struct FM { float m[4][4]; };
FM fm_from(const DM &d) { FM r; for (int c = 0; c < 4; ++c) for (int rr = 0; rr < 4; ++rr) r.m[c][rr] = (float)d.m[c][rr]; return r; }
FM fm_mul(const FM &a, const FM &b)   // glm::mat4 operator*: ((A0*b0 + A1*b1) + A2*b2) + A3*b3 per component
{
    FM r;
    for (int c = 0; c < 4; ++c)
        for (int rr = 0; rr < 4; ++rr) {
            volatile float t0 = a.m[0][rr] * b.m[c][0], t1 = a.m[1][rr] * b.m[c][1];
            volatile float t2 = a.m[2][rr] * b.m[c][2], t3 = a.m[3][rr] * b.m[c][3];
            volatile float s = t0 + t1; s = s + t2; s = s + t3;
            r.m[c][rr] = s;
        }
    return r;
}
void fm_rows(const FM &a, float *rows) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) rows[r * 4 + c] = a.m[c][r]; }

} // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char *pcr_host_last_error(void) { return g_err.c_str(); }
void pcr_host_free(void *p) { std::free(p); }

void pcr_morton_key(uint32_t x, uint32_t y, uint32_t z, uint32_t *hi, uint64_t *lo)
{
    MortonKey k = morton_key(x, y, z);
    *hi = k.hi; *lo = k.lo;
}

void pcr_bc1_encode_block(const uint32_t *colors16, uint8_t *out8) { bc1_encode(colors16, out8); }
void pcr_bc7_encode_block(const uint32_t *colors16, uint8_t *out16) { pcr_codec::bc7_mode6_encode(colors16, out16); }

int pcr_encode_points(const int32_t *x, const int32_t *y, const int32_t *z, const uint32_t *color,
                      int64_t n, const pcr_las_info *las, int flags, int64_t chunk_points,
                      int nthreads, void **out_bytes, size_t *out_len, pcr_encode_stats *stats)
{
    if (!x || !y || !z || !color || !las || !out_bytes || !out_len) return fail("null argument");
    if (n <= 0) return fail("no points");
    if (chunk_points <= 0) chunk_points = PCR_DEFAULT_CHUNK_POINTS;
    int64_t nchunks = (n + chunk_points - 1) / chunk_points;
    std::vector<ChunkOut> chunks((size_t)nchunks);
    int rc = parallel_chunks(nchunks, nthreads, [&](int64_t c) {
        int64_t a = c * chunk_points, b = std::min(n, a + chunk_points);
        std::vector<int32_t> cx(x + a, x + b), cy(y + a, y + b), cz(z + a, z + b);
        std::vector<uint32_t> cc(color + a, color + b);
        return encode_chunk(cx, cy, cz, cc, *las, flags, chunks[(size_t)c]);
    });
    if (rc) return rc;
    return assemble(chunks, n, out_bytes, out_len, stats);
}

int pcr_synth_las_info(int64_t total_points, uint64_t seed, pcr_las_info *las)
{
    (void)total_points; (void)seed;
    if (!las) return fail("null argument");
    for (int k = 0; k < 3; ++k) { las->scale[k] = 0.001; las->offset[k] = 0.0; las->min[k] = 0.0; }
    las->max[0] = 1000.0; las->max[1] = 1000.0; las->max[2] = 100.0;
    return 0;
}

int pcr_synth_points(int64_t total_points, uint64_t seed, int64_t first, int64_t count,
                     int32_t *x, int32_t *y, int32_t *z, uint32_t *color)
{
    if (total_points <= 0 || first < 0 || count < 0 || first + count > total_points) return fail("bad range");
    Scene sc(total_points, seed);
    for (int64_t i = 0; i < count; ++i) {
        int32_t X, Y, Z; uint32_t C;
        sc.point(first + i, X, Y, Z, C);
        if (x) x[i] = X;
        if (y) y[i] = Y;
        if (z) z[i] = Z;
        if (color) color[i] = C;
    }
    return 0;
}

int pcr_synth_encode(int64_t total_points, uint64_t seed, int64_t first, int64_t count,
                     int64_t chunk_points, int nthreads, void **out_bytes, size_t *out_len,
                     pcr_encode_stats *stats)
{
    if (!out_bytes || !out_len) return fail("null argument");
    if (chunk_points <= 0) chunk_points = PCR_DEFAULT_CHUNK_POINTS;
    if (total_points <= 0 || first < 0 || count <= 0 || first + count > total_points) return fail("bad range");
    if (first % chunk_points) return fail("first must be a multiple of chunk_points");
    pcr_las_info las;
    pcr_synth_las_info(total_points, seed, &las);
    Scene sc(total_points, seed);
    int64_t nchunks = (count + chunk_points - 1) / chunk_points;
    std::vector<ChunkOut> chunks((size_t)nchunks);
    int rc = parallel_chunks(nchunks, nthreads, [&](int64_t c) {
        int64_t a = first + c * chunk_points, b = std::min(first + count, a + chunk_points);
        size_t m = (size_t)(b - a);
        std::vector<int32_t> cx(m), cy(m), cz(m); std::vector<uint32_t> cc(m);
        for (size_t i = 0; i < m; ++i) sc.point(a + (int64_t)i, cx[i], cy[i], cz[i], cc[i]);
        return encode_chunk(cx, cy, cz, cc, las, PCR_ENCODE_MORTON_SORT, chunks[(size_t)c]);
    });
    if (rc) return rc;
    return assemble(chunks, count, out_bytes, out_len, stats);
}

int pcr_huffman_build(const int32_t *symbols, int64_t n, int32_t *dt_values, int32_t *dt_cwlen,
                      const int32_t *query, int64_t num_query, uint32_t *out_cw, int32_t *out_len)
{
    if (!symbols || n <= 0 || !dt_values || !dt_cwlen) return fail("bad argument");
    std::vector<int32_t> sorted(symbols, symbols + n);
    std::sort(sorted.begin(), sorted.end());
    std::vector<int32_t> syms; std::vector<uint32_t> freqs;
    for (size_t i = 0; i < sorted.size();) {
        size_t j = i;
        while (j < sorted.size() && sorted[j] == sorted[i]) ++j;
        syms.push_back(sorted[i]); freqs.push_back((uint32_t)(j - i));
        i = j;
    }
    HuffmanCode hc;
    build_code(syms, freqs, hc);
    if (table_from_codes(hc.symbols.data(), hc.codes.data(), hc.symbols.size(), dt_values, dt_cwlen))
        return fail("decoder table incomplete");
    for (int64_t q = 0; q < num_query; ++q) {
        auto it = std::lower_bound(hc.symbols.begin(), hc.symbols.end(), query[q]);
        if (it == hc.symbols.end() || *it != query[q]) return fail("query symbol not in alphabet");
        const Code &c = hc.codes[(size_t)(it - hc.symbols.begin())];
        out_cw[q] = c.cw; out_len[q] = c.len;
    }
    return 0;
}

int pcr_pack_chain(const int32_t *symbols, int n, const int32_t *dict_symbols, const uint32_t *dict_cw,
                   const int32_t *dict_len, int64_t dict_n, uint32_t **words, int32_t *num_words,
                   int32_t **separate, int32_t *num_separate, int32_t **num_cw)
{
    std::unordered_map<int32_t, Code> d;
    for (int64_t i = 0; i < dict_n; ++i) d[dict_symbols[i]] = Code{dict_cw[i], dict_len[i]};
    for (int i = 0; i < n; ++i) if (!d.count(symbols[i])) return fail("symbol not in dictionary");
    PackedChain pc;
    pack_chain(symbols, n, [&](int32_t s) -> const Code & { return d.at(s); }, pc);
    auto dup = [](const void *p, size_t bytes) { void *q = std::malloc(bytes ? bytes : 1); if (q && bytes) std::memcpy(q, p, bytes); return q; };
    *words = (uint32_t *)dup(pc.words.data(), pc.words.size() * 4); *num_words = (int32_t)pc.words.size();
    *separate = (int32_t *)dup(pc.separate.data(), pc.separate.size() * 4); *num_separate = (int32_t)pc.separate.size();
    *num_cw = (int32_t *)dup(pc.num_cw.data(), pc.num_cw.size() * 4);
    return 0;
}

int pcr_table_from_dict(const int32_t *dict_symbols, const uint32_t *dict_cw, const int32_t *dict_len,
                        int64_t dict_n, int32_t *dt_values, int32_t *dt_cwlen)
{
    std::vector<Code> codes((size_t)dict_n);
    for (int64_t i = 0; i < dict_n; ++i) codes[(size_t)i] = Code{dict_cw[i], dict_len[i]};
    int rc = table_from_codes(dict_symbols, codes.data(), (size_t)dict_n, dt_values, dt_cwlen);
    if (rc == -2) return fail("table has untouched entries");
    if (rc) return fail("invalid code in dictionary");
    return 0;
}

int pcr_las_quantize(const int32_t *x, const int32_t *y, const int32_t *z, const uint32_t *color, int64_t n,
                     const pcr_las_info *las, pcr_xyz_batch *batches, uint32_t *xyz12, uint32_t *xyz8,
                     uint32_t *xyz4, uint32_t *rgba, int nthreads)
{
    if (!x || !y || !z || !color || !las || !batches || !xyz12 || !xyz8 || !xyz4 || !rgba || n <= 0) return fail("bad argument");
    const int64_t PPB = PCR_POINTS_PER_BATCH, WG = PCR_WORKGROUP_SIZE;
    const int64_t nb = (n + PPB - 1) / PPB;
    const float fscale[3] = {(float)las->scale[0], (float)las->scale[1], (float)las->scale[2]};   // uScale is a vec3
    const float fmin[3] = {(float)las->min[0], (float)las->min[1], (float)las->min[2]};           // uBoxMin is a vec3
    return parallel_chunks(nb, nthreads, [&](int64_t b) {
        const int64_t first = b * PPB, cnt = std::min(PPB, n - first);
        std::vector<float> px((size_t)cnt), py((size_t)cnt), pz((size_t)cnt);
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int64_t k = 0; k < cnt; ++k) {
            // getPoint, computeLasLoader.cs:178-180: float(double(X) * uScale + uOffset - double(uBoxMin))
            volatile double ax = (double)x[first + k] * (double)fscale[0], ay = (double)y[first + k] * (double)fscale[1],
                            az = (double)z[first + k] * (double)fscale[2];
            const float fx = (float)(ax + las->offset[0] - (double)fmin[0]);
            const float fy = (float)(ay + las->offset[1] - (double)fmin[1]);
            const float fz = (float)(az + las->offset[2] - (double)fmin[2]);
            px[(size_t)k] = fx; py[(size_t)k] = fy; pz[(size_t)k] = fz;
            mn[0] = std::min(mn[0], fx); mn[1] = std::min(mn[1], fy); mn[2] = std::min(mn[2], fz);
            mx[0] = std::max(mx[0], fx); mx[1] = std::max(mx[1], fy); mx[2] = std::max(mx[2], fz);
        }
        pcr_xyz_batch g{};
        g.min_x = mn[0]; g.min_y = mn[1]; g.min_z = mn[2]; g.max_x = mx[0]; g.max_y = mx[1]; g.max_z = mx[2];
        g.num_points = (int32_t)cnt;                                           // processPoints :271-273
        batches[b] = g;
        const float size[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
        auto q30 = [](float p, float lo, float sz) -> uint32_t {                // :288-294
            if (!(sz > 0.0f)) return 0u;                                        // flat box: the shader divides 0 by 0
            volatile float t = (p - lo) / sz;
            volatile float u = t * 1073741824.0f;
            uint32_t v = u >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)u;
            return std::min(v, (uint32_t)(1073741824u - 1u));
        };
        for (int64_t k = 0; k < PPB; ++k) {
            // slot k of the batch in (iteration, lane) order holds input point first + k
            const int64_t slot = first + k;
            if (k >= cnt) { xyz12[slot] = xyz8[slot] = xyz4[slot] = rgba[slot] = 0; continue; }
            const uint32_t X = q30(px[(size_t)k], mn[0], size[0]), Y = q30(py[(size_t)k], mn[1], size[1]), Z = q30(pz[(size_t)k], mn[2], size[2]);
            xyz4[slot] = ((X >> 20) & 1023u) | (((Y >> 20) & 1023u) << 10) | (((Z >> 20) & 1023u) << 20);   // :313-322
            xyz8[slot] = ((X >> 10) & 1023u) | (((Y >> 10) & 1023u) << 10) | (((Z >> 10) & 1023u) << 20);   // :325-334
            xyz12[slot] = (X & 1023u) | ((Y & 1023u) << 10) | ((Z & 1023u) << 20);                           // :337-346
            rgba[slot] = color[first + k];                                      // :284
        }
        (void)WG;
        return 0;
    });
}

int pcr_camera_orbit(double yaw, double pitch, double radius, const double target[3],
                     int width, int height, double fovy_deg, double near_plane, double far_plane,
                     pcr_render_params *out)
{
    if (!out || !target || width <= 0 || height <= 0) return fail("bad argument");
    // OrbitControls::update, include/OrbitControls.h:116-134
    DM flip{};
    flip.m[0][0] = 1; flip.m[1][2] = 1; flip.m[2][1] = -1; flip.m[3][3] = 1;
    DM world = dm_mul(dm_mul(dm_mul(dm_mul(dm_translate(target[0], target[1], target[2]), dm_rotate(yaw, 0, 0, 1)),
                                    dm_rotate(pitch, 1, 0, 0)), flip), dm_translate(0, 0, radius));
    DM view;
    if (!dm_inverse(world, view)) return fail("singular camera matrix");          // Camera.h:33
    // glm::perspective (RH, depth -1..1), Camera.h:35-36
    double fovy = 3.14159265358979323846 * fovy_deg / 180.0;
    double aspect = (double)width / (double)height, th = std::tan(fovy / 2.0);
    DM proj{};
    proj.m[0][0] = 1.0 / (aspect * th);
    proj.m[1][1] = 1.0 / th;
    proj.m[2][2] = -(far_plane + near_plane) / (far_plane - near_plane);
    proj.m[2][3] = -1.0;
    proj.m[3][2] = -(2.0 * far_plane * near_plane) / (far_plane - near_plane);
    // huffman_hqs.h:157-169: world = identity; float view/proj; worldViewProj = proj * view * world
    FM fview = fm_from(view), fproj = fm_from(proj);
    FM ident = fm_from(dm_identity());
    FM world_view = fm_mul(fview, ident);
    FM wvp = fm_mul(fm_mul(fproj, fview), ident);
    std::memset(out, 0, sizeof *out);
    fm_rows(wvp, out->transform);
    fm_rows(world_view, out->world_view);
    fm_rows(fproj, out->proj);
    out->width = width; out->height = height;
    out->points_per_thread = PCR_POINTS_PER_THREAD;
    out->lod_percent = 10;             // (int)(Debug::LOD * 100), include/Debug.h:21
    out->enable_frustum_culling = 1;   // include/Debug.h:23
    return 0;
}

} // extern "C"
