// ref_harness.cpp — ORACLE-SIDE DRIVER (TEST INFRASTRUCTURE) that exposes the reference's own,
// UNMODIFIED sources through a C ABI so tests can pin the build's restatements against them.
//
// Compiled by oracle/Makefile ONLY when /root/reference is present (this container), from the
// sources where they lie; output goes to oracle/_ref/libpcr_ref.so (git-ignored, travels with gpurun).
// Nothing of the reference is copied into this repository: this file only #includes
//     /root/reference/include/huffman.h      (Huffman<T>: tree, dictionary, table, packer, chain decoder)
//     /root/reference/src/mymorton.h         (96-bit Morton key + stable order)
//     /root/reference/include/rgbcx.h        (+ src/rgbcx.cpp linked: BC1 encoder/unpacker)
//     /root/reference/include/bc7enc.h, bc7decomp.h (+ src/bc7enc.cpp, src/bc7decomp.cpp linked: BC7 encoder/unpacker)
// and calls their functions. The reference's `preprocess`/BatchDumpData/loader cannot be built here
// without stand-ins for GL/CUDA headers (compute/Resources.h -> Renderer.h, CudaProgram.h), so they are
// treated as unbuildable and are restated instead.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "huffman.h"     // reference: include/huffman.h
#include "mymorton.h"    // reference: src/mymorton.h
#include "rgbcx.h"       // reference: include/rgbcx.h
#include "bc7enc.h"      // reference: include/bc7enc.h
#include "bc7decomp.h"   // reference: include/bc7decomp.h

namespace {
struct RefCode {
    Huffman<int32_t> h;
    std::unordered_map<int32_t, std::pair<uint32_t, int>> dict;
    std::vector<std::pair<int32_t, int>> table;
};
template <class T> T *dup(const std::vector<T> &v)
{
    T *p = (T *)std::malloc(v.size() * sizeof(T) + 1);
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}
bool g_rgbcx_init = false;
bool g_bc7_init = false;
}

extern "C" {

// Batch::calculate, src/preprocess.cpp:765-770, on the reference's own Huffman<int32_t>.
void *ref_code_build(const int32_t *symbols, int64_t n)
{
    auto *rc = new RefCode();
    std::vector<int32_t> data(symbols, symbols + n);
    rc->h.calculate_frequencies(data);
    rc->h.generate_huffman_tree_priority_queue();
    rc->dict = rc->h.create_dictionary_pjn<uint32_t>(4096);
    rc->table = rc->h.get_gpu_huffman_table_pjn<uint32_t>(rc->dict, 4096);
    rc->h.clear_huffman_tree();
    return rc;
}
// test_huffman.cpp:35-40 uses the sort-based tree instead
void *ref_code_build_sorted(const int32_t *symbols, int64_t n)
{
    auto *rc = new RefCode();
    std::vector<int32_t> data(symbols, symbols + n);
    rc->h.calculate_frequencies(data);
    rc->h.generate_huffman_tree();
    rc->dict = rc->h.create_dictionary_pjn<uint32_t>(4096);
    rc->table = rc->h.get_gpu_huffman_table_pjn<uint32_t>(rc->dict, 4096);
    rc->h.clear_huffman_tree();
    return rc;
}
void ref_code_free(void *p) { delete (RefCode *)p; }
int64_t ref_code_dict_size(void *p) { return (int64_t)((RefCode *)p)->dict.size(); }
void ref_code_dict(void *p, int32_t *symbols, uint32_t *cw, int32_t *len)
{
    int64_t i = 0;
    for (auto &kv : ((RefCode *)p)->dict) { symbols[i] = kv.first; cw[i] = kv.second.first; len[i] = kv.second.second; ++i; }
}
void ref_code_table(void *p, int32_t *values, int32_t *lens)
{
    auto &t = ((RefCode *)p)->table;
    for (size_t i = 0; i < t.size(); ++i) { values[i] = t[i].first; lens[i] = t[i].second; }
}
// include/huffman.h:242-300
void ref_code_pack(void *p, const int32_t *symbols, int n, uint32_t **words, int32_t *num_words,
                   int32_t **separate, int32_t *num_separate, int32_t **num_cw)
{
    RefCode *rc = (RefCode *)p;
    std::vector<int32_t> data(symbols, symbols + n);
    auto ret = rc->h.compress_udtype_subarray_fast_pjn_idea<uint32_t, std::vector<int32_t>::iterator, int32_t>(
        data.begin(), data.end(), rc->dict, 4096);
    *words = dup(std::get<0>(ret)); *num_words = (int32_t)std::get<0>(ret).size();
    *separate = dup(std::get<1>(ret)); *num_separate = (int32_t)std::get<1>(ret).size();
    *num_cw = dup(std::get<2>(ret));
}
// include/huffman.h:433-477
void ref_code_unpack(void *p, const uint32_t *words, int num_words, const int32_t *separate, int num_separate,
                     int n, int32_t *out)
{
    RefCode *rc = (RefCode *)p;
    std::vector<uint32_t> bs(words, words + num_words);
    std::vector<int32_t> sep(separate, separate + num_separate);
    std::vector<int32_t> dec((size_t)n);
    rc->h.decompress_udtype_subarray_fast_pjn_idea<uint32_t, std::vector<int32_t>::iterator, int32_t>(
        dec.begin(), dec.end(), bs, sep, rc->table);
    std::memcpy(out, dec.data(), (size_t)n * 4);
}
void ref_free(void *p) { std::free(p); }

// src/mymorton.h:12-37 and :39-58
void ref_morton_key(uint32_t x, uint32_t y, uint32_t z, uint32_t *hi, uint64_t *lo)
{
    auto c = mymorton::get_morton_code_3D(x, y, z);
    *hi = c.first; *lo = c.second;
}
void ref_morton_order(const int32_t *x, const int32_t *y, const int32_t *z, int64_t n, uint32_t *order)
{
    std::vector<int32_t> X(x, x + n), Y(y, y + n), Z(z, z + n);
    auto o = mymorton::get_morton_order(X, Y, Z);
    std::memcpy(order, o.data(), (size_t)n * 4);
}

// src/rgbcx.cpp: encoder exactly as Chain::encode_color_bc1 calls it (src/preprocess.cpp:282-297, 1171)
void ref_bc1_encode(const uint32_t *colors16, uint8_t *out8)
{
    if (!g_rgbcx_init) { rgbcx::init(rgbcx::bc1_approx_mode::cBC1Ideal); g_rgbcx_init = true; }
    uint32_t block[16];
    for (int i = 0; i < 16; ++i) block[i] = colors16[i] | 0xFF000000u;
    rgbcx::encode_bc1(8, out8, (const uint8_t *)block, false, false);
}
void ref_bc1_unpack(const uint8_t *block8, uint32_t *colors16)
{
    if (!g_rgbcx_init) { rgbcx::init(rgbcx::bc1_approx_mode::cBC1Ideal); g_rgbcx_init = true; }
    rgbcx::unpack_bc1(block8, colors16, true, rgbcx::bc1_approx_mode::cBC1Ideal);
}

// src/bc7enc.cpp: encoder exactly as Chain::encode_color_bc7 calls it (src/preprocess.cpp:299-316: default parameters with
// m_mode_mask = 1 << 6; bc7enc_compress_block_init() once, :1172)
void ref_bc7_encode(const uint32_t *colors16, uint8_t *out16)
{
    if (!g_bc7_init) { bc7enc_compress_block_init(); g_bc7_init = true; }
    uint32_t block[16];
    std::memcpy(block, colors16, sizeof block);
    bc7enc_compress_block_params params;
    bc7enc_compress_block_params_init(&params);
    params.m_mode_mask = 1 << 6;
    bc7enc_compress_block(out16, (const uint8_t *)block, &params);
}
// src/bc7decomp.cpp: the specification's decoder (what src/preprocess.cpp:893-910 checks its blocks with)
int ref_bc7_unpack(const uint8_t *block16, uint32_t *colors16)
{
    return bc7decomp::unpack_bc7(block16, (bc7decomp::color_rgba *)colors16) ? 1 : 0;
}

} // extern "C"
