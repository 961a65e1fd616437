/*
 * pcr_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE). See pcr_oracle.h.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -mfma -fPIC -shared   (oracle/Makefile)
 *   -ffp-contract=off : every fused multiply-add below is written out with fmaf()/fma(); nothing else fuses.
 *   -mfma             : fmaf()/fma() compile to the hardware instruction (same bits as glibc's soft path).
 *
 * Numeric contract = SURVEY.md Appendix C (IEEE-strict restatement; the reference JIT-compiles with
 * --use_fast_math, include/CudaProgram.h:35-39, whose bits are not reproducible off NVIDIA hardware).
 * All citations are relative to the reference checkout.
 */
#include "pcr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * small vector helpers (modules/huffman_mem_iter_cuda/helper_math.h: plain per-component ops)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z, w; } f4;

/* helper_math.h:1266-1269 dot(float4,float4) with left-to-right FMA contraction (Appendix C.1). */
static inline float dot4(const float *r, f4 v)
{
    return fmaf(r[3], v.w, fmaf(r[2], v.z, fmaf(r[1], v.y, r[0] * v.x)));
}

/* render.cu:208-211 matMul */
static inline f4 mat_mul(const float *m, f4 v)
{
    f4 o = { dot4(m + 0, v), dot4(m + 4, v), dot4(m + 8, v), dot4(m + 12, v) };
    return o;
}

static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------------------------------------
 * frustum cull: render.cu:218-274 (same text in huffman_hqs/depth.cu:69-125, huffman_hqs/render.cu:70-126)
 * ---------------------------------------------------------------------------------------------- */
static inline float t_elem(const pcr_render_params *p, int index)
{
    int a = index % 4, b = index / 4;           /* render.cu:228-233: rows[a] component b */
    return p->transform[a * 4 + b];
}

static int plane_accepts(float x, float y, float z, float w, const float bmin[3], const float bmax[3])
{
    /* createPlane, render.cu:239-246 */
    float nl = sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
    float nx = x / nl, ny = y / nl, nz = z / nl, c = w / nl;
    /* p-vertex, render.cu:261-264 */
    float vx = nx > 0.0f ? bmax[0] : bmin[0];
    float vy = ny > 0.0f ? bmax[1] : bmin[1];
    float vz = nz > 0.0f ? bmax[2] : bmin[2];
    /* distanceToPoint, render.cu:235-237 */
    float d = fmaf(nz, vz, fmaf(ny, vy, nx * vx)) + c;
    return !(d < 0.0f);
}

static int intersects_frustum(const pcr_render_params *p, const float bmin[3], const float bmax[3])
{
#define T(i) t_elem(p, (i))
    /* plane order/signs: render.cu:249-256 */
    if (!plane_accepts(T(3) - T(0), T(7) - T(4), T(11) - T(8),  T(15) - T(12), bmin, bmax)) return 0;
    if (!plane_accepts(T(3) + T(0), T(7) + T(4), T(11) + T(8),  T(15) + T(12), bmin, bmax)) return 0;
    if (!plane_accepts(T(3) + T(1), T(7) + T(5), T(11) + T(9),  T(15) + T(13), bmin, bmax)) return 0;
    if (!plane_accepts(T(3) - T(1), T(7) - T(5), T(11) - T(9),  T(15) - T(13), bmin, bmax)) return 0;
    if (!plane_accepts(T(3) - T(2), T(7) - T(6), T(11) - T(10), T(15) - T(14), bmin, bmax)) return 0;
    if (!plane_accepts(T(3) + T(2), T(7) + T(6), T(11) + T(10), T(15) + T(14), bmin, bmax)) return 0;
#undef T
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * cull + LOD: render.cu:333-379 (mem_iter), huffman_hqs/depth.cu:186-227, huffman_hqs/render.cu:351-392
 * ---------------------------------------------------------------------------------------------- */
int pcr_oracle_batch_lod(const pcr_gpu_batch *b, const pcr_render_params *p, int variant,
                         int *num_points_to_render, int *use_double)
{
    /* render.cu:336 — las_min narrowed to float first */
    float lm[3] = { (float)b->las_min_x, (float)b->las_min_y, (float)b->las_min_z };
    float bmin[3] = { b->min_x - lm[0], b->min_y - lm[1], b->min_z - lm[2] };   /* :340 */
    float bmax[3] = { b->max_x - lm[0], b->max_y - lm[1], b->max_z - lm[2] };   /* :341 */

    if (p->enable_frustum_culling && !intersects_frustum(p, bmin, bmax)) return 0; /* :342-344 */

    /* :349-367 */
    f4 ctr = { 0.5f * (bmin[0] + bmax[0]), 0.5f * (bmin[1] + bmax[1]), 0.5f * (bmin[2] + bmax[2]), 1.0f };
    float dx = bmin[0] - bmax[0], dy = bmin[1] - bmax[1], dz = bmin[2] - bmax[2];
    float rad = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    f4 vc = mat_mul(p->world_view, ctr);
    f4 ve = { vc.x + rad, vc.y + 0.0f, vc.z + 0.0f, vc.w + 0.0f };
    f4 pc = mat_mul(p->proj, vc);
    f4 pe = mat_mul(p->proj, ve);
    float fw = (float)p->width, fh = (float)p->height;
    float scx = fw * (0.5f * (pc.x / pc.w + 1.0f));
    float scy = fh * (0.5f * (pc.y / pc.w + 1.0f));
    float sex = fw * (0.5f * (pe.x / pe.w + 1.0f));
    float sey = fh * (0.5f * (pe.y / pe.w + 1.0f));
    float ddx = sex - scx, ddy = sey - scy;
    float px = sqrtf(fmaf(ddy, ddy, ddx * ddx));

    *use_double = px >= 100.0f;                                         /* :370 */
    if (variant == PCR_ORACLE_MEM_ITER) px = px / 100.0f;               /* mem_iter render.cu:372 */
    else                                px = (float)((double)px / 100.0); /* hqs depth.cu:223, render.cu:388 */
    float pct = (float)((double)(1.8f * px) - 0.3);                     /* :373 */
    pct = fmaxf((float)p->lod_percent / 100.0f, fminf(pct, 1.0f));      /* :374, helper_math clamp */
    int npr = (int)(pct * (float)p->points_per_thread);                 /* :375 */
    if (npr > p->points_per_thread) npr = p->points_per_thread;
    *num_points_to_render = npr;
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * BC1 colour: render.cu:23-65 (always 4-colour mode)
 * ---------------------------------------------------------------------------------------------- */
uint32_t pcr_oracle_decode_bc1(uint64_t point_index, const uint8_t *colors)
{
    uint64_t block = point_index / 16, local = point_index % 16;
    const uint8_t *bp = colors + block * 8;
    uint32_t l = bp[0] | ((uint32_t)bp[1] << 8);
    int cr0 = (l >> 11) & 31, cg0 = (l >> 5) & 63, cb0 = l & 31;
    int r0 = (cr0 << 3) | (cr0 >> 2), g0 = (cg0 << 2) | (cg0 >> 4), b0 = (cb0 << 3) | (cb0 >> 2);
    uint32_t h = bp[2] | ((uint32_t)bp[3] << 8);
    int cr1 = (h >> 11) & 31, cg1 = (h >> 5) & 63, cb1 = h & 31;
    int r1 = (cr1 << 3) | (cr1 >> 2), g1 = (cg1 << 2) | (cg1 >> 4), b1 = (cb1 << 3) | (cb1 >> 2);
    int word = (bp[4 + local / 4] >> (2 * (local % 4))) & 3;
    int r, g, b;
    switch (word) {
    case 0:  r = r0; g = g0; b = b0; break;
    case 1:  r = r1; g = g1; b = b1; break;
    case 2:  r = (r0 * 2 + r1) / 3; g = (g0 * 2 + g1) / 3; b = (b0 * 2 + b1) / 3; break;
    default: r = (r0 + r1 * 2) / 3; g = (g0 + g1 * 2) / 3; b = (b0 + b1 * 2) / 3; break;
    }
    return (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
}

/* ------------------------------------------------------------------------------------------------
 * BC7 mode-6 colour as the reference's kernels decode it (COLOR_COMPRESSION == 7): huffman_hqs/render.cu:240-273 (the same
 * function in huffman_mem_iter_cuda/render.cu:121-154 and resolve.cu:115). struct bc7_mode_6 (render.cu:66-110): low
 * quadword = mode:7 r0:7 r1:7 g0:7 g1:7 b0:7 b1:7 a0:7 a1:7 p0:1, high quadword = p1:1 then the sixteen indices, 3 bits for
 * pixel 0 and 4 for the others. Endpoint = (7 bits << 1) | p-bit. The kernel takes `(hi >> (4 * local)) & 15` as the index of
 * EVERY pixel, so pixels 1..15 get their BC7 index and pixel 0 gets (index << 1) | p1 -- reproduced, it is what the
 * reference draws (`if (idx == 0) idx = idx >> 1` changes nothing). Weight = round(idx * 64 / 15) computed in float
 * (linspace_idx), which is BC7's 4-bit weight table; channel = (e0 * (64 - w) + e1 * w + 32) >> 6, alpha in bits 24..31.
 * ---------------------------------------------------------------------------------------------- */
uint32_t pcr_oracle_decode_bc7(uint64_t point_index, const uint8_t *colors)
{
    uint64_t block = point_index / 16, local = point_index % 16;
    uint64_t lo, hi;
    memcpy(&lo, colors + block * 16, 8);
    memcpy(&hi, colors + block * 16 + 8, 8);
    uint32_t p0 = (uint32_t)(lo >> 63) & 1u, p1 = (uint32_t)hi & 1u;
    uint32_t r0 = (((uint32_t)(lo >> 7) & 127u) << 1) | p0, r1 = (((uint32_t)(lo >> 14) & 127u) << 1) | p1;
    uint32_t g0 = (((uint32_t)(lo >> 21) & 127u) << 1) | p0, g1 = (((uint32_t)(lo >> 28) & 127u) << 1) | p1;
    uint32_t b0 = (((uint32_t)(lo >> 35) & 127u) << 1) | p0, b1 = (((uint32_t)(lo >> 42) & 127u) << 1) | p1;
    uint32_t a0 = (((uint32_t)(lo >> 49) & 127u) << 1) | p0, a1 = (((uint32_t)(lo >> 56) & 127u) << 1) | p1;
    int idx = (int)((hi >> (local * 4)) & 0xF);
    float step = (64.0f - 0.0f) / (float)(16 - 1);          /* linspace_idx(0, 64, 16, idx), render.cu:113-118 */
    float val = 0.0f + (float)idx * step;
    uint32_t w = (uint32_t)(int)roundf(val), iw = 64u - w;
    return ((uint32_t)(uint8_t)((r0 * iw + r1 * w + 32) >> 6)) |
           ((uint32_t)(uint8_t)((g0 * iw + g1 * w + 32) >> 6) << 8) |
           ((uint32_t)(uint8_t)((b0 * iw + b1 * w + 32) >> 6) << 16) |
           ((uint32_t)(uint8_t)((a0 * iw + a1 * w + 32) >> 6) << 24);
}

/* ------------------------------------------------------------------------------------------------
 * Lane-accurate decode of one 32-lane cluster: render.cu:404-451. The GPU runs the 32 lanes in
 * lockstep; each symbol step ends with a ballot of the lanes whose current word ran dry, and those
 * lanes fetch consecutive stream words in ascending lane order. Reproduced including the tail
 * over-reads.
 * ---------------------------------------------------------------------------------------------- */
static inline uint32_t enc_read(const pcr_oracle_stream *s, int64_t i)
{
    return (i >= 0 && i < s->encoded_words) ? s->encoded[i] : 0u;
}
static inline int32_t sep_read(const pcr_oracle_stream *s, int64_t i)
{
    return (i >= 0 && i < s->separate_words) ? s->separate[i] : 0;
}

typedef void (*point_sink)(void *ctx, int chain, int i, int32_t x, int32_t y, int32_t z);

static void decode_cluster(const pcr_oracle_stream *s, int64_t batch, int cluster, int npr,
                           point_sink sink, void *ctx)
{
    const pcr_gpu_batch *b = &s->batches[batch];
    const int32_t *tv = s->dt_values + b->decoder_table_offset;   /* render.cu:385, 392 */
    const int32_t *tl = s->dt_cwlen + b->decoder_table_offset;
    const int max_cw = (int)b->max_cw_len;
    const uint32_t mask = ((1u << max_cw) - 1u) << (32 - max_cw);  /* :386 */

    int64_t enc_ptr = b->encoding_batch_offset;                    /* :404 */
    if (cluster >= 1) enc_ptr += s->cluster_sizes[batch * 32 + cluster - 1]; /* :407-410 */

    uint32_t cur[32], nxt[32];
    int cur_bits[32];
    int64_t sep_ptr[32];
    int32_t prev[32][3];
    for (int l = 0; l < 32; ++l) {
        int tid = cluster * 32 + l;
        sep_ptr[l] = b->separate_batch_offset;                     /* :405 */
        if (tid != 0) sep_ptr[l] += s->separate_sizes[batch * 1024 + tid - 1]; /* :411-413 */
        cur[l] = enc_read(s, enc_ptr + l);                         /* :416 */
        nxt[l] = enc_read(s, enc_ptr + 32 + l);                    /* :417 */
        cur_bits[l] = 32;                                          /* :419 */
        const int32_t *sv = s->start_values + ((int64_t)batch * 1024 + tid) * 3; /* :421-424 */
        prev[l][0] = sv[0]; prev[l][1] = sv[1]; prev[l][2] = sv[2];
    }
    int64_t already_read = 64;                                     /* :418 */

    for (int i = 0; i < npr; ++i) {                                /* :428 */
        int32_t decoded[32][3];
        for (int j = 0; j < 3; ++j) {                              /* :430 */
            uint32_t warp_mask = 0;
            for (int l = 0; l < 32; ++l) {
                uint32_t L = cur_bits[l] == 32 ? cur[l] : (cur[l] << (32 - cur_bits[l]));  /* :431 */
                uint32_t R = cur_bits[l] == 32 ? 0u : (nxt[l] >> cur_bits[l]);             /* :432 */
                uint32_t key = ((L | R) & mask) >> (32 - max_cw);                           /* :433 */
                int32_t symbol = tv[key];                                                  /* :435 */
                int cw = (signed char)tl[key];              /* :393 narrows to char, :436 */
                decoded[l][j] = cw > 0 ? symbol : sep_read(s, sep_ptr[l]++);               /* :438 */
                cur_bits[l] -= abs(cw);                                                    /* :439 */
                if (cur_bits[l] <= 0) warp_mask |= 1u << l;                                /* :442-443 */
            }
            int offset = 0;
            for (int l = 0; l < 32; ++l) {
                if (warp_mask & (1u << l)) {                                               /* :444 */
                    /* offset == popc(warp_mask & lanes below l)  (:445; shift-by-32 == 0 on CUDA) */
                    cur[l] = nxt[l];                                                       /* :446 */
                    nxt[l] = enc_read(s, enc_ptr + already_read + offset);                 /* :447 */
                    cur_bits[l] += 32;                                                     /* :448 */
                    ++offset;
                }
            }
            already_read += offset;                                                        /* :450 */
        }
        for (int l = 0; l < 32; ++l) {
            /* :454-456, int32 wrap-around as on the GPU */
            int32_t x = (int32_t)((uint32_t)decoded[l][0] + (uint32_t)prev[l][0]);
            int32_t y = (int32_t)((uint32_t)decoded[l][1] + (uint32_t)prev[l][1]);
            int32_t z = (int32_t)((uint32_t)decoded[l][2] + (uint32_t)prev[l][2]);
            prev[l][0] = x; prev[l][1] = y; prev[l][2] = z;        /* :463 */
            sink(ctx, cluster * 32 + l, i, x, y, z);
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * The lane-major form the HIP path uses (k_transcode / k_render), restated on the CPU so that the
 * equivalence it rests on is checked without a GPU: (1) walk a cluster in lockstep exactly as above and write down, per
 * chain, the words it receives; (2) decode one chain from its own word sequence alone, as a plain bit stream.
 * tests/test_oracle_format.py asserts (2) == pcr_oracle_decode_batch for every chain, tail garbage included.
 * ---------------------------------------------------------------------------------------------- */
int pcr_oracle_lane_words(const pcr_oracle_stream *s, int64_t batch, int rows, uint32_t *out, int32_t *counts)
{
    const pcr_gpu_batch *b = &s->batches[batch];
    const int32_t *tl = s->dt_cwlen + b->decoder_table_offset;
    const int max_cw = (int)b->max_cw_len;
    const uint32_t mask = ((1u << max_cw) - 1u) << (32 - max_cw);
    for (int cluster = 0; cluster < 32; ++cluster) {
        int64_t enc_ptr = b->encoding_batch_offset;
        if (cluster >= 1) enc_ptr += s->cluster_sizes[batch * 32 + cluster - 1];
        uint32_t cur[32], nxt[32];
        int cur_bits[32], n[32];
        for (int l = 0; l < 32; ++l) {
            const int chain = cluster * 32 + l;
            cur[l] = enc_read(s, enc_ptr + l);
            nxt[l] = enc_read(s, enc_ptr + 32 + l);
            cur_bits[l] = 32;
            if (rows < 2) return -1;
            out[0 * 1024 + chain] = cur[l];
            out[1 * 1024 + chain] = nxt[l];
            n[l] = 2;
        }
        int64_t already_read = 64;
        for (int k = 0; k < 64 * 3; ++k) {
            uint32_t warp_mask = 0;
            for (int l = 0; l < 32; ++l) {
                uint32_t L = cur_bits[l] == 32 ? cur[l] : (cur[l] << (32 - cur_bits[l]));
                uint32_t R = cur_bits[l] == 32 ? 0u : (nxt[l] >> cur_bits[l]);
                uint32_t key = ((L | R) & mask) >> (32 - max_cw);
                cur_bits[l] -= abs((signed char)tl[key]);
                if (cur_bits[l] <= 0) warp_mask |= 1u << l;
            }
            int offset = 0;
            for (int l = 0; l < 32; ++l) {
                if (!(warp_mask & (1u << l))) continue;
                cur[l] = nxt[l];
                nxt[l] = enc_read(s, enc_ptr + already_read + offset);
                cur_bits[l] += 32;
                ++offset;
                if (n[l] >= rows) return -1;
                out[(size_t)n[l] * 1024 + cluster * 32 + l] = nxt[l];
                ++n[l];
            }
            already_read += offset;
        }
        for (int l = 0; l < 32; ++l) counts[cluster * 32 + l] = n[l];
    }
    return 0;
}

void pcr_oracle_decode_chain_from_lane_words(const pcr_oracle_stream *s, int64_t batch, int chain, const uint32_t *words,
                                             int num_words, int npr, int32_t *out_xyz)
{
    const pcr_gpu_batch *b = &s->batches[batch];
    const int32_t *tv = s->dt_values + b->decoder_table_offset;
    const int32_t *tl = s->dt_cwlen + b->decoder_table_offset;
    int64_t sep_ptr = b->separate_batch_offset + (chain ? s->separate_sizes[batch * 1024 + chain - 1] : 0);
    const int32_t *sv = s->start_values + ((int64_t)batch * 1024 + chain) * 3;
    int32_t prev[3] = { sv[0], sv[1], sv[2] };
    int64_t pos = 0;                                   /* bits consumed of the chain's own stream */
    for (int i = 0; i < npr; ++i) {
        for (int j = 0; j < 3; ++j) {
            /* the 12 bits at `pos` of the concatenated words (words[r * 1024] is the chain's r-th word) */
            uint64_t w = 0;
            const int64_t r = pos >> 5;
            for (int k = 0; k < 2; ++k) w = (w << 32) | (r + k < num_words ? words[(size_t)(r + k) * 1024] : 0u);
            const uint32_t key = (uint32_t)(w >> (64 - 12 - (pos & 31))) & 0xFFFu;
            const int cw = (signed char)tl[key];
            const int32_t d = cw > 0 ? tv[key] : sep_read(s, sep_ptr++);
            pos += abs(cw);
            prev[j] = (int32_t)((uint32_t)prev[j] + (uint32_t)d);
        }
        out_xyz[i * 3 + 0] = prev[0]; out_xyz[i * 3 + 1] = prev[1]; out_xyz[i * 3 + 2] = prev[2];
    }
}

static void sink_store(void *ctx, int chain, int i, int32_t x, int32_t y, int32_t z)
{
    int32_t *o = (int32_t *)ctx + ((size_t)chain * 64 + i) * 3;
    o[0] = x; o[1] = y; o[2] = z;
}

void pcr_oracle_decode_batch(const pcr_oracle_stream *s, int64_t batch, int npr, int32_t *out_xyz)
{
    for (int c = 0; c < 32; ++c) decode_cluster(s, batch, c, npr, sink_store, out_xyz);
}

/* ------------------------------------------------------------------------------------------------
 * rasterize variants
 * ---------------------------------------------------------------------------------------------- */
enum { MODE_BASIC, MODE_HQS_DEPTH, MODE_HQS_COLOR, MODE_TIE_COUNT };

typedef struct {
    const pcr_oracle_stream *s;
    const pcr_render_params *p;
    int mode;
    int64_t batch;
    int npr, use_double;
    double scale[3], offd[3];    /* double path: render.cu:399-400 */
    float scalef[3], offf[3];    /* float path:  render.cu:469-470 */
    uint64_t *fb;                /* basic / hqs depth: written; hqs colour: read */
    uint64_t *rg, *ba;
    uint32_t *tie;               /* MODE_TIE_COUNT: per pixel, bits 30:0 = points at the winning depth, bit 31 = one of them has another colour */
    size_t fb_elems;
    int shared_fb;               /* several host threads draw into fb (the multi-threaded CPU baseline): atomic min */
} raster_ctx;

static void sink_raster(void *vctx, int chain, int i, int32_t cx, int32_t cy, int32_t cz)
{
    raster_ctx *c = (raster_ctx *)vctx;
    const pcr_render_params *p = c->p;
    f4 pt;
    if (c->use_double) {   /* render.cu:459-461 */
        pt.x = (float)fma((double)cx, c->scale[0], c->offd[0]);
        pt.y = (float)fma((double)cy, c->scale[1], c->offd[1]);
        pt.z = (float)fma((double)cz, c->scale[2], c->offd[2]);
    } else {               /* render.cu:529-531 */
        pt.x = fmaf((float)cx, c->scalef[0], c->offf[0]);
        pt.y = fmaf((float)cy, c->scalef[1], c->offf[1]);
        pt.z = fmaf((float)cz, c->scalef[2], c->offf[2]);
    }
    pt.w = 1.0f;
    /* render.cu:453: local (ctx-relative) point index — only used to address Colors */
    uint64_t index = (uint64_t)c->batch * PCR_POINTS_PER_BATCH + (uint64_t)chain * 64 + (uint64_t)i;

    /* rasterize(): render.cu:276-303 / hqs depth.cu:127-154 / hqs render.cu:274-316 */
    f4 pos = mat_mul(p->transform, pt);
    float nx = pos.x / pos.w, ny = pos.y / pos.w;
    /* inside test :296. Written so that NaNs are rejected (the reference's form would index the
     * framebuffer with an undefined pixel id: SURVEY Appendix C.2). */
    if (!(pos.w > 0.0f && nx >= -1.0f && nx <= 1.0f && ny >= -1.0f && ny <= 1.0f)) return;
    float ix = fmaf(nx, 0.5f, 0.5f) * (float)p->width;    /* :283 */
    float iy = fmaf(ny, 0.5f, 0.5f) * (float)p->height;
    int64_t pix = (int64_t)(int)ix + (int64_t)(int)iy * p->width;   /* :284-285 */
    if (pix < 0 || (size_t)pix >= c->fb_elems) return;
    uint32_t depth = f32_bits(pos.w);                                /* :287 */

    if (c->mode == MODE_TIE_COUNT) {
        /* Second walk over a finished basic frame: which pixels had several points at the winning depth? Where those
         * points differ in colour the reference's result depends on the order its threads ran in (its pre-read compares
         * depth<<32|pointIndex, render.cu:294-299); "plain min" picks the smallest colour. */
        uint64_t win = c->fb[pix];
        if ((uint32_t)(win >> 32) == depth) {
            uint32_t t = c->tie[pix];
            if ((t & 0x7FFFFFFFu) != 0x7FFFFFFFu) ++t;
            if (pcr_oracle_decode_bc1(index, c->s->colors) != (uint32_t)win) t |= 0x80000000u;
            c->tie[pix] = t;
        }
        return;
    }

    if (c->mode == MODE_HQS_COLOR) {
        uint64_t old = c->fb[pix];
        float old_depth = bits_f32((uint32_t)(old >> 32));
        if ((double)pos.w <= (double)old_depth * 1.01) {             /* hqs render.cu:296 */
            uint32_t rgba = c->s->color_format == PCR_COLOR_BC7 ? pcr_oracle_decode_bc7(index, c->s->colors)     /* :297-303 */
                                                               : pcr_oracle_decode_bc1(index, c->s->colors);
            uint64_t r = rgba & 255u, g = (rgba >> 8) & 255u, b = (rgba >> 16) & 255u;
            c->rg[pix] += (r << 32) | g;                              /* :309-310 */
            c->ba[pix] += (b << 32) | 1u;                             /* :311-312 */
        }
        return;
    }

    uint64_t key;
    if (c->mode == MODE_BASIC) {
        /* mem_iter writes depth<<32|colour whatever the debug flags say (:299 overwrites newPoint; the
         * flags only change the operand of the pre-read compare). Canonical result = min over all inside
         * points of depth<<32|colour: the reference's non-atomic pre-read filter
         * (:297-298) only prunes candidates that cannot win, except at exact depth ties. */
        uint64_t hi = (uint64_t)depth << 32;
        if (hi > (__atomic_load_n(&c->fb[pix], __ATOMIC_RELAXED) | 0xFFFFFFFFull)) return;
        key = hi | pcr_oracle_decode_bc1(index, c->s->colors);       /* :299 */
    } else {
        uint32_t payload = 0;                                                               /* hqs depth.cu:144 */
        if (p->show_num_points)      payload = (uint32_t)c->npr;                            /* depth.cu:139-140 */
        else if (p->colorize_chunks) payload = (uint32_t)(c->s->batch_index_base + c->batch); /* :141-142 blockIdx.x */
        key = ((uint64_t)depth << 32) | payload;
    }
    if (!c->shared_fb) {
        if (key < c->fb[pix]) c->fb[pix] = key;                       /* :300 atomicMin */
    } else {
        /* several host threads draw into ONE framebuffer (pcr_oracle_render_basic_mt): the same atomicMin as a compare-exchange loop */
        uint64_t old = __atomic_load_n(&c->fb[pix], __ATOMIC_RELAXED);
        while (key < old && !__atomic_compare_exchange_n(&c->fb[pix], &old, key, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
    }
}

static void render_range_ex(const pcr_oracle_stream *s, const pcr_render_params *p, int mode,
                            int64_t first, int64_t count, uint64_t *fb, uint64_t *rg, uint64_t *ba,
                            pcr_render_stats *stats, int shared_fb)
{
    raster_ctx c;
    memset(&c, 0, sizeof c);
    c.s = s; c.p = p; c.mode = mode; c.fb = fb; c.rg = rg; c.ba = ba; c.shared_fb = shared_fb;
    c.fb_elems = pcr_fb_elems(p->width, p->height);
    for (int64_t bi = first; bi < first + count; ++bi) {
        const pcr_gpu_batch *b = &s->batches[bi];
        int npr = 0, use_double = 0;
        if (stats) stats->batches_total++;
        int variant = (mode == MODE_BASIC || mode == MODE_TIE_COUNT) ? PCR_ORACLE_MEM_ITER : PCR_ORACLE_HQS;
        if (!pcr_oracle_batch_lod(b, p, variant, &npr, &use_double)) {
            if (stats) stats->batches_culled++;
            continue;
        }
        if (stats) {
            stats->points_iterated += (int64_t)PCR_WORKGROUP_SIZE * npr;
            stats->batches_double += use_double;
        }
        c.batch = bi; c.npr = npr; c.use_double = use_double;
        c.scale[0] = b->scale_x; c.scale[1] = b->scale_y; c.scale[2] = b->scale_z;
        c.offd[0] = b->offset_x - b->las_min_x;
        c.offd[1] = b->offset_y - b->las_min_y;
        c.offd[2] = b->offset_z - b->las_min_z;
        for (int k = 0; k < 3; ++k) { c.scalef[k] = (float)c.scale[k]; c.offf[k] = (float)c.offd[k]; }
        for (int cl = 0; cl < 32; ++cl) decode_cluster(s, bi, cl, npr, sink_raster, &c);
    }
}

static void render_range(const pcr_oracle_stream *s, const pcr_render_params *p, int mode,
                         int64_t first, int64_t count, uint64_t *fb, uint64_t *rg, uint64_t *ba,
                         pcr_render_stats *stats)
{
    render_range_ex(s, p, mode, first, count, fb, rg, ba, stats, 0);
}

void pcr_oracle_render_basic(const pcr_oracle_stream *s, const pcr_render_params *p,
                             int64_t first, int64_t count, uint64_t *fb, pcr_render_stats *stats)
{
    render_range(s, p, MODE_BASIC, first, count, fb, NULL, NULL, stats);
}

void pcr_oracle_render_hqs_depth(const pcr_oracle_stream *s, const pcr_render_params *p,
                                 int64_t first, int64_t count, uint64_t *fb, pcr_render_stats *stats)
{
    render_range(s, p, MODE_HQS_DEPTH, first, count, fb, NULL, NULL, stats);
}

void pcr_oracle_render_hqs_color(const pcr_oracle_stream *s, const pcr_render_params *p,
                                 int64_t first, int64_t count, const uint64_t *fb,
                                 uint64_t *rg, uint64_t *ba, pcr_render_stats *stats)
{
    render_range(s, p, MODE_HQS_COLOR, first, count, (uint64_t *)fb, rg, ba, stats);
}

/* Depth ties of a finished basic frame `fb` (rendered from the same batches with the same parameters): pixels whose
 * winning depth was reached by more than one point, and those among them where the tied points differ in colour. */
int pcr_oracle_count_depth_ties(const pcr_oracle_stream *s, const pcr_render_params *p, int64_t first, int64_t count,
                                const uint64_t *fb, int64_t *tie_pixels, int64_t *tie_pixels_other_colour)
{
    size_t n = pcr_fb_elems(p->width, p->height);
    uint32_t *tie = (uint32_t *)calloc(n, sizeof *tie);
    if (!tie) return -1;
    raster_ctx c;
    memset(&c, 0, sizeof c);
    /* render_range sets the per-batch fields; the tie plane travels in a context of the same shape */
    {
        c.s = s; c.p = p; c.mode = MODE_TIE_COUNT; c.fb = (uint64_t *)fb; c.tie = tie; c.fb_elems = n;
        for (int64_t bi = first; bi < first + count; ++bi) {
            const pcr_gpu_batch *b = &s->batches[bi];
            int npr = 0, use_double = 0;
            if (!pcr_oracle_batch_lod(b, p, PCR_ORACLE_MEM_ITER, &npr, &use_double)) continue;
            c.batch = bi; c.npr = npr; c.use_double = use_double;
            c.scale[0] = b->scale_x; c.scale[1] = b->scale_y; c.scale[2] = b->scale_z;
            c.offd[0] = b->offset_x - b->las_min_x;
            c.offd[1] = b->offset_y - b->las_min_y;
            c.offd[2] = b->offset_z - b->las_min_z;
            for (int k = 0; k < 3; ++k) { c.scalef[k] = (float)c.scale[k]; c.offf[k] = (float)c.offd[k]; }
            for (int cl = 0; cl < 32; ++cl) decode_cluster(s, bi, cl, npr, sink_raster, &c);
        }
    }
    int64_t any = 0, other = 0;
    for (size_t i = 0; i < n; ++i) {
        if ((tie[i] & 0x7FFFFFFFu) >= 2) ++any;
        if (tie[i] & 0x80000000u) ++other;
    }
    free(tie);
    *tie_pixels = any; *tie_pixels_other_colour = other;
    return 0;
}

/* Multi-threaded basic render for the CPU baseline: batches handed out one at a time (an atomic counter), ONE
 * framebuffer that every thread draws into with the CPU form of the kernels' atomicMin (a compare-exchange loop behind the same
 * pre-read filter, rasterize() above). min is associative and commutative: the frame does not depend on who drew what, nor when.
 * (Round 3 gave every thread a framebuffer of its own and merged them: at 256 threads that is 4 GB of page faults and a merge
 * for 10 ms of rendering per thread -- the row "all host cores" of bench.py measured the allocator.) */
typedef struct {
    const pcr_oracle_stream *s; const pcr_render_params *p;
    int64_t first, count; uint64_t *fb; int64_t next;
} mt_shared;
typedef struct { mt_shared *sh; pcr_render_stats stats; } mt_job;

static void *mt_worker(void *arg)
{
    mt_job *j = (mt_job *)arg;
    mt_shared *sh = j->sh;
    for (;;) {
        const int64_t b = __atomic_fetch_add(&sh->next, 1, __ATOMIC_RELAXED);
        if (b >= sh->count) break;
        render_range_ex(sh->s, sh->p, MODE_BASIC, sh->first + b, 1, sh->fb, NULL, NULL, &j->stats, 1);
    }
    return NULL;
}

int pcr_oracle_render_basic_mt(const pcr_oracle_stream *s, const pcr_render_params *p,
                               int64_t first, int64_t count, uint64_t *fb, int nthreads,
                               pcr_render_stats *stats)
{
    if (nthreads < 1) nthreads = 1;
    if ((int64_t)nthreads > count && count > 0) nthreads = (int)count;
    mt_shared sh;
    memset(&sh, 0, sizeof sh);
    sh.s = s; sh.p = p; sh.first = first; sh.count = count; sh.fb = fb;
    mt_job *jobs = (mt_job *)calloc((size_t)nthreads, sizeof *jobs);
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof *th);
    if (!jobs || !th) { free(jobs); free(th); return -1; }
    int started = 0;
    for (int t = 0; t < nthreads; ++t) {
        jobs[t].sh = &sh;
        if (pthread_create(&th[t], NULL, mt_worker, &jobs[t])) break;       /* (fewer threads draw the same frame) */
        ++started;
    }
    if (!started) mt_worker(&jobs[0]);
    for (int t = 0; t < (started ? started : 1); ++t) {
        if (started) pthread_join(th[t], NULL);
        if (stats) {
            stats->batches_total += jobs[t].stats.batches_total;
            stats->batches_culled += jobs[t].stats.batches_culled;
            stats->points_iterated += jobs[t].stats.points_iterated;
            stats->batches_double += jobs[t].stats.batches_double;
        }
    }
    free(jobs); free(th);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * resolve: huffman_mem_iter_cuda/resolve.cu:149-191, huffman_hqs/resolve.cu:2-47
 * ---------------------------------------------------------------------------------------------- */
void pcr_oracle_resolve_basic(const pcr_render_params *p, const uint64_t *fb, uint32_t *rgba)
{
    for (int y = 0; y < p->height; ++y)
        for (int x = 0; x < p->width; ++x) {
            int pix = x + y * p->width;
            uint32_t id = (uint32_t)fb[pix];
            uint32_t color = PCR_BACKGROUND_COLOR;
            if (id < 0xFFFFFFFFu) {
                if (p->show_num_points) {
                    int npr = (int)id;
                    uint32_t shade = (uint32_t)(((double)(float)npr / 64.0) * 255.0);   /* :170 */
                    color = (shade << 24) | (shade << 16) | (shade << 8) | shade;
                } else if (p->colorize_chunks) {
                    color = id * 1234567u;                                             /* :174 */
                } else {
                    color = id;                                                        /* :181 (BC1 mode) */
                }
            }
            rgba[pix] = color;
        }
}

void pcr_oracle_resolve_hqs(const pcr_render_params *p, const uint64_t *fb,
                            const uint64_t *rg, const uint64_t *ba, uint32_t *rgba)
{
    for (int y = 0; y < p->height; ++y)
        for (int x = 0; x < p->width; ++x) {
            int pix = x + y * p->width;
            uint32_t id = (uint32_t)fb[pix];
            uint32_t color = PCR_BACKGROUND_COLOR;
            if (id < 0xFFFFFFFFu) {
                if (p->show_num_points) {
                    int npr = (int)id;
                    uint32_t shade = (uint32_t)(((double)(float)npr / 512.0) * 255.0); /* hqs resolve.cu:24 */
                    color = (shade << 24) | (shade << 16) | (shade << 8) | shade;
                } else if (p->colorize_chunks) {
                    color = id * 1234567u;
                } else {
                    uint32_t cnt = (uint32_t)(ba[pix] & 0xFFFFFFFFull);                /* :33 */
                    if (cnt == 0) {
                        color = 0; /* unreachable for a consistent depth/colour pair; defined instead of dividing by 0 */
                    } else {
                        uint32_t r = (uint32_t)(rg[pix] >> 32) / cnt;
                        uint32_t g = (uint32_t)(rg[pix] & 0xFFFFFFFFull) / cnt;
                        uint32_t b = (uint32_t)(ba[pix] >> 32) / cnt;
                        color = (b << 16) | (g << 8) | r;                              /* :37 */
                    }
                }
            }
            rgba[pix] = color;
        }
}

/* ------------------------------------------------------------------------------------------------
 * 10-10-10 path: modules/compute_loop_las_cuda/render.cu
 * ---------------------------------------------------------------------------------------------- */
int pcr_oracle_las_level(const pcr_xyz_batch *b, const pcr_render_params *p)
{
    const float bmin[3] = { b->min_x, b->min_y, b->min_z }, bmax[3] = { b->max_x, b->max_y, b->max_z };
    if (p->enable_frustum_culling && !intersects_frustum(p, bmin, bmax)) return -1;   /* render.cu:153-155 */
    /* :157-186, same expressions as the Huffman kernels' LOD block */
    f4 ctr = { 0.5f * (bmin[0] + bmax[0]), 0.5f * (bmin[1] + bmax[1]), 0.5f * (bmin[2] + bmax[2]), 1.0f };
    float dx = bmin[0] - bmax[0], dy = bmin[1] - bmax[1], dz = bmin[2] - bmax[2];
    float rad = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    f4 vc = mat_mul(p->world_view, ctr);
    f4 ve = { vc.x + rad, vc.y + 0.0f, vc.z + 0.0f, vc.w + 0.0f };
    f4 pc = mat_mul(p->proj, vc), pe = mat_mul(p->proj, ve);
    float fw = (float)p->width, fh = (float)p->height;
    float scx = fw * (0.5f * (pc.x / pc.w + 1.0f)), scy = fh * (0.5f * (pc.y / pc.w + 1.0f));
    float sex = fw * (0.5f * (pe.x / pe.w + 1.0f)), sey = fh * (0.5f * (pe.y / pe.w + 1.0f));
    float ddx = sex - scx, ddy = sey - scy;
    float px = sqrtf(fmaf(ddy, ddy, ddx * ddx));
    if (px < 100.0f) return 4;            /* :187-197 */
    if (px < 200.0f) return 3;
    if (px < 500.0f) return 2;
    if (px < 10000.0f) return 1;
    return 0;
}

void pcr_oracle_render_las(const pcr_xyz_batch *batches, int64_t num_batches, const uint32_t *xyz12,
                           const uint32_t *xyz8, const uint32_t *xyz4, const pcr_render_params *p,
                           uint64_t *fb, pcr_render_stats *stats)
{
    const size_t fb_elems = pcr_fb_elems(p->width, p->height);
    for (int64_t b = 0; b < num_batches; ++b) {
        const pcr_xyz_batch *g = &batches[b];
        if (stats) stats->batches_total++;
        int level = pcr_oracle_las_level(g, p);
        if (level < 0) { if (stats) stats->batches_culled++; continue; }
        if (b == num_batches - 1) continue;                       /* render.cu:201-202 */
        if (stats) stats->points_iterated += PCR_POINTS_PER_BATCH;
        const float bs[3] = { g->max_x - g->min_x, g->max_y - g->min_y, g->max_z - g->min_z };   /* :145 */
        const float lo[3] = { g->min_x, g->min_y, g->min_z };
        const float div = level >= 2 ? 1024.0f : 1073741824.0f;  /* STEPS_10BIT / STEPS_30BIT */
        const float sc[3] = { bs[0] / div, bs[1] / div, bs[2] / div };
        for (int64_t k = 0; k < PCR_POINTS_PER_BATCH; ++k) {
            const uint32_t index = (uint32_t)(b * PCR_POINTS_PER_BATCH + k);   /* :331 */
            uint32_t X, Y, Z;
            const uint32_t b4 = xyz4[index];
            if (level >= 2) {                                      /* :381-392 */
                X = b4 & 1023u; Y = (b4 >> 10) & 1023u; Z = (b4 >> 20) & 1023u;
            } else {
                const uint32_t b8 = xyz8[index];
                X = ((b4 & 1023u) << 20) | ((b8 & 1023u) << 10);
                Y = (((b4 >> 10) & 1023u) << 20) | (((b8 >> 10) & 1023u) << 10);
                Z = (((b4 >> 20) & 1023u) << 20) | (((b8 >> 20) & 1023u) << 10);
                if (level == 0) {                                  /* :333-356 */
                    const uint32_t b12 = xyz12[index];
                    X |= b12 & 1023u; Y |= (b12 >> 10) & 1023u; Z |= (b12 >> 20) & 1023u;
                }
            }
            f4 pt = { fmaf((float)X, sc[0], lo[0]), fmaf((float)Y, sc[1], lo[1]), fmaf((float)Z, sc[2], lo[2]), 1.0f };
            /* rasterize, render.cu:108-128 */
            f4 pos = mat_mul(p->transform, pt);
            float nx = pos.x / pos.w, ny = pos.y / pos.w;
            if (!(pos.w > 0.0f && nx >= -1.0f && nx <= 1.0f && ny >= -1.0f && ny <= 1.0f)) continue;
            float ix = fmaf(nx, 0.5f, 0.5f) * (float)p->width, iy = fmaf(ny, 0.5f, 0.5f) * (float)p->height;
            int64_t pix = (int64_t)(int)ix + (int64_t)(int)iy * p->width;
            if (pix < 0 || (size_t)pix >= fb_elems) continue;
            uint64_t key = ((uint64_t)f32_bits(pos.w) << 32) | index;
            if (key < fb[pix]) fb[pix] = key;
        }
    }
}

void pcr_oracle_resolve_las(const pcr_render_params *p, const uint64_t *fb, const uint32_t *rgba_points,
                            uint32_t *rgba)
{
    /* resolve.cu: every pixel (the reference launches floor(w/16) x floor(h/16) tiles and leaves the rest untouched) */
    for (int i = 0; i < p->width * p->height; ++i) {
        uint32_t id = (uint32_t)fb[i];
        rgba[i] = id < 0x7FFFFFFFu ? rgba_points[id] : PCR_BACKGROUND_COLOR;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Scalar per-chain decoder: include/huffman.h:433-477 (decompress_udtype_subarray_fast_pjn_idea)
 * ---------------------------------------------------------------------------------------------- */
void pcr_oracle_decode_chain(const uint32_t *words, int64_t num_words, const int32_t *separate,
                             const int32_t *dt_values, const int32_t *dt_cwlen,
                             int num_symbols, int32_t *out)
{
    const int nb = 32, max_cw = PCR_MAX_CW_LEN;
    int sep_ptr = 0;
    int64_t cur_ptr = 0;
    int cur_bits = nb;
    const uint32_t mask = ((1u << max_cw) - 1u) << (nb - max_cw);
#define W(i) ((i) < num_words ? words[i] : 0u)   /* huffman.h:445 pushes one dummy 0 word */
    for (int k = 0; k < num_symbols; ++k) {
        uint32_t L = cur_bits == nb ? W(cur_ptr) : (W(cur_ptr) << (nb - cur_bits));
        uint32_t R = cur_bits == nb ? 0u : (W(cur_ptr + 1) >> cur_bits);
        uint32_t key = ((L | R) & mask) >> (nb - max_cw);
        int32_t symbol = dt_values[key];
        int len = dt_cwlen[key];
        int cw = abs(len);
        out[k] = len > 0 ? symbol : separate[sep_ptr++];
        int mb = cw < cur_bits ? cw : cur_bits;     /* :464-472 */
        cur_bits -= mb; cw -= mb;
        if (cw < cur_bits) cur_bits -= cw;
        else { cur_ptr += 1; cur_bits = cur_bits + nb - cw; }
    }
#undef W
}

/* ------------------------------------------------------------------------------------------------
 * .huffman parsing: HuffmanLasLoader.h:57-85 (header), BatchDumpData.h:60-149 (record),
 * HuffmanLasLoader.cpp:176-299 (flat arrays with running offsets)
 * ---------------------------------------------------------------------------------------------- */
struct pcr_oracle_file {
    pcr_oracle_stream s;
    pcr_gpu_batch *batches;
    int32_t *start_values, *separate, *separate_sizes, *dt_values, *dt_cwlen, *cluster_sizes;
    uint32_t *encoded;
    uint8_t *colors;
};

void pcr_oracle_file_free(pcr_oracle_file *f)
{
    if (!f) return;
    free(f->batches); free(f->start_values); free(f->separate); free(f->separate_sizes);
    free(f->dt_values); free(f->dt_cwlen); free(f->cluster_sizes); free(f->encoded); free(f->colors);
    free(f);
}

const pcr_oracle_stream *pcr_oracle_file_stream(const pcr_oracle_file *f) { return &f->s; }

#define FAIL(...) do { if (err) snprintf(err, errlen, __VA_ARGS__); pcr_oracle_file_free(f); return NULL; } while (0)

pcr_oracle_file *pcr_oracle_file_parse(const void *bytes, size_t n, char *err, size_t errlen)
{
    const uint8_t *p = (const uint8_t *)bytes;
    pcr_oracle_file *f = (pcr_oracle_file *)calloc(1, sizeof *f);
    if (!f) return NULL;
    if (n < sizeof(pcr_file_header)) FAIL("file shorter than its header");
    pcr_file_header h;
    memcpy(&h, p, sizeof h);
    if (h.num_batches < 0 || h.num_points != h.num_batches * PCR_POINTS_PER_BATCH)
        FAIL("header: numPoints %lld is not numBatches %lld * 65536", (long long)h.num_points, (long long)h.num_batches);
    size_t off = sizeof h + 8 * (size_t)h.num_batches;
    if (n < off) FAIL("file shorter than its batch size table");
    const int64_t nB = h.num_batches;
    int64_t enc_words = h.encoded_bytes / 4 + PCR_ENCODED_PAD_WORDS;
    int64_t sep_words = h.separate_bytes / 4 + PCR_SEPARATE_PAD_WORDS;
    f->batches = (pcr_gpu_batch *)calloc((size_t)nB + 1, sizeof(pcr_gpu_batch));
    f->start_values = (int32_t *)calloc((size_t)nB * 3072 + 1, 4);
    f->separate_sizes = (int32_t *)calloc((size_t)nB * 1024 + 1, 4);
    f->dt_values = (int32_t *)calloc((size_t)nB * 4096 + 1, 4);
    f->dt_cwlen = (int32_t *)calloc((size_t)nB * 4096 + 1, 4);
    f->cluster_sizes = (int32_t *)calloc((size_t)nB * 32 + 1, 4);
    f->encoded = (uint32_t *)calloc((size_t)enc_words, 4);
    f->separate = (int32_t *)calloc((size_t)sep_words, 4);
    f->colors = (uint8_t *)calloc((size_t)nB * PCR_COLOR_BYTES_PER_BATCH_BC7 + 1, 1);      /* (room for either format) */
    size_t color_bytes = 0;                                 /* of this file's records: the first record tells */
    if (!f->batches || !f->start_values || !f->separate_sizes || !f->dt_values || !f->dt_cwlen ||
        !f->cluster_sizes || !f->encoded || !f->separate || !f->colors) FAIL("out of memory");

    int64_t enc_ptr = 0, sep_ptr = 0;
    for (int64_t b = 0; b < nB; ++b) {
        int64_t size;
        memcpy(&size, p + sizeof h + 8 * (size_t)b, 8);
        if (size < PCR_BATCH_FIXED_HEADER || off + (size_t)size > n) FAIL("batch %lld: record exceeds file", (long long)b);
        const uint8_t *r = p + off;
        int32_t hdr[5];
        memcpy(hdr, r, 20);
        double sc[3], of[3];
        float bmin[3], bmax[3], lmin[3], lmax[3];
        int32_t dt_size, num_clusters;
        memcpy(sc, r + 20, 24); memcpy(of, r + 44, 24);
        memcpy(bmin, r + 68, 12); memcpy(bmax, r + 80, 12);
        memcpy(lmin, r + 92, 12); memcpy(lmax, r + 104, 12);
        memcpy(&dt_size, r + 116, 4); memcpy(&num_clusters, r + 120, 4);
        if (hdr[1] != PCR_POINTS_PER_BATCH || hdr[2] != PCR_WORKGROUP_SIZE || hdr[3] != PCR_POINTS_PER_THREAD ||
            hdr[4] != PCR_CLUSTERS_PER_THREAD || dt_size != PCR_HUFFMAN_TABLE_SIZE || num_clusters != PCR_CLUSTERS_PER_BATCH)
            FAIL("batch %lld: unsupported geometry", (long long)b);
        size_t o = PCR_BATCH_FIXED_HEADER;
        size_t fixed = o + 4u * (3072 + 1024 + 4096 + 4096 + 32);
        if ((size_t)size < fixed) FAIL("batch %lld: record too short", (long long)b);
        memcpy(f->start_values + b * 3072, r + o, 3072 * 4); o += 3072 * 4;
        memcpy(f->separate_sizes + b * 1024, r + o, 1024 * 4); o += 1024 * 4;
        memcpy(f->dt_values + b * 4096, r + o, 4096 * 4); o += 4096 * 4;
        memcpy(f->dt_cwlen + b * 4096, r + o, 4096 * 4); o += 4096 * 4;
        memcpy(f->cluster_sizes + b * 32, r + o, 32 * 4); o += 32 * 4;
        int64_t ne = f->cluster_sizes[b * 32 + 31], ns = f->separate_sizes[b * 1024 + 1023];
        /* BatchDumpData.h:130-136: the colour array of a record is 8 (BC1) or 16 (BC7 mode 6) bytes per 16 points, fixed
         * when the reference is compiled (COLOR_COMPRESSION); here the record's size tells which */
        if (ne < 0 || ns < 0) FAIL("batch %lld: negative stream sizes", (long long)b);
        if (color_bytes == 0) {
            if ((size_t)size == fixed + 4u * (size_t)(ne + ns) + PCR_COLOR_BYTES_PER_BATCH) color_bytes = PCR_COLOR_BYTES_PER_BATCH;
            else if ((size_t)size == fixed + 4u * (size_t)(ne + ns) + PCR_COLOR_BYTES_PER_BATCH_BC7) color_bytes = PCR_COLOR_BYTES_PER_BATCH_BC7;
        }
        if (color_bytes == 0 || (size_t)size != fixed + 4u * (size_t)(ne + ns) + color_bytes)
            FAIL("batch %lld: record size mismatch", (long long)b);          /* BatchDumpData.h:148 */
        if (enc_ptr + ne > enc_words - PCR_ENCODED_PAD_WORDS || sep_ptr + ns > sep_words - PCR_SEPARATE_PAD_WORDS)
            FAIL("batch %lld: stream exceeds header byte counts", (long long)b);
        memcpy(f->encoded + enc_ptr, r + o, (size_t)ne * 4); o += (size_t)ne * 4;
        memcpy(f->separate + sep_ptr, r + o, (size_t)ns * 4); o += (size_t)ns * 4;
        memcpy(f->colors + (size_t)b * color_bytes, r + o, color_bytes);

        pcr_gpu_batch *g = &f->batches[b];      /* HuffmanLasLoader.cpp:188-211 */
        g->min_x = bmin[0]; g->min_y = bmin[1]; g->min_z = bmin[2];
        g->max_x = bmax[0]; g->max_y = bmax[1]; g->max_z = bmax[2];
        g->scale_x = sc[0]; g->scale_y = sc[1]; g->scale_z = sc[2];
        g->offset_x = of[0]; g->offset_y = of[1]; g->offset_z = of[2];
        g->las_min_x = lmin[0]; g->las_min_y = lmin[1]; g->las_min_z = lmin[2];
        g->las_max_x = lmax[0]; g->las_max_y = lmax[1]; g->las_max_z = lmax[2];
        g->encoding_batch_offset = enc_ptr;
        g->separate_batch_offset = sep_ptr;
        g->decoder_table_offset = b * 4096;
        g->cluster_sizes_offset = b * 32;
        g->max_cw_len = PCR_MAX_CW_LEN;        /* (long long) log2(dt_size) */
        enc_ptr += ne; sep_ptr += ns;
        off += (size_t)size;
    }
    f->s.num_batches = nB;
    f->s.batches = f->batches; f->s.start_values = f->start_values;
    f->s.encoded = f->encoded; f->s.encoded_words = enc_words;
    f->s.separate = f->separate; f->s.separate_words = sep_words;
    f->s.separate_sizes = f->separate_sizes; f->s.dt_values = f->dt_values; f->s.dt_cwlen = f->dt_cwlen;
    f->s.cluster_sizes = f->cluster_sizes; f->s.colors = f->colors; f->s.batch_index_base = 0;
    f->s.color_format = color_bytes == PCR_COLOR_BYTES_PER_BATCH_BC7 ? PCR_COLOR_BC7 : PCR_COLOR_BC1;
    return f;
}
