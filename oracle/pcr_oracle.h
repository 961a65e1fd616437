/*
 * pcr_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C, lane-accurate, IEEE-strict restatement of the reference's Huffman decode + rasterize
 * kernels. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (pcrhpg24_amd/) never links, imports or calls it.
 *
 * PARITY PINNING STATUS
 *   pinned   : Huffman dictionary/table/bit-packing/per-chain decode semantics, against the
 *              reference's own include/huffman.h compiled unmodified (oracle/_ref, tests/test_ref_pin.py);
 *              Morton key against src/mymorton.h; BC1 decode against src/rgbcx.cpp's unpack_bc1.
 *   UNPINNED : the kernel-level semantics (32-lane interleaved fetch order, LOD, cull, projection,
 *              atomicMin packing, HQS). The reference holds no golden vectors for them and its CUDA
 *              kernels cannot run here ("parity unpinned" for these: see DESIGN.md). They follow the
 *              reference sources line by line, cited at each function below.
 */
#ifndef PCR_ORACLE_H
#define PCR_ORACLE_H

#include "pcr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Flat arrays exactly as the reference loader lays them out on the device
 * (modules/compute/HuffmanLasLoader.cpp:176-299). Reads past encoded_words / separate_words
 * are defined as 0 (the zero pad of HuffmanLasLoader.cpp:39-41, extended: SURVEY B.4). */
typedef struct pcr_oracle_stream {
    int64_t              num_batches;
    const pcr_gpu_batch *batches;          /* [nB]            */
    const int32_t       *start_values;     /* [nB*1024*3]     */
    const uint32_t      *encoded;          /* [encoded_words] */
    int64_t              encoded_words;
    const int32_t       *separate;         /* [separate_words]*/
    int64_t              separate_words;
    const int32_t       *separate_sizes;   /* [nB*1024] inclusive, batch-local */
    const int32_t       *dt_values;        /* [nB*4096]       */
    const int32_t       *dt_cwlen;         /* [nB*4096]       */
    const int32_t       *cluster_sizes;    /* [nB*32] inclusive, batch-local   */
    const uint8_t       *colors;           /* [nB*32768] BC1 or [nB*65536] BC7 mode 6 */
    int64_t              batch_index_base; /* global index of batch 0 (colorize_chunks payload when sharded) */
    int64_t              color_format;     /* PCR_COLOR_BC1 / PCR_COLOR_BC7 (0 reads as BC1) */
} pcr_oracle_stream;

enum { PCR_ORACLE_MEM_ITER = 0, PCR_ORACLE_HQS = 1 };

/* Cull + LOD decision of one batch. variant selects the `pixelSize /= 100.0f` (mem_iter) vs
 * `/= 100.0` (hqs) expression. Returns 0 if the batch is frustum-culled, else 1. */
int pcr_oracle_batch_lod(const pcr_gpu_batch *b, const pcr_render_params *p, int variant,
                         int *num_points_to_render, int *use_double);

/* Lane-accurate decode of one batch: writes npr points per chain as raw int32 XYZ to
 * out_xyz[(chain*64 + i)*3 + k], i < npr. */
void pcr_oracle_decode_batch(const pcr_oracle_stream *s, int64_t batch, int npr, int32_t *out_xyz);

uint32_t pcr_oracle_decode_bc1(uint64_t point_index, const uint8_t *colors);
uint32_t pcr_oracle_decode_bc7(uint64_t point_index, const uint8_t *colors);    /* BC7 mode 6 as the reference's kernels decode it */

/* huffman_mem_iter_cuda/render.cu kernel over batches [first, first+count). fb has pcr_fb_elems(w,h) u64. */
void pcr_oracle_render_basic(const pcr_oracle_stream *s, const pcr_render_params *p,
                             int64_t first, int64_t count, uint64_t *fb, pcr_render_stats *stats);
/* Same, batches striped over nthreads pthreads with private framebuffers + min merge (CPU baseline). */
int pcr_oracle_render_basic_mt(const pcr_oracle_stream *s, const pcr_render_params *p,
                               int64_t first, int64_t count, uint64_t *fb, int nthreads,
                               pcr_render_stats *stats);
/* Depth ties of a finished basic frame (fb = pcr_oracle_render_basic of the same batches and parameters): *tie_pixels =
 * pixels whose winning depth was reached by more than one point; *tie_pixels_other_colour = those where a tied point has
 * another colour than the winner — there the reference's result depends on thread order (render.cu:294-299 pre-reads
 * depth<<32|pointIndex) while this build and the oracle define "plain min of depth<<32|colour" (SURVEY Appendix C.5). */
int pcr_oracle_count_depth_ties(const pcr_oracle_stream *s, const pcr_render_params *p, int64_t first, int64_t count,
                                const uint64_t *fb, int64_t *tie_pixels, int64_t *tie_pixels_other_colour);
/* huffman_hqs/depth.cu */
void pcr_oracle_render_hqs_depth(const pcr_oracle_stream *s, const pcr_render_params *p,
                                 int64_t first, int64_t count, uint64_t *fb, pcr_render_stats *stats);
/* huffman_hqs/render.cu (reads fb, accumulates into rg/ba) */
void pcr_oracle_render_hqs_color(const pcr_oracle_stream *s, const pcr_render_params *p,
                                 int64_t first, int64_t count, const uint64_t *fb,
                                 uint64_t *rg, uint64_t *ba, pcr_render_stats *stats);

/* huffman_mem_iter_cuda/resolve.cu and huffman_hqs/resolve.cu; rgba has w*h u32. */
void pcr_oracle_resolve_basic(const pcr_render_params *p, const uint64_t *fb, uint32_t *rgba);
void pcr_oracle_resolve_hqs(const pcr_render_params *p, const uint64_t *fb,
                            const uint64_t *rg, const uint64_t *ba, uint32_t *rgba);

/* ---- 10-10-10 path: modules/compute_loop_las_cuda/render.cu:130-442 (non-prefetch loop) + resolve.cu -------------
 * Buffers as the reference's ComputeLasData holds them: point (batch b, iteration i, lane t) at b*65536 + i*1024 + t.
 * PARITY UNPINNED (no reference vectors; the method is disabled in the reference's main.cpp). The reference kernel
 * skips its last workgroup (render.cu:201-202); reproduced. fb key = depth<<32 | point index. */
void pcr_oracle_render_las(const pcr_xyz_batch *batches, int64_t num_batches, const uint32_t *xyz12,
                           const uint32_t *xyz8, const uint32_t *xyz4, const pcr_render_params *p,
                           uint64_t *fb, pcr_render_stats *stats);
/* 0: full 30 bit, 1: 20 bit, 2..4: 10 bit; -1: culled. (render.cu:153-197) */
int pcr_oracle_las_level(const pcr_xyz_batch *b, const pcr_render_params *p);
void pcr_oracle_resolve_las(const pcr_render_params *p, const uint64_t *fb, const uint32_t *rgba_points,
                            uint32_t *rgba);

/* Lane-major restatement (what k_transcode / k_render do on the GPU), for the CPU equivalence test:
 * pcr_oracle_lane_words walks batch `batch` in lockstep and stores the r-th word chain c receives at out[r * 1024 + c]
 * (rows rows available; counts[c] = words received; returns -1 if rows is too small);
 * pcr_oracle_decode_chain_from_lane_words decodes npr points of one chain from `words` = &out[chain] alone. */
int pcr_oracle_lane_words(const pcr_oracle_stream *s, int64_t batch, int rows, uint32_t *out, int32_t *counts);
void pcr_oracle_decode_chain_from_lane_words(const pcr_oracle_stream *s, int64_t batch, int chain, const uint32_t *words,
                                             int num_words, int npr, int32_t *out_xyz);

/* Per-chain scalar table decoder (include/huffman.h:433-477), used to pin table semantics. */
void pcr_oracle_decode_chain(const uint32_t *words, int64_t num_words, const int32_t *separate,
                             const int32_t *dt_values, const int32_t *dt_cwlen,
                             int num_symbols, int32_t *out);

/* --- .huffman file parsing (modules/compute/HuffmanLasLoader.{h,cpp}, include/BatchDumpData.h:60-149) --- */
typedef struct pcr_oracle_file pcr_oracle_file; /* owns the flat arrays */
pcr_oracle_file *pcr_oracle_file_parse(const void *bytes, size_t n, char *err, size_t errlen);
const pcr_oracle_stream *pcr_oracle_file_stream(const pcr_oracle_file *f);
void pcr_oracle_file_free(pcr_oracle_file *f);

#ifdef __cplusplus
}
#endif
#endif
