/*
 * pcr_encode.h — C ABI of the host-side (CPU) encoder and synthetic-scene generator: libpcr_host.so.
 *
 * This is the producer of every input the HIP path consumes; it restates the reference's offline
 * `preprocess` tool (src/preprocess.cpp:925-1279) natively:
 *   pad to a multiple of 65 536 by repeating the last point          src/preprocess.cpp:945-955
 *   per-chunk Morton sort                                             src/preprocess.cpp:959-977, src/mymorton.h:12-58
 *   1024 chains x 64 points, delta coding, xyz interleave             src/preprocess.cpp:211-227, 318-343
 *   one 12-bit-clipped Huffman code per batch + 4096-entry table      include/huffman.h:94-113, 180-240
 *   MSB-first packing into u32 + escape ("separate") stream           include/huffman.h:242-300
 *   32-lane (time, lane) interleave of the chain words                src/preprocess.cpp:540-587
 *   BC1 colour blocks, 16 points each (always 4-colour mode)          src/preprocess.cpp:282-297
 *   batch record + file layout                                        include/BatchDumpData.h:151-202, src/preprocess.cpp:1205-1234
 *
 * Nothing here touches the GPU. All functions return 0 on success, a negative code on failure and
 * leave a message retrievable with pcr_host_last_error() (thread-local).
 */
#ifndef PCR_ENCODE_H
#define PCR_ENCODE_H

#include "pcr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* LAS header quantities the encoder copies into every batch record (src/preprocess.cpp:93-115). */
typedef struct pcr_las_info {
    double scale[3];
    double offset[3];
    double min[3];
    double max[3];
} pcr_las_info;

typedef struct pcr_encode_stats {
    int64_t num_points_in;       /* before padding */
    int64_t num_points;          /* after padding  */
    int64_t num_batches;
    int64_t encoded_bytes;
    int64_t separate_bytes;
    int64_t cluster_bytes;
    int64_t escaped_symbols;
    int64_t total_symbols;
    int64_t file_bytes;
} pcr_encode_stats;

#define PCR_DEFAULT_CHUNK_POINTS 6553600  /* MAX_POINTS_PER_BATCH = 100 * 65536, Resources.h:10 */

const char *pcr_host_last_error(void);
void pcr_host_free(void *p);

/* Encoder flags. PCR_ENCODE_MORTON_SORT is the reference CLI's <sort> argument. PCR_ENCODE_PAD_TAILS is not in the
 * reference: it queues a zero word for each refill the decoder performs after a chain's last real word, which removes
 * the tail artefact of the reference's interleave (1-2 refills per chain for words that do not
 * exist desynchronise the other lanes). Files written with it decode exactly, with the reference's kernels too; the
 * default (flag clear) reproduces the reference's stream byte for byte, quirk included. */
#define PCR_ENCODE_MORTON_SORT 1
#define PCR_ENCODE_PAD_TAILS   2
/* colours as BC7 mode-6 blocks (16 B per 16 points) instead of BC1: the file a reference built with COLOR_COMPRESSION == 7
 * writes and reads (src/preprocess.cpp:299-316, :1129-1138). Only the HQS method draws such a stream (include/pcr_hip.h). */
#define PCR_ENCODE_BC7         4

/* Encode n points (int32 LAS coordinates + 0x00BBGGRR colours) into a complete .huffman file image.
 * Points are processed in chunks of chunk_points (<=0: default); each chunk is padded, optionally
 * Morton-sorted, cut into batches and encoded, exactly like `preprocess in.las out.huffman <sort>`
 * (flags = sort ? PCR_ENCODE_MORTON_SORT : 0). *out_bytes is malloc'ed; release with pcr_host_free.
 * nthreads <= 0: hardware concurrency. */
int pcr_encode_points(const int32_t *x, const int32_t *y, const int32_t *z, const uint32_t *color,
                      int64_t n, const pcr_las_info *las, int flags, int64_t chunk_points,
                      int nthreads, void **out_bytes, size_t *out_len, pcr_encode_stats *stats);

/* Deterministic synthetic scene: a heightfield surface over a square tile with LAS scale
 * 0.001 and int32 coordinates in [0, 1e6], smooth colour field. The scene of `total_points` points is
 * a jittered grid; this call materialises points [first, first+count) of it (row-major grid order, so
 * consecutive chunks are strips of the tile). Any of x/y/z/color may be NULL. */
int pcr_synth_points(int64_t total_points, uint64_t seed, int64_t first, int64_t count,
                     int32_t *x, int32_t *y, int32_t *z, uint32_t *color);
int pcr_synth_las_info(int64_t total_points, uint64_t seed, pcr_las_info *las);

/* Generate + encode points [first, first+count) of the synthetic scene chunk by chunk (chunks are
 * generated, sorted and encoded in parallel; peak memory ~ nthreads chunks + the output). `first`
 * must be a multiple of chunk_points so shards line up with the single-stream encoding. */
int pcr_synth_encode(int64_t total_points, uint64_t seed, int64_t first, int64_t count,
                     int64_t chunk_points, int nthreads, void **out_bytes, size_t *out_len,
                     pcr_encode_stats *stats);

/* 96-bit Morton key of the reference (src/mymorton.h:12-37) for already shifted coordinates. */
void pcr_morton_key(uint32_t x, uint32_t y, uint32_t z, uint32_t *hi, uint64_t *lo);

/* Building blocks exposed for the pinning tests (tests/test_ref_pin.py). ---------------------------
 * Build the clipped Huffman code of `n` symbols: writes the 4096-entry table and returns, for the
 * `num_query` symbols in `query`, their code word and signed length (negative: escape). */
int pcr_huffman_build(const int32_t *symbols, int64_t n, int32_t *dt_values, int32_t *dt_cwlen,
                      const int32_t *query, int64_t num_query, uint32_t *out_cw, int32_t *out_len);
/* Pack one chain given a dictionary (parallel arrays, any order). Outputs are malloc'ed. */
int pcr_pack_chain(const int32_t *symbols, int n, const int32_t *dict_symbols, const uint32_t *dict_cw,
                   const int32_t *dict_len, int64_t dict_n, uint32_t **words, int32_t *num_words,
                   int32_t **separate, int32_t *num_separate, int32_t **num_cw);
/* Table from a dictionary (include/huffman.h:220-240). */
int pcr_table_from_dict(const int32_t *dict_symbols, const uint32_t *dict_cw, const int32_t *dict_len,
                        int64_t dict_n, int32_t *dt_values, int32_t *dt_cwlen);
/* BC1-encode 16 colours (0x00BBGGRR) into 8 bytes, 4-colour mode only. */
void pcr_bc1_encode_block(const uint32_t *colors16, uint8_t *out8);
/* BC7 mode-6 encode 16 colours into 16 bytes (alpha 255). */
void pcr_bc7_encode_block(const uint32_t *colors16, uint8_t *out16);

/* 10-10-10 three-level quantisation of the `loop_las_cuda` method: what the reference's loader shader produces
 * (modules/compute/computeLasLoader.cs:147-190 getPoint, 193-252 computeBoundingBox, 255-357 processPoints), for
 * workgroups of 1024 x 64 points as its CUDA renderer expects (modules/compute/Resources.h:4-8): point (batch b,
 * iteration i, lane t) lives at index b*65536 + i*1024 + t of every array. Points are taken in input order.
 * Outputs are caller-allocated: batches[ceil(n/65536)], xyz12/xyz8/xyz4/rgba[ceil(n/65536)*65536] (unused slots 0). */
int pcr_las_quantize(const int32_t *x, const int32_t *y, const int32_t *z, const uint32_t *color, int64_t n,
                     const pcr_las_info *las, pcr_xyz_batch *batches, uint32_t *xyz12, uint32_t *xyz8,
                     uint32_t *xyz4, uint32_t *rgba, int nthreads);

/* Camera matrices as the reference builds them (include/OrbitControls.h:116-134, include/Camera.h:18-38,
 * modules/huffman_hqs/huffman_hqs.h:157-183): orbit (yaw, pitch, radius, target) -> world -> view =
 * inverse(world), proj = perspective(fovy, aspect, near, far) in double, narrowed to float, then
 * transform = proj*view, world_view = view, proj (float products as glm::mat4 operator*). */
int pcr_camera_orbit(double yaw, double pitch, double radius, const double target[3],
                     int width, int height, double fovy_deg, double near_plane, double far_plane,
                     pcr_render_params *out);

#ifdef __cplusplus
}
#endif
#endif
