/*
 * pcr_gpu_encode.h — C ABI of the GPU encoder in libpcr_hip.so (second half).
 *
 * Same job as pcr_encode_points (pcr_encode.h; the reference's offline `preprocess`, src/preprocess.cpp:925-1279),
 * run on the MI355X: Morton sort per chunk, per-chain deltas, per-batch symbol histogram, 12-bit-clipped Huffman code,
 * decoder table, MSB-first packing with the escape stream, 32-lane (time, lane) interleave, BC1 colour blocks. The
 * output is the same `.huffman` file image, byte for byte, as the CPU encoder produces for the same input and flags
 * (tests/test_gpu_encoder.py), so everything that pins the CPU encoder to the reference's own code pins this one too.
 */
#ifndef PCR_GPU_ENCODE_H
#define PCR_GPU_ENCODE_H

#include "pcr_encode.h"
#include "pcr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* x, y, z, color: host arrays of n points (int32 LAS coordinates, 0x00BBGGRR). flags: PCR_ENCODE_MORTON_SORT |
 * PCR_ENCODE_PAD_TAILS. chunk_points <= 0: PCR_DEFAULT_CHUNK_POINTS (it must be a multiple of 65 536 and at most 1024 x 65 536;
 * device scratch is ~16 MB per batch of a chunk).
 * *out_bytes is malloc'ed; release with pcr_gpu_encode_free. Work is enqueued on the context's stream and the call
 * returns when the image is complete. Returns PCR_OK or a negative PCR_E_* (message: pcr_last_error(ctx)). */
int  pcr_gpu_encode_points(pcr_ctx *ctx, const int32_t *x, const int32_t *y, const int32_t *z, const uint32_t *color,
                           int64_t n, const pcr_las_info *las, int flags, int64_t chunk_points,
                           void **out_bytes, size_t *out_len, pcr_encode_stats *stats);
void pcr_gpu_encode_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
