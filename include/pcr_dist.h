/*
 * pcr_dist.h — C ABI of the multi-GPU layer: libpcr_dist.so (C++ over RCCL, links libpcr_hip.so).
 *
 * The reference is single-GPU (src/main.cpp:61 creates one context on device 0; no NCCL/MPI call site anywhere: SURVEY
 * 2.3), so nothing here replaces a reference interface; it is the "batches statically sharded over the GPUs of a node,
 * per-pixel min merge of the partial framebuffers over RCCL/xGMI" of BASELINE.json's north_star, built the way the
 * reference's host code is built: C++ calling the device library through a C ABI.
 *
 * Model: one pcr_ctx per GPU holds a contiguous range of the stream's batches (pcr_dist_shard_range) and a full-size
 * private framebuffer. A frame is: every rank renders its shard, then ONE exchange step merges the partial frames --
 *     basic :  ncclReduce / ncclAllReduce(fb, ncclUint64, ncclMin)      min is associative and commutative on the packed
 *                                                                       {depth, colour} words: bit-identical to one GPU
 *     HQS   :  depth pass -> ncclAllReduce(fb, ncclUint64, ncclMin)     every rank tests against the GLOBAL depth
 *              colour pass -> ncclReduce(RG | BA, ncclUint64, ncclSum)  packed 2 x 32-bit sums, no carry between halves
 * in place, on the context's own stream (stream order against the render kernels, no host synchronisation).
 * The sliced exchange (large frames: 4096 x 4096 is 134 MB, which a reduce moves whole into rank 0): the frame is cut into
 * N equal slices, rank r ends up with the merged slice r, resolves it, and only the RGBA8 pixels travel on --
 *     basic :  ncclReduceScatter(fb, ncclUint64, ncclMin) -> resolve of the own slice -> ncclGather / ncclAllGather(rgba)
 *              (or, PCR_DIST_EXCHANGE_SLICED_P2P: ncclAllToAll of the slices + a local min: one slice per xGMI link, whatever
 *               algorithm RCCL would pick for the collective)
 *     HQS   :  depth all-reduce as above, then ncclReduceScatter(RG | BA, ncclSum) -> resolve of the own slice -> gather
 * Per link that is 1/N of the frame + 1/N of the image instead of the whole frame (DESIGN.md 6 has the byte counts).
 * RCCL's native unsigned 64-bit min is used, so the framebuffer keeps its all-ones "empty" word (the torch transport of
 * pcrhpg24_amd/dist.py needs pcr_set_int64_mergeable because torch exposes signed int64 only).
 *
 * Two ways to form the communicator:
 *   one process per GPU   rank 0 calls pcr_dist_unique_id, the launcher carries the 128 bytes to every rank (environment,
 *                         file, MPI, torch.distributed: the library does not care), every rank calls pcr_dist_create;
 *   one process, N GPUs   pcr_dist_create_local over N contexts on N different devices; collectives of the N ranks are
 *                         then issued between pcr_dist_group_begin / pcr_dist_group_end (pcr_render_dist does this).
 * All functions return PCR_OK or a negative PCR_E_* code; the message is available from pcr_dist_last_error().
 */
#ifndef PCR_DIST_H
#define PCR_DIST_H

#include "pcr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcr_dist pcr_dist;       /* one rank: a communicator bound to one pcr_ctx */

#define PCR_DIST_ID_BYTES 128           /* sizeof(ncclUniqueId) */
#define PCR_DIST_ALL (-1)               /* root value: every rank receives the merged frame (all-reduce) */

const char *pcr_dist_last_error(void);  /* thread-local */

/* Contiguous [first, first + count) of `units` (batches, or loader chunks) for `rank` of `world`; sizes differ by at most
 * one, earlier ranks take the larger shares. Pure arithmetic (the same split as pcrhpg24_amd/dist.py::shard_range). */
void pcr_dist_shard_range(int64_t units, int world, int rank, int64_t *first, int64_t *count);

int pcr_dist_unique_id(unsigned char id[PCR_DIST_ID_BYTES]);
int pcr_dist_create(pcr_ctx *ctx, const unsigned char id[PCR_DIST_ID_BYTES], int rank, int world, pcr_dist **out);
/* n contexts of this process, each on a different device, become ranks 0..n-1 of one communicator; out[n]. */
int pcr_dist_create_local(pcr_ctx *const *ctxs, int n, pcr_dist **out);
void pcr_dist_destroy(pcr_dist *d);
int pcr_dist_rank(const pcr_dist *d);
int pcr_dist_world(const pcr_dist *d);
/* How many ranks RCCL itself says the communicator has (ncclCommCount): what a bench line quotes as `rccl_ranks`, so that a
 * multi-GPU number can be told from one taken on a one-rank communicator. 0: no communicator, -1: RCCL refused. */
int pcr_dist_comm_ranks(const pcr_dist *d);

/* Bracket the collectives of several ranks issued by one thread (ncclGroupStart / ncclGroupEnd). */
int pcr_dist_group_begin(void);
int pcr_dist_group_end(void);

/* The exchange step, in place on the context's current framebuffers and stream. root = rank that receives the result, or
 * PCR_DIST_ALL. After a reduce the other ranks' buffers are unspecified (they are cleared by the next frame anyway). */
int pcr_dist_merge_min(pcr_dist *d, int root);          /* fb: u64 min */
int pcr_dist_merge_sum(pcr_dist *d, int root);          /* RG and BA: u64 sum */

/* Which exchange the whole-frame calls below use. AUTO (default) = REDUCE: the sliced forms have not met a peer on hardware yet
 * and stay an explicit choice until they have (PCR_DIST_SLICED_MIN_BYTES: the frame size from which SLICED is expected to win, 4096x4096).
 * pcr_dist_exchange: what AUTO resolves to (or the mode that was set). */
#define PCR_DIST_EXCHANGE_AUTO        0
#define PCR_DIST_EXCHANGE_REDUCE      1   /* ncclReduce / ncclAllReduce of the whole u64 frame */
#define PCR_DIST_EXCHANGE_SLICED      2   /* ncclReduceScatter + resolve of the own slice + gather of the RGBA8 image */
#define PCR_DIST_EXCHANGE_SLICED_P2P  3   /* the same with ncclAllToAll + a local min instead of the reduce-scatter (basic method) */
#define PCR_DIST_SLICED_MIN_BYTES     (64u << 20)
int pcr_dist_set_exchange(pcr_dist *d, int mode);
int pcr_dist_exchange(const pcr_dist *d);
/* Slice `rank` of a frame of `elems` words cut into `world` slices: [first, first + count), count = ceil(elems / world) rounded
 * up to an even number of words (16-byte aligned slices), the same for every rank; the last slices reach into the frame's
 * pad (pcr_framebuffer_capacity). Pure arithmetic (pcrhpg24_amd/dist.py::slice_elems is the same). */
void pcr_dist_slice_range(size_t elems, int world, int rank, size_t *first, size_t *count);
/* The sliced merges, in place: afterwards words [first, first + count) of this rank's fb (RG, BA) hold the merged slice, the
 * rest of the buffer is unspecified. With PCR_DIST_EXCHANGE_SLICED_P2P the merged fb slice is in a scratch buffer of the
 * communicator instead. pcr_dist_merged_slice returns where it is after the last pcr_dist_merge_min_sliced (device pointer)
 * and, for the P2P form, enqueues the local min over the received copies -- call it after pcr_dist_group_end when the
 * merge was issued inside a group. */
int pcr_dist_merge_min_sliced(pcr_dist *d);
int pcr_dist_merge_sum_sliced(pcr_dist *d);
const void *pcr_dist_merged_slice(pcr_dist *d);
/* Every rank has resolved the pixels of its slice into pcr_device_rgba at the slice's offset; assemble the image on `root`
 * (ncclGather) or on every rank (PCR_DIST_ALL: ncclAllGather). */
int pcr_dist_gather_image(pcr_dist *d, int root);

/* Whole frames: clear + shard render + merge (+ resolve where the result ends up), by the exchange that is set. These are
 * for one process per GPU. With one thread driving several ranks (pcr_dist_create_local) the collectives of the N ranks have to
 * be issued together instead: pcr_frame_begin / pcr_render_* for every rank, then pcr_dist_group_begin, pcr_dist_merge_* for
 * every rank, pcr_dist_group_end, then pcr_resolve_* (sliced: pcr_resolve_*_range on the own slice, and pcr_dist_gather_image
 * inside a second group) -- pcr_render_dist.cpp is that sequence. After a sliced frame no rank holds the whole merged u64
 * frame; the image (pcr_read_rgba) is complete on root (or on every rank). */
int pcr_dist_frame_basic(pcr_dist *d, const pcr_render_params *p, int root);
int pcr_dist_frame_hqs(pcr_dist *d, const pcr_render_params *p, int root);
/* The basic frame in its steady-state form: render + merge + one launch for resolve, CLEAR and the next frame's prepass
 * (pcr_frame_turn) on the ranks that hold the result, CLEAR + prepass on the others. Prime the loop with one
 * pcr_frame_begin(ctx, p, PCR_METHOD_BASIC) per rank; the merged u64 framebuffer is empty after the step (the image is not). */
int pcr_dist_step_basic(pcr_dist *d, const pcr_render_params *p, int root);

#ifdef __cplusplus
}
#endif
#endif
