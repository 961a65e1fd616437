/*
 * pcr_hip.h — C ABI of libpcr_hip.so: the MI355X (gfx950) Huffman decode + rasterize path.
 *
 * This is the drop-in boundary for the reference's Huffman rendering methods. Each entry point names
 * the reference interface it replaces (paths relative to the reference checkout). The reference binds
 * its kernels through the CUDA driver API from C++ `Method`/`Resource` plugins; a maintainer replaces
 * those cu* calls with the functions below (INTEGRATION.md shows the adapter).
 *
 * Conventions: every function returns 0 on success or a negative PCR_E_* code; a human-readable
 * message is available from pcr_last_error(ctx) (ctx == NULL: message of the last failed pcr_create
 * on this thread). A context is externally synchronised (one caller thread at a time) and owns one
 * HIP stream on which all of its work is enqueued in call order; calls that return data to the host
 * synchronise that stream. No torch / C++ types cross this boundary.
 */
#ifndef PCR_HIP_H
#define PCR_HIP_H

#include "pcr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PCR_OK            0
#define PCR_E_ARG        -1   /* bad argument / call order */
#define PCR_E_FORMAT     -2   /* malformed batch record or unsupported geometry */
#define PCR_E_HIP        -3   /* a HIP runtime call failed */
#define PCR_E_NOMEM      -4
#define PCR_E_NODEVICE   -5   /* no gfx950-class device / kernels not loadable */

typedef struct pcr_ctx pcr_ctx;

/* ---- lifecycle -------------------------------------------------------------------------------
 * replaces: cuInit/cuDeviceGet/cuCtxCreate (src/main.cpp:58-62) and the per-method
 * CudaProgram JIT (include/CudaProgram.h:15-70; here the code object is compiled ahead of time). */
int         pcr_create(int device, pcr_ctx **out);
void        pcr_destroy(pcr_ctx *ctx);
const char *pcr_last_error(const pcr_ctx *ctx);
/* Borrow an existing HIP stream (hipStream_t) instead of the context's own; NULL restores it (so the legacy default
 * stream, whose handle is NULL, cannot be borrowed: a caller that works on it has to create a stream and move there). Does
 * not synchronise: calls made afterwards are enqueued on the new stream, ordering against work already enqueued on the
 * previous one is the caller's (HIP events). */
int         pcr_set_stream(pcr_ctx *ctx, void *hip_stream);
int         pcr_synchronize(pcr_ctx *ctx);

/* ---- resource side: HuffmanLasData (modules/compute/HuffmanLasLoader.{h,cpp}) -------------------
 * pcr_stream_begin  <- HuffmanLasData::load buffer creation (HuffmanLasLoader.cpp:32-77): allocates the
 *                      stream buffers for `hdr` (+ zero pads, PCR_ENCODED_PAD_WORDS / PCR_SEPARATE_PAD_WORDS)
 *                      and zero-fills them. batch_index_base = global index of this context's first batch
 *                      when the file is sharded across GPUs (0 otherwise).
 * pcr_upload_batch  <- HuffmanLasData::uploadBatch (HuffmanLasLoader.cpp:176-299): `blob` is one batch
 *                      record (include/BatchDumpData.h:151-202), borrowed for the call. Batches must
 *                      arrive in index order 0,1,2,... (the reference's running offsets assume it too).
 * pcr_upload_tail   <- no reference counterpart (multi-GPU only): the first words of the batch that
 *                      follows this shard in the global stream, so that the reference's tail over-reads
 *                       see the same bytes as on one GPU. At most the pad sizes.
 * pcr_stream_unload <- HuffmanLasData::unload (HuffmanLasLoader.cpp:152-174). */
int     pcr_stream_begin(pcr_ctx *ctx, const pcr_file_header *hdr, int64_t batch_index_base);
int     pcr_upload_batch(pcr_ctx *ctx, int64_t batch_index, const void *blob, size_t n);
/* A whole loader task at once (HuffmanLasData::process hands over <= 100 records, HuffmanLasLoader.cpp:301-313):
 * records first_index .. first_index+count-1, validated up front (nothing is uploaded if one is malformed), packed
 * into pinned staging memory and moved with nine large asynchronous copies. The records may be released on return;
 * the copies complete in stream order before any later render call. Behind the copies the context also writes its own
 * HBM layout of the new batches (lane-major word order, packed decoder tables: DESIGN.md 4) — loading work, once per
 * batch, nothing decoded. */
int     pcr_upload_batches(pcr_ctx *ctx, int64_t first_index, int64_t count, const void *const *blobs, const size_t *sizes);
int     pcr_upload_tail(pcr_ctx *ctx, const uint32_t *encoded_words, size_t n_encoded,
                        const int32_t *separate_words, size_t n_separate);
int     pcr_stream_unload(pcr_ctx *ctx);
int64_t pcr_batches_loaded(const pcr_ctx *ctx);   /* HuffmanLasData::numBatchesLoaded */
int64_t pcr_points_loaded(const pcr_ctx *ctx);    /* HuffmanLasData::numPointsLoaded  */

/* ---- method side: HuffmanMemIter / HuffmanHQS ---------------------------------------------------
 * pcr_set_image_size <- cuMemAlloc(&fb, 8*2048*2048) (+RG, BA) in the method constructors
 *                       (modules/huffman_hqs/huffman_hqs.h:52-54); sized by resolution here
 *                       (pcr_fb_elems(w,h) u64 each) and cleared.
 * pcr_clear          <- the CLEAR block (huffman_hqs.h:266-270): fb <- all ones, RG/BA <- 0.
 * pcr_render_basic   <- cuLaunchKernel(renderProg) of HuffmanMemIter::render
 *                       (modules/huffman_mem_iter_cuda/huffman_mem_iter_cuda.h:185-195; kernel render.cu:315-540).
 * pcr_render_hqs_depth / pcr_render_hqs_color
 *                    <- the two launches of HuffmanHQS::render (huffman_hqs.h:191-213;
 *                       kernels huffman_hqs/depth.cu:166-397, huffman_hqs/render.cu:328-562).
 * pcr_resolve_basic / pcr_resolve_hqs
 *                    <- the RESOLVE blocks (huffman_mem_iter_cuda.h:226-247, huffman_hqs.h:240-263;
 *                       kernels resolve.cu:149-191, huffman_hqs/resolve.cu:2-47). Output is a device
 *                       RGBA8 buffer (no GL surface), read back with pcr_read_rgba.
 * All render/resolve calls only enqueue work. */
int pcr_set_image_size(pcr_ctx *ctx, int width, int height);
int pcr_clear(pcr_ctx *ctx);
int pcr_render_basic(pcr_ctx *ctx, const pcr_render_params *p);
int pcr_render_hqs_depth(pcr_ctx *ctx, const pcr_render_params *p);
int pcr_render_hqs_color(pcr_ctx *ctx, const pcr_render_params *p);
int pcr_resolve_basic(pcr_ctx *ctx, const pcr_render_params *p);
int pcr_resolve_hqs(pcr_ctx *ctx, const pcr_render_params *p);

/* ---- the 10-10-10 method: ComputeLasData + ComputeLoopLasCUDA ("loop_las_cuda") ---------------------
 * pcr_las_begin   <- ComputeLasData::load buffer creation (modules/compute/ComputeLasLoader.cpp:14-38): the batch
 *                    table and the four 4-byte-per-point arrays (three 10-10-10 levels + colour) for
 *                    ceil(num_points/65536) batches, zero-filled.
 * pcr_las_upload  <- ComputeLasData::process: upload + quantisation dispatch (ComputeLasLoader.cpp:140-262, computeLasLoader.cs):
 *                    `count` batches starting at first_batch, already quantised (pcr_las_quantize, pcr_encode.h);
 *                    batches arrive in index order. Arrays are borrowed for the call.
 * pcr_las_unload  <- ComputeLasData::unload (ComputeLasLoader.cpp:114-131).
 * pcr_render_las  <- cuLaunchKernel(renderProg) of ComputeLoopLasCUDA::render (modules/compute_loop_las_cuda/
 *                    compute_loop_las_cuda.h:164-182; kernel render.cu:130-442): one workgroup per loaded batch, the
 *                    last one does not draw (render.cu:201-202). Keys are depth<<32 | point index.
 * pcr_resolve_las <- the resolve launch (compute_loop_las_cuda.h:185-207; kernel resolve.cu): pixel <- colour of the
 *                    winning point index, background 0x00443322; all pixels (the reference skips partial 16x16 tiles).
 * pcr_las_algorithmic_bytes: HBM bytes the last pcr_render_las had to read at least once (4/8/12 B per point by
 *                    level + 64 B per drawn batch); synchronises. */
int     pcr_las_begin(pcr_ctx *ctx, int64_t num_points);
int     pcr_las_upload(pcr_ctx *ctx, int64_t first_batch, int64_t count, const pcr_xyz_batch *batches,
                       const uint32_t *xyz12, const uint32_t *xyz8, const uint32_t *xyz4, const uint32_t *rgba);
int     pcr_las_unload(pcr_ctx *ctx);
int64_t pcr_las_batches_loaded(const pcr_ctx *ctx);
int     pcr_render_las(pcr_ctx *ctx, const pcr_render_params *p);
int     pcr_resolve_las(pcr_ctx *ctx, const pcr_render_params *p);
int64_t pcr_las_algorithmic_bytes(pcr_ctx *ctx);

/* Counters of the most recent render call (synchronises). */
int pcr_get_stats(pcr_ctx *ctx, pcr_render_stats *out);

/* ---- readback (the reference's cuMemcpyDtoH depth dump, huffman_hqs.h:217-237, generalised) ----- */
int pcr_read_framebuffer(pcr_ctx *ctx, uint64_t *host, size_t n_elems);          /* n_elems <= pcr_fb_elems(w,h) */
int pcr_read_accum(pcr_ctx *ctx, uint64_t *host_rg, uint64_t *host_ba, size_t n_elems);
int pcr_read_rgba(pcr_ctx *ctx, uint32_t *host, size_t n_pixels);                /* n_pixels <= w*h */

/* ---- multi-GPU plumbing (no reference counterpart) ------------------------------------
 * Device pointers of the context's buffers so a collective library (RCCL through torch.distributed)
 * can reduce them in place, or externally owned buffers to render into. */
/* What a collective library needs to merge partial frames in place (include/pcr_dist.h does it with RCCL): the HIP stream
 * the context enqueues on, its device ordinal and the length of each framebuffer in 64-bit words. */
void *pcr_get_stream(pcr_ctx *ctx);
int pcr_get_device(const pcr_ctx *ctx);
size_t pcr_framebuffer_elems(const pcr_ctx *ctx);
/* Words each of the three buffers can hold: pcr_framebuffer_elems + PCR_FRAME_PAD_ELEMS for the context's own buffers (the
 * pad holds the identity of min / sum: a collective over whole slices may include it), pcr_framebuffer_elems while
 * external buffers are in use. pcr_device_rgba: the RGBA8 image the resolves write (same capacity, in pixels). */
size_t pcr_framebuffer_capacity(const pcr_ctx *ctx);
void *pcr_device_rgba(pcr_ctx *ctx);
/* Raw device pointers to the context's framebuffers. CONTRACT: the library resolves and clears only the 64 x 16-pixel tiles its own
 * kernels wrote in (dirty tiles, pcr_frame_turn). Whoever holds one of these pointers may write anywhere behind the library's back
 * (a collective, a merge of their own), so from the first call of a getter on, EVERY frame turn and clear walks the whole frame --
 * correct whatever is written through the pointer, 10-20 us slower per 4096x4096 frame -- until pcr_framebuffer_private(ctx) says
 * the pointers are no longer written through. (pcr_merge_*, pcr_use_external_buffers, pcr_set_int64_mergeable and the 10-10-10
 * method drop the tile tracking by themselves until the next full clear.) */
void *pcr_device_framebuffer(pcr_ctx *ctx);
void *pcr_device_rg(pcr_ctx *ctx);
void *pcr_device_ba(pcr_ctx *ctx);
int   pcr_framebuffer_private(pcr_ctx *ctx);
int   pcr_use_external_buffers(pcr_ctx *ctx, void *dev_fb, void *dev_rg, void *dev_ba); /* NULLs: back to own */
/* fb[i] = min(fb[i], other[i]) over pcr_fb_elems elements; rg/ba[i] += other[i] (NULL: skip). */
int   pcr_merge_min(pcr_ctx *ctx, const void *dev_other_fb);
int   pcr_merge_sum(pcr_ctx *ctx, const void *dev_other_rg, const void *dev_other_ba);
/* x ^= 1<<63 on every framebuffer element: maps unsigned order to signed order so that a signed
 * int64 MIN all-reduce (the dtype torch.distributed exposes) computes the u64 min. */
int   pcr_flip_sign(pcr_ctx *ctx);

/* All-to-all form of the multi-GPU merge (pcrhpg24_amd/dist.py, merge="a2a"): the frame is cut into N contiguous slices
 * of slice_elems words; after an all-to-all a rank holds everyone's copy of the slice it owns, back to back.
 * pcr_merge_min_slices leaves their element-wise min in the first slice; pcr_resolve_basic_range resolves `count` pixels of
 * a framebuffer range as pcr_resolve_basic does (the basic resolve depends on the word only, resolve.cu:149-191) into
 * `rgba`, which an all-gather then assembles into the image. Both work on device pointers the caller owns and enqueue
 * on the context's stream. */
int pcr_merge_min_slices(pcr_ctx *ctx, void *slices, int nslices, size_t slice_elems);
int pcr_resolve_basic_range(pcr_ctx *ctx, const pcr_render_params *params, const void *fb, size_t count, void *rgba);
/* The same for the HQS resolve (huffman_hqs/resolve.cu:2-47 depends on the pixel's three words only): `count` pixels of
 * corresponding ranges of fb / RG / BA. */
int pcr_resolve_hqs_range(pcr_ctx *ctx, const pcr_render_params *params, const void *fb, const void *rg, const void *ba,
                          size_t count, void *rgba);

/* Ordering between two streams of the context's device without the system-scope release a default HIP event carries
 * (which writes the L2 back: the 16.6 MB framebuffer the next kernel is about to read). slot in [0, 8). A frame rendered
 * on one stream and merged on another uses two of these per frame. stream NULL = the context's stream. */
int pcr_fence_record(pcr_ctx *ctx, int slot, void *hip_stream);
int pcr_fence_wait(pcr_ctx *ctx, int slot, void *hip_stream);

/* HBM layout the context gives the next stream it loads (pcr_stream_begin fixes it). All hold the same bits and decode to
 * the same points; the Huffman decode runs every frame in all of them. Only what the layout's kernel reads is kept: the
 * raw cluster-interleaved words of the file, the int32/int8 decoder tables and the cluster prefix are released by the first
 * frame after the last batch was uploaded (pcr_upload_tail has to come before that frame).
 *   PCR_LAYOUT_WORDS          per chain the sequence of 32-bit words it consumes, kept compact: per 64 chains only the rows their
 *                             longest chain consumed (~2.9 B per point resident and read on the benchmark stream: the memory-lean
 *                             layout, 4.2 B per point with all side data against 3.7 in the file); the decode keeps a five-word queue
 *                             per lane. Loading it waits for the device once per 128 batches (the compact size is read back).
 *   PCR_LAYOUT_POINT_WINDOWS  (default) per point the 40 bits of its chain's stream that start at the point's first bit
 *                             (5 B per point, a u32 and a u8 plane): no queue in the decode, a point's first table read
 *                             off the dependent chain, fewer instructions per point for ~1.4x the bytes per frame, on a
 *                             kernel bound by issue and latency, not by HBM.
 *   PCR_LAYOUT_BOTH           both resident: either decode variant can draw a frame (pcr_set_render_variant).
 *   PCR_LAYOUT_AUTO           POINT_WINDOWS unless the stream's windows would exceed the budget of pcr_set_hbm_budget (bytes; 0 = no
 *                             budget, the default): then WORDS. pcr_stream_layout tells which one the loaded stream got. */
#define PCR_LAYOUT_WORDS 0
#define PCR_LAYOUT_POINT_WINDOWS 1
#define PCR_LAYOUT_BOTH 2
#define PCR_LAYOUT_AUTO 3
int pcr_set_stream_layout(pcr_ctx *ctx, int layout);
int pcr_set_hbm_budget(pcr_ctx *ctx, int64_t bytes);
int pcr_stream_layout(const pcr_ctx *ctx);
/* Which decode variant draws. AUTO (default): the one the stream's layout holds; with PCR_LAYOUT_BOTH the point-window
 * variant while the image has at most 4096 pixels (one LDS framebuffer window) per loaded batch, the packed-words variant
 * beyond that, where the frame is bound by global framebuffer traffic and the smaller stream wins. WORDS / POINT_WINDOWS
 * force one; a render call fails with PCR_E_ARG if the stream's layout does not hold it. Results are identical. */
#define PCR_VARIANT_AUTO 0
#define PCR_VARIANT_WORDS 1
#define PCR_VARIANT_POINT_WINDOWS 2
int pcr_set_render_variant(pcr_ctx *ctx, int variant);

/* Workgroups per batch of the render kernels. The reference launches one 1024-thread block per batch
 * (modules/huffman_mem_iter_cuda/render.cu:328, huffman_mem_iter_cuda.h: cuLaunchKernel grid = numBatches); this build can also
 * draw a batch with two workgroups of 512 threads (chains 0..511 and 512..1023, four workgroups per CU) -- same frames, finer
 * turnover of LDS and wave slots. parts: 0 = chosen by the library (default), 1 = whole batches, 2 = half-batches. Takes effect
 * with the next prepass (pcr_frame_begin / pcr_frame_turn / a render call). */
int pcr_set_workgroup_parts(pcr_ctx *ctx, int parts);
/* Device bytes the loaded stream occupies right now (every per-stream allocation of the context, pads and guards included;
 * framebuffers excluded). Drops when the first frame after the last upload releases what only the load-time transcode reads. */
int64_t pcr_stream_resident_bytes(const pcr_ctx *ctx);
/* Colour format of the loaded stream: PCR_COLOR_BC1, PCR_COLOR_BC7 (a file written by a reference built with
 * COLOR_COMPRESSION == 7, include/BatchDumpData.h:130-136: 16 colour bytes per 16 points; told by the size of the first
 * record uploaded), 0 before the first record. A BC7 stream is drawn by the HQS method only (pcr_render_hqs_color decodes
 * mode-6 blocks as huffman_hqs/render.cu:240-273 does); pcr_render_basic refuses it: the reference's basic method decodes
 * the array as BC1 whatever the setting (huffman_mem_iter_cuda/render.cu:299) and its resolve then indexes it with a colour
 * (resolve.cu:183), which is not a result to be identical to. */
int pcr_stream_color_format(const pcr_ctx *ctx);
/* Version tag of the render/transcode kernels in this library ("rNN.vMM"): stored measurements name the tag they belong to. */
const char *pcr_kernel_version(void);

/* pcr_clear and the cull/LOD prepass of the frame's first render call in ONE launch: equivalent to pcr_clear followed by
 * what pcr_render_basic (method PCR_METHOD_BASIC) or pcr_render_hqs_depth (PCR_METHOD_HQS) would do first. The render call
 * that follows skips its prepass if it is given the same parameters and the loaded batches have not changed; otherwise it
 * runs its own, so calling this is never wrong, only sometimes useless. The frame loop of the adapters uses it where the
 * reference clears (huffman_hqs.h:266-270 / huffman_mem_iter_cuda.h:250-252). */
#define PCR_METHOD_BASIC 0
#define PCR_METHOD_HQS 1
int pcr_frame_begin(pcr_ctx *ctx, const pcr_render_params *params, int method);
/* The reference's RESOLVE + CLEAR at the end of a frame (huffman_hqs.h:240-270) and pcr_frame_begin's prepass for the next
 * frame, in ONE launch and one pass over the framebuffer: pcr_resolve_basic / pcr_resolve_hqs with the flags of `done`, then
 * pcr_frame_begin(next, method). The image is where pcr_resolve_* leaves it (pcr_read_rgba); the u64 framebuffer is empty
 * afterwards, so read it first if it is wanted. A steady frame loop is then two launches: pcr_render_*, pcr_frame_turn. */
int pcr_frame_turn(pcr_ctx *ctx, const pcr_render_params *done, const pcr_render_params *next, int method);

/* Multi-GPU merges through a library that only has a SIGNED 64-bit MIN (RCCL as torch.distributed exposes it): with
 * on = 1, pcr_clear writes INT64_MAX (0x7FFF...F) into empty pixels instead of the reference's all-ones word. Every key a
 * point can produce has a clear top bit (the depth half is the bit pattern of a positive float), so the kernels' unsigned
 * atomicMin, the resolves (they test the low half against 0xFFFFFFFF) and the HQS depth test (both words read as NaN)
 * behave exactly as before, signed and unsigned order agree on the whole framebuffer, and no sign-flip passes are
 * needed around the collective. pcr_read_framebuffer reports empty pixels as all-ones in either mode. Takes effect with
 * the next pcr_clear. */
int pcr_set_int64_mergeable(pcr_ctx *ctx, int on);

/* Asynchronous loader. Off (default): pcr_upload_batches enqueues its copies and the transcode on the
 * context's stream, in order with the frames, and a frame draws every batch handed over so far, as the reference's
 * process() does (HuffmanLasLoader.cpp:301-313). On: they run on a loader stream of the context's own, the call
 * returns once the records are packed into a pinned arena, and a frame draws the batches whose loader task is known to
 * have completed (never the last arrived batch of an incomplete stream: its chains' tail over-reads reach into the
 * next batch's words), so frames do not wait for PCIe. pcr_batches_resident: how many batches the next frame draws at
 * least (= pcr_batches_loaded when the mode is off). Switching synchronises. */
int pcr_set_async_upload(pcr_ctx *ctx, int on);
int64_t pcr_batches_resident(pcr_ctx *ctx);
int64_t pcr_last_frame_batches(const pcr_ctx *ctx);   /* batches drawn by the last pcr_render_* call */

/* ---- measurement ---------------------------------------------------------------------------------
 * HIP events on the context's stream: begin/end bracket any sequence of enqueued calls;
 * pcr_timing_end synchronises and returns the elapsed milliseconds between the two events. */
int pcr_timing_begin(pcr_ctx *ctx);
int pcr_timing_end(pcr_ctx *ctx, float *elapsed_ms);
/* Practical HBM ceiling of this device, set beside the 8 TB/s spec peak in the roofline: best of `reps`
 * passes of a streaming read and of a streaming copy (read + write bytes counted) over temporary buffers of `bytes`
 * each (use >= 1 GiB: the Infinity Cache holds 256 MiB). GB/s = 1e9 bytes per second. */
int pcr_measure_hbm(pcr_ctx *ctx, size_t bytes, int reps, float *read_gbps, float *copy_gbps);

/* Per-launch duration of the dominant kernel: with every = n > 0, every n-th pcr_render_* call brackets its
 * decode+rasterize kernel (not the prepass) with a HIP event pair (device-scope release) on the stream it is launched on
 * (an event pair still costs stream time, hence the stride); every = 0 switches it off. pcr_kernel_timing_read synchronises and returns
 * the average over the most recent bracketed launches (at most 64) since enabling, and how many those were. */
int pcr_kernel_timing_enable(pcr_ctx *ctx, int every);
int pcr_kernel_timing_read(pcr_ctx *ctx, float *avg_ms, int *launches);

/* Algorithmic HBM bytes of one render launch over the loaded stream, SURVEY 8d's B_dec x points: every byte of the
 * compressed representation once (encoded + separate + cluster prefix + per batch 160 + 12 288 + 4 096 + 32 768). */
int64_t pcr_stream_algorithmic_bytes(const pcr_ctx *ctx);
/* The same for the frame the last render call drew (synchronises): the batches the cull kept, and of each batch's encoded +
 * escape words the share npr / 64 of the points per chain its level of detail decodes (a chain's prefix; proportional, the
 * exact prefix lengths are not recorded) + its side data whole. Equals pcr_stream_algorithmic_bytes for LOD 100 %, no culling. */
int64_t pcr_last_frame_algorithmic_bytes(pcr_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PCR_HIP_H */
