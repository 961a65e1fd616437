// pcr_api.hip — implementation of the C ABI in include/pcr_hip.h (libpcr_hip.so, gfx950 only).
//
// Host-side structure follows the reference's split into a Resource (HuffmanLasData: stream buffers,
// modules/compute/HuffmanLasLoader.cpp) and Methods (HuffmanMemIter / HuffmanHQS: framebuffers + launches,
// modules/huffman_mem_iter_cuda/huffman_mem_iter_cuda.h, modules/huffman_hqs/huffman_hqs.h), but one context
// owns both so a foreign-language caller needs a single handle. There is no CPU fallback: every entry point
// that needs the device fails with PCR_E_HIP / PCR_E_NODEVICE when HIP does.
#include "pcr_hip.h"
#include "pcr_kernels.hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

using namespace pcr;

constexpr int PCR_STATS_PARTIALS = PCR_MAX_PREPASS_WORKGROUPS;   // one partial record per prepass workgroup
#ifndef PCR_DEFAULT_PARTS
#define PCR_DEFAULT_PARTS 2                     /* workgroups per batch of k_render unless pcr_set_workgroup_parts says otherwise */
#endif
constexpr int64_t TRANSCODE_CHUNK = 128;        // batches per k_transcode launch when the lane-major words are scratch (40 MB)

struct pcr_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    // resource (HuffmanLasData)
    bool stream_open = false;
    pcr_file_header hdr{};
    int64_t batch_index_base = 0;
    int64_t batches_loaded = 0, points_loaded = 0;
    int64_t enc_ptr = 0, sep_ptr = 0;           // running word offsets (HuffmanLasLoader.h:50-53)
    int64_t enc_words = 0, sep_words = 0;       // allocation sizes in words incl. pads
    std::vector<int32_t> h_stream_words;        // per uploaded batch: encoded + escape words (pcr_last_frame_algorithmic_bytes)
    pcr_gpu_batch *d_batches = nullptr;
    int32_t *d_start = nullptr;
    uint32_t *d_encoded = nullptr;
    int32_t *d_separate = nullptr;
    int32_t *d_sep_sizes = nullptr;
    int32_t *d_table_values = nullptr;
    int8_t *d_table_lens = nullptr;
    int32_t *d_cluster_sizes = nullptr;
    uint8_t *d_colors = nullptr;                // as uploaded (k_transcode's input; released with the other raw arrays)
    uint8_t *d_colors_t = nullptr;              // segment-major copy k_render reads (StreamView::colors_t)
    size_t color_bytes = 0;                     // per batch: PCR_COLOR_BYTES_PER_BATCH (BC1) or ..._BC7, told by the first record; 0 = not known yet
    uint32_t *d_lod = nullptr;
    WinPlan *d_win = nullptr;                   // LDS framebuffer windows of the frame's batches (prepass): two plans of hdr.num_batches
                                                // entries, slot 0 for 8-byte pixels (basic / HQS depth), slot 1 for the colour pass's 20-byte pixels
    uint32_t *d_batch_runs = nullptr;           // runs of chains and their bounding boxes, RUN_WORDS per batch (k_bounds)
    uint32_t *d_batch_flags = nullptr;          // BF_* per batch (k_transcode)
    // dense lists of the batches a frame draws, compacted by k_lod_prepass per workgroup (see RenderArgs): d_order[2][order_stride],
    // d_chunk_count[WORK_CLASSES + 1][PCR_MAX_PREPASS_WORKGROUPS]
    DrawRec *d_order = nullptr;
    uint32_t *d_chunk_count = nullptr;
    uint32_t order_stride = 0;
    // "some batch of this stream was ever flagged BF_GENERIC_SLOW_PATH": set on the device by k_transcode, copied to a
    // pinned word behind every transcode; while the copy is in flight the answer is "maybe" and the checked kernel is launched
    uint32_t *d_any_generic = nullptr;
    uint32_t *h_any_generic = nullptr;
    hipEvent_t any_generic_ev = nullptr;
    bool any_generic_pending = false;
    uint32_t *d_packed_table = nullptr;         // k_render's table entries, 4096 per batch (k_transcode)
    uint32_t *d_lane_words = nullptr;           // k_transcode's scratch: lane-major copy of the word stream, LW_ROWS x 1024 per batch of a chunk
    // PCR_LAYOUT_WORDS / _BOTH: the compact copy k_render's packed-words variant reads (k_pack_words). One device allocation per
    // transcode chunk, sized from the rows the chunk's waves consumed (read back: loading such a stream synchronises per chunk)
    bool keep_words = false;
    std::vector<uint32_t *> lw_segments;
    const uint32_t **d_lw_block = nullptr;      // [nB] the batch's block inside its chunk's segment
    uint32_t *d_lw_wave_row = nullptr;          // [nB * (LWC_WAVES + 1)] first row of every wave inside the block, and the end
    uint32_t *d_lw_prov = nullptr;              // [LWC_WAVES * LW_ROWS * 64] uncompacted rows of the provisional last batch of a stream that is still loading
    uint32_t *d_wave_rows = nullptr;            // [TRANSCODE_CHUNK * LWC_WAVES] rows per wave of the chunk being transcoded (k_transcode)
    uint32_t *h_wave_rows = nullptr;            // pinned: the same on the host, then the staging of the two arrays above
    int64_t provisional_for = -1;               // batches_loaded when the provisional walk of an incomplete stream's last batch was last done
    uint8_t *d_point_windows = nullptr;         // PCR_LAYOUT_POINT_WINDOWS: 40-bit view per point (k_transcode), PW_BATCH_BYTES per batch
    int layout = PCR_LAYOUT_POINT_WINDOWS;      // of the stream being loaded (pcr_set_stream_layout, fixed at pcr_stream_begin)
    int next_layout = PCR_LAYOUT_POINT_WINDOWS;
    int64_t hbm_budget = 0;                     // PCR_LAYOUT_AUTO: streams whose point windows would take more than this are loaded as packed words
    int variant = PCR_VARIANT_AUTO;             // which k_render variant draws a stream that has both layouts resident
    int64_t transcoded = 0;                     // batches [0, transcoded) of d_lane_words / d_point_windows are final
    // PCR_LAYOUT_POINT_WINDOWS keeps no lane-major words: d_lane_words is then k_transcode's scratch for TRANSCODE_CHUNK
    // batches at a time (launches of a stream run in order, so consecutive launches may reuse it)
    bool lane_words_scratch = false;
    // set by the first frame after the last batch: the buffers only k_transcode reads (raw word stream, int32/int8 tables,
    // cluster prefix, the scratch) have been released; pcr_upload_tail is refused from then on
    bool finalized = false;
    size_t stream_bytes = 0;                    // device bytes the loaded stream occupies right now (pcr_stream_resident_bytes)
    pcr_render_stats *d_stats = nullptr;        // PCR_STATS_PARTIALS partial records, one per prepass workgroup
    int stats_partials = 0;                     // how many the last render launch wrote
    // pinned staging arenas of the loader (double-buffered)
    uint8_t *arena[2] = {nullptr, nullptr};
    size_t arena_size[2] = {0, 0};
    hipEvent_t arena_done[2] = {nullptr, nullptr};
    bool arena_busy[2] = {false, false};
    int arena_next = 0;
    // asynchronous loader (pcr_set_async_upload): copies + transcode on a stream of their own; a render launch draws the
    // batches whose loader task is known to have completed, so frames never wait for the PCIe copy
    bool async_upload = false;
    hipStream_t copy_stream = nullptr;
    struct LoaderTask { hipEvent_t done; int64_t resident_after; };
    std::deque<LoaderTask> loader_tasks;
    std::vector<hipEvent_t> loader_events;      // recycled
    int64_t batches_resident = 0;
    int64_t last_frame_batches = 0;
    uint64_t empty_key = ~0ull;                 // what pcr_clear writes (pcr_set_int64_mergeable)
    // pcr_frame_begin ran the prepass of the next render call already, for exactly these inputs
    bool prepass_ready = false;
    pcr_render_params prepass_params{};
    int prepass_variant_hqs = 0;
    unsigned prepass_slots = 0;                 // which of the two window plans that prepass wrote (bit 0: 8-byte pixels, bit 1: 20-byte)
    uint32_t prepass_dyn_lds = 0;
    int prepass_parts = 0;
    int parts_mode = 0;                         // pcr_set_workgroup_parts: 0 = chosen per frame, 1 = whole batches, 2 = half-batches
    bool big_lds_ready = false;                 // hipFuncSetAttribute done for the 140 KiB launches
    int64_t prepass_batches = 0;
    static constexpr int FENCES = 8;
    hipEvent_t fence[FENCES] = {};              // pcr_fence_record / pcr_fence_wait: device-scope ordering between streams
    int64_t visible_batches() const { return async_upload ? batches_resident : batches_loaded; }

    // resource of the 10-10-10 path (ComputeLasData)
    bool las_open = false;
    int64_t las_capacity = 0, las_loaded = 0;   // batches
    pcr_xyz_batch *d_xyzb = nullptr;
    uint32_t *d_xyz12 = nullptr, *d_xyz8 = nullptr, *d_xyz4 = nullptr, *d_point_rgba = nullptr;
    int32_t *d_las_level = nullptr;
    uint2 *d_las_win = nullptr;
    uint32_t *d_las_order = nullptr, *d_las_chunk_count = nullptr;      // LasArgs::order / chunk_count

    // method (framebuffers)
    int width = 0, height = 0;
    size_t fb_elems = 0;
    size_t fb_alloc = 0;                        // elements of each own buffer: fb_elems + PCR_FRAME_PAD_ELEMS (pad: fb all ones, RG/BA/rgba zero, written once)
    uint64_t *own_fb = nullptr, *own_rg = nullptr, *own_ba = nullptr;
    uint64_t *fb = nullptr, *rg = nullptr, *ba = nullptr;
    uint32_t *d_rgba = nullptr;
    bool accum_dirty = true;    // RG/BA hold something other than zeros (only the HQS colour pass writes them)
    // dirty tiles (FrameView::tiles): two arrays of one byte per 64 x 16 pixels; `tile_cur` is the one the frame being drawn marks,
    // the other says where the image may hold something. `tiles_tracked`: every framebuffer write since the last clear went
    // through a marking kernel (false after merges, external buffers, the 10-10-10 method: the next turn does the whole frame)
    uint8_t *d_tiles = nullptr;                 // three arrays of ntiles bytes, then three "everything" words
    uint32_t tiles_x = 0, ntiles = 0, tiles_stride = 0;
    int tile_cur = 0, tile_prev = 1, tile_spare = 2;    // roles of the three arrays: marked by the frame being drawn / image state / all zero
    uint32_t tile_e_cur = 1, tile_e_prev = 0, tile_epochs = 1;      // epoch that means "everything" in the cur / prev array's word
    bool tiles_tracked = false;
    // a raw pointer to one of the context's own buffers has been handed out (pcr_device_framebuffer / _rg / _ba): whoever holds it
    // may write anywhere at any time, so tiles are not used at all until pcr_framebuffer_private says the pointers are dead
    bool fb_exposed = false;
#ifdef PCR_EXP_NO_TILES     /* experiment: the whole frame is resolved and cleared, nothing marks tiles */
    bool tiles_usable() const { return false; }
#else
    bool tiles_usable() const { return d_tiles && !fb_exposed && fb == own_fb && rg == own_rg && ba == own_ba; }
#endif
    uint8_t *tiles_half(int which) const { return d_tiles + (size_t)which * tiles_stride; }
    uint32_t *tiles_all(int which) const { return reinterpret_cast<uint32_t *>(d_tiles + 3 * (size_t)tiles_stride) + which; }

    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    // per-launch timing of the dominant kernel (pcr_kernel_timing_*): event pairs around k_render / k_las_render only
    static constexpr int KT_PAIRS = 64;
    hipEvent_t kt_begin[KT_PAIRS] = {}, kt_end[KT_PAIRS] = {};
    int kt_every = 0;                           // 0 = off, n = bracket every n-th launch
    int64_t kt_launches = 0;                    // launches since enable
    int64_t kt_samples = 0;                     // bracketed launches since enable (only the last KT_PAIRS are kept)
    bool kt_sample_now() const { return kt_every > 0 && kt_launches % kt_every == 0; }
};

namespace {

thread_local std::string g_create_err;

int set_err(pcr_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

#define HIP_TRY(c, call)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return set_err((c), PCR_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <class T> void dfree(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

// Async loader: retire the tasks whose event has fired (or, with wait, all of them).
void poll_loader(pcr_ctx *c, bool wait)
{
    while (!c->loader_tasks.empty()) {
        pcr_ctx::LoaderTask &t = c->loader_tasks.front();
        if (wait) (void)hipEventSynchronize(t.done);
        else if (hipEventQuery(t.done) != hipSuccess) break;
        c->batches_resident = t.resident_after;
        c->loader_events.push_back(t.done);
        c->loader_tasks.pop_front();
    }
}

void free_stream_buffers(pcr_ctx *c)
{
    poll_loader(c, true);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    c->batches_resident = 0;
    dfree(c->d_batches); dfree(c->d_start); dfree(c->d_encoded); dfree(c->d_separate); dfree(c->d_sep_sizes);
    dfree(c->d_table_values); dfree(c->d_table_lens); dfree(c->d_cluster_sizes); dfree(c->d_colors); dfree(c->d_colors_t); dfree(c->d_lod); dfree(c->d_win);
    for (uint32_t *seg : c->lw_segments) (void)hipFree(seg);
    c->lw_segments.clear(); dfree(c->d_lw_block); dfree(c->d_lw_wave_row); dfree(c->d_wave_rows); dfree(c->d_lw_prov); c->provisional_for = -1;
    dfree(c->d_lane_words); dfree(c->d_batch_flags); dfree(c->d_packed_table); dfree(c->d_point_windows); dfree(c->d_batch_runs); c->transcoded = 0;
    dfree(c->d_order); dfree(c->d_chunk_count); dfree(c->d_any_generic); c->order_stride = 0;

    if (c->any_generic_pending && c->any_generic_ev) (void)hipEventSynchronize(c->any_generic_ev);
    c->any_generic_pending = false;
    if (c->h_any_generic) *c->h_any_generic = 0;
    c->stream_open = false; c->batches_loaded = c->points_loaded = 0; c->prepass_ready = false;
    c->finalized = false; c->lane_words_scratch = false; c->stream_bytes = 0; c->color_bytes = 0;
    c->enc_ptr = c->sep_ptr = 0; c->enc_words = c->sep_words = 0;
    c->h_stream_words.clear();
}

void free_las_buffers(pcr_ctx *c)
{
    dfree(c->d_xyzb); dfree(c->d_xyz12); dfree(c->d_xyz8); dfree(c->d_xyz4); dfree(c->d_point_rgba);
    dfree(c->d_las_level); dfree(c->d_las_win); dfree(c->d_las_order); dfree(c->d_las_chunk_count);
    c->las_open = false; c->las_capacity = c->las_loaded = 0;
}

void free_frame_buffers(pcr_ctx *c)
{
    dfree(c->own_fb); dfree(c->own_rg); dfree(c->own_ba); dfree(c->d_rgba); dfree(c->d_tiles);
    c->tiles_tracked = false; c->ntiles = c->tiles_x = 0;
    c->fb = c->rg = c->ba = nullptr; c->fb_elems = 0; c->fb_alloc = 0; c->width = c->height = 0;
}

template <class T> int dalloc_zero(pcr_ctx *c, T *&p, size_t count, size_t *account = nullptr)
{
    size_t bytes = (count ? count : 1) * sizeof(T);
    HIP_TRY(c, hipMalloc((void **)&p, bytes));
    HIP_TRY(c, hipMemsetAsync(p, 0, bytes, c->stream));
    if (account) *account += bytes;
    return PCR_OK;
}
template <class T> void dfree_counted(pcr_ctx *c, T *&p, size_t count)
{
    if (!p) return;
    c->stream_bytes -= (count ? count : 1) * sizeof(T);
    dfree(p);
}

int check_params(pcr_ctx *c, const pcr_render_params *p)
{
    if (!c) return PCR_E_ARG;
    if (!p) return set_err(c, PCR_E_ARG, "render params are NULL");
    if (!c->stream_open) return set_err(c, PCR_E_ARG, "no stream loaded (call pcr_stream_begin / pcr_upload_batch)");
    if (!c->fb) return set_err(c, PCR_E_ARG, "no framebuffer (call pcr_set_image_size)");
    if (p->width != c->width || p->height != c->height)
        return set_err(c, PCR_E_ARG, "params image size %dx%d != framebuffer %dx%d", p->width, p->height, c->width, c->height);
    if (p->points_per_thread != PCR_POINTS_PER_THREAD)
        return set_err(c, PCR_E_ARG, "points_per_thread must be %d", PCR_POINTS_PER_THREAD);
    if (p->lod_percent < 0) return set_err(c, PCR_E_ARG, "lod_percent must be >= 0");
    return PCR_OK;
}

// Workgroups per batch of a frame's k_render launches (RenderArgs::parts): half-batch workgroups unless told otherwise
// (pcr_set_workgroup_parts; PCR_PARTS in the environment for experiments).
int frame_parts(const pcr_ctx *c)
{
    static const char *force = getenv("PCR_PARTS");
    if (force && (force[0] == '1' || force[0] == '2')) return force[0] - '0';
    return c->parts_mode ? c->parts_mode : PCR_DEFAULT_PARTS;
}

StreamView make_stream_view(pcr_ctx *c)
{
    StreamView s;
    s.batches = c->d_batches; s.start_values = c->d_start; s.encoded = c->d_encoded;
    s.separate = c->d_separate; s.separate_sizes = c->d_sep_sizes; s.table_values = c->d_table_values;
    s.table_lens = c->d_table_lens; s.cluster_sizes = c->d_cluster_sizes; s.colors = c->d_colors; s.colors_t = c->d_colors_t; s.color_block_bytes = c->color_bytes == PCR_COLOR_BYTES_PER_BATCH_BC7 ? 16u : 8u;
    s.lane_words = c->d_lane_words; s.batch_flags = c->d_batch_flags; s.packed_table = c->d_packed_table;
    s.lw_block = c->d_lw_block; s.lw_wave_row = c->d_lw_wave_row;
    s.point_windows = c->d_point_windows; s.batch_runs = c->d_batch_runs;
    s.encoded_words = c->enc_words; s.separate_words = c->sep_words;
    s.num_batches = c->visible_batches(); s.batch_index_base = c->batch_index_base;
    return s;
}

RenderArgs make_args(pcr_ctx *c, const pcr_render_params *p, int variant_hqs)
{
    RenderArgs a;
    a.p = *p;
    a.s = make_stream_view(c);
    a.f.fb = c->fb; a.f.rg = c->rg; a.f.ba = c->ba; a.f.fb_elems = (uint32_t)c->fb_elems;
    a.f.tiles = c->tiles_usable() && c->tiles_tracked ? c->tiles_half(c->tile_cur) : nullptr;
    a.f.tiles_all = c->d_tiles ? c->tiles_all(c->tile_cur) : nullptr;
    a.f.tiles_x = c->tiles_x; a.f.tiles_epoch = c->tile_e_cur; a.f.tiles_total = c->ntiles;
    a.lod = c->d_lod; a.win = c->d_win; a.win_hqs = nullptr; a.stats = c->d_stats; a.variant_hqs = variant_hqs;
    a.parts = frame_parts(c);
    a.order = c->d_order;
    a.chunk_count = c->d_chunk_count;
    // chunks of the batches this frame draws (the prepass of the frame covered exactly these)
    a.order_stride = (uint32_t)((a.s.num_batches + PREPASS_BATCHES - 1) / PREPASS_BATCHES) * PREPASS_BATCHES;
    a.work_classes = a.s.num_batches > 8192 ? 1u : (uint32_t)WORK_CLASSES;
    return a;
}

// Does the stream hold a batch for the checked kernel (of workgroups of 1024 / parts threads)? false only when the device has said so.
bool maybe_generic_batches(pcr_ctx *c, int parts)
{
    if (c->any_generic_pending && hipEventQuery(c->any_generic_ev) == hipSuccess) c->any_generic_pending = false;
    return c->any_generic_pending || (*c->h_any_generic & bf_generic(parts)) != 0;
}

// Lane-major word sequences + packed tables (k_transcode) for every loaded batch that does not have them yet. A batch is
// final once the batch behind it is loaded (its chains' tail over-reads, SURVEY B.4, reach into those words) or the
// stream is complete; the last batch of an incomplete stream is walked provisionally (`include_provisional`, render
// time only) and walked again when more data arrives. Part of loading: pcr_upload_batches calls this for what it can.
int enqueue_transcode(pcr_ctx *c, bool include_provisional, hipStream_t st)
{
    const int64_t loaded = c->batches_loaded;
    const int64_t final_end = loaded == c->hdr.num_batches ? loaded : loaded - 1;
    const int64_t end = include_provisional ? loaded : final_end;
    // (the provisional walk of the last batch is the same until more data arrives: once per arrival, not once per frame)
    if (end > c->transcoded && c->transcoded == final_end && c->provisional_for == loaded) return PCR_OK;
    if (end > c->transcoded) {
        for (int64_t b0 = c->transcoded; b0 < end; ) {
            // (the provisional batch is a launch of its own: with the packed-words layout its rows go to a fixed scratch block)
            const int64_t n = b0 < final_end ? std::min(TRANSCODE_CHUNK, final_end - b0) : 1;
            const bool provisional = b0 >= final_end;
            hipLaunchKernelGGL(k_transcode, dim3((unsigned)n), dim3(PCR_WORKGROUP_SIZE), 0, st,
                               make_stream_view(c), c->d_lane_words, c->d_batch_flags, c->d_packed_table, c->d_point_windows, c->d_colors_t,
                               c->d_any_generic, (int)b0, (int)b0, c->keep_words ? c->d_wave_rows : nullptr);
            hipLaunchKernelGGL(k_bounds, dim3((unsigned)n), dim3(PCR_WORKGROUP_SIZE), 0, st,
                               make_stream_view(c), c->d_lane_words, c->d_batch_runs, (int)b0, (int)b0);
            if (c->keep_words && provisional) {
                // The last batch of a stream that is still loading, walked in front of a render call: no read-back, no allocation,
                // no host synchronisation (a segment per provisional walk leaked ~190 KB per upload until the stream
                // was unloaded, and the render call blocked). Its rows are copied uncompacted -- LW_ROWS per wave -- into one
                // scratch block that every provisional walk reuses; the batch is packed into its chunk's segment once it is final.
                hipLaunchKernelGGL(k_provisional_block, dim3(1), dim3(64), 0, st, (const uint32_t **)c->d_lw_block, c->d_lw_wave_row, c->d_lw_prov, (int)b0);
                hipLaunchKernelGGL(k_pack_words, dim3(1), dim3(PCR_WORKGROUP_SIZE), 0, st, c->d_lane_words, (uint32_t *const *)c->d_lw_block,
                                   c->d_lw_wave_row, (int)b0, (int)b0);
            } else if (c->keep_words) {
                // the compact copy: how many rows did every wave of the chunk consume? (the one place loading waits for the device)
                uint32_t *h = c->h_wave_rows;
                HIP_TRY(c, hipMemcpyAsync(h, c->d_wave_rows, (size_t)n * LWC_WAVES * 4, hipMemcpyDeviceToHost, st));
                HIP_TRY(c, hipStreamSynchronize(st));
                const uint32_t **h_block = reinterpret_cast<const uint32_t **>(h + TRANSCODE_CHUNK * LWC_WAVES);
                uint32_t *h_rows = h + TRANSCODE_CHUNK * LWC_WAVES + TRANSCODE_CHUNK * 2;
                std::vector<size_t> first_row((size_t)n);
                size_t total_rows = 0;
                for (int64_t i = 0; i < n; ++i) {
                    first_row[(size_t)i] = total_rows;
                    uint32_t r = 0;
                    for (int w = 0; w < LWC_WAVES; ++w) {
                        h_rows[i * (LWC_WAVES + 1) + w] = r;
                        r += std::min<uint32_t>(std::max<uint32_t>(h[i * LWC_WAVES + w], 2u), (uint32_t)LW_ROWS);
                    }
                    h_rows[i * (LWC_WAVES + 1) + LWC_WAVES] = r;
                    total_rows += r;
                }
                uint32_t *seg = nullptr;
                total_rows += LWC_PAD_ROWS;                 // (requests run up to four rows past a wave's last row: LWC_PAD_ROWS)
                if (hipMalloc((void **)&seg, total_rows * LWC_ROW_BYTES) != hipSuccess)
                    return set_err(c, PCR_E_NOMEM, "out of device memory for %zu bytes of packed words", total_rows * (size_t)LWC_ROW_BYTES);
                c->lw_segments.push_back(seg);
                c->stream_bytes += total_rows * LWC_ROW_BYTES;
                for (int64_t i = 0; i < n; ++i) h_block[i] = seg + first_row[(size_t)i] * 64;
                HIP_TRY(c, hipMemcpyAsync(c->d_lw_block + b0, h_block, (size_t)n * sizeof(uint32_t *), hipMemcpyHostToDevice, st));
                HIP_TRY(c, hipMemcpyAsync(c->d_lw_wave_row + b0 * (LWC_WAVES + 1), h_rows, (size_t)n * (LWC_WAVES + 1) * 4, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_pack_words, dim3((unsigned)n), dim3(PCR_WORKGROUP_SIZE), 0, st, c->d_lane_words, (uint32_t *const *)c->d_lw_block,
                                   c->d_lw_wave_row, (int)b0, (int)b0);
                HIP_TRY(c, hipStreamSynchronize(st));       // (the pinned staging is reused by the next chunk)
            }
            b0 += n;
        }
        c->transcoded = std::max(c->transcoded, final_end);
        c->provisional_for = end > final_end ? loaded : -1;
        // the sticky "some batch needs the checked kernel" word follows every transcode to the host
        (void)hipMemcpyAsync(c->h_any_generic, c->d_any_generic, 4, hipMemcpyDeviceToHost, st);
        (void)hipEventRecord(c->any_generic_ev, st);
        c->any_generic_pending = true;
    }
    return PCR_OK;
}

// Dynamic LDS of a frame's k_render launches: the small configuration (two workgroups per CU) unless the image has more
// pixels than the loaded batches' small windows hold together -- then most points would leave the windows and the frame is
// bound by global pre-reads and atomics; one workgroup per CU with windows of ~17 000 pixels keeps them in LDS.
uint32_t frame_dyn_lds(const pcr_ctx *c, int64_t nB)
{
    static const char *force = getenv("PCR_EXP_DYN_LDS");              // experiments: "small" / "big"
    const bool halves = frame_parts(c) == 2;
    const uint32_t small = halves ? (uint32_t)DYN_LDS_BYTES_HALF : (uint32_t)DYN_LDS_BYTES, big = halves ? (uint32_t)DYN_LDS_BYTES_HALF_BIG : (uint32_t)DYN_LDS_BYTES_BIG;
    if (force && force[0] == 's') return small;
    if (force && force[0] == 'b') return big;
    if (force && force[0] >= '1' && force[0] <= '9') return (uint32_t)atoi(force) * 1024u;     // KiB (must not exceed `big`)
    return (int64_t)c->width * c->height > nB * (int64_t)WIN_PIXELS ? big : small;
}

template <int MODE, int LAYOUT, bool GENERIC> hipError_t allow_big_lds()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_render<MODE, LAYOUT, GENERIC, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DYN_LDS_BYTES_BIG);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_render<MODE, LAYOUT, GENERIC, 2>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, DYN_LDS_BYTES_HALF_BIG);
    return e;
}
int enable_big_lds(pcr_ctx *c)
{
    if (c->big_lds_ready) return PCR_OK;
    hipError_t e = hipSuccess;
#define PCR_ALLOW(M, L) if (e == hipSuccess) e = allow_big_lds<M, L, false>(); if (e == hipSuccess) e = allow_big_lds<M, L, true>()
    PCR_ALLOW(MODE_BASIC, LAYOUT_WORDS); PCR_ALLOW(MODE_BASIC, LAYOUT_POINT_WINDOWS);
    PCR_ALLOW(MODE_HQS_DEPTH, LAYOUT_WORDS); PCR_ALLOW(MODE_HQS_DEPTH, LAYOUT_POINT_WINDOWS);
    PCR_ALLOW(MODE_HQS_COLOR, LAYOUT_WORDS); PCR_ALLOW(MODE_HQS_COLOR, LAYOUT_POINT_WINDOWS);
    PCR_ALLOW(MODE_HQS_COLOR_BC7, LAYOUT_WORDS); PCR_ALLOW(MODE_HQS_COLOR_BC7, LAYOUT_POINT_WINDOWS);
#undef PCR_ALLOW
    if (e != hipSuccess) return set_err(c, PCR_E_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(e));
    c->big_lds_ready = true;
    return PCR_OK;
}

// Once the whole stream is loaded and every batch has its final transcode, what only k_transcode reads is dead weight:
// the raw cluster-interleaved words (the largest array of the file), the int32 / int8 decoder tables (k_render reads the
// packed entries; the int32 values stay if some batch has `wide` entries), the cluster prefix and the transcode scratch.
// Called from the first frame after completion; hipFree waits for the device, once.
void maybe_finalize(pcr_ctx *c)
{
    if (c->finalized || c->batches_loaded != c->hdr.num_batches || c->transcoded != c->batches_loaded) return;
    if (c->async_upload && (!c->loader_tasks.empty() || c->batches_resident != c->batches_loaded)) return;   // the loader stream is still at it
    if (c->any_generic_pending) { (void)hipEventSynchronize(c->any_generic_ev); c->any_generic_pending = false; }
    const size_t nB = (size_t)c->hdr.num_batches;
    dfree_counted(c, c->d_encoded, (size_t)c->enc_words + PCR_GUARD_WORDS);
    dfree_counted(c, c->d_table_lens, nB * 4096);
    dfree_counted(c, c->d_cluster_sizes, nB * 32);
    dfree_counted(c, c->d_colors, nB * (c->color_bytes ? c->color_bytes : (size_t)PCR_COLOR_BYTES_PER_BATCH));     // k_render reads colors_t
    if (*c->h_any_generic == 0) dfree_counted(c, c->d_table_values, nB * 4096);
    dfree_counted(c, c->d_lane_words, (size_t)std::min<int64_t>(TRANSCODE_CHUNK, (int64_t)nB) * LW_ROWS * PCR_WORKGROUP_SIZE);
    if (c->d_wave_rows) dfree_counted(c, c->d_wave_rows, (size_t)TRANSCODE_CHUNK * LWC_WAVES);
    if (c->d_lw_prov) dfree_counted(c, c->d_lw_prov, (size_t)(LWC_WAVES * LW_ROWS + LWC_PAD_ROWS) * 64);
    c->finalized = true;
}

template <int MODE> int launch_render(pcr_ctx *c, const pcr_render_params *p)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    if (c->async_upload) poll_loader(c, false);
    const int64_t nB = c->visible_batches();         // "don't execute a workgroup until all points inside are loaded"
    c->last_frame_batches = nB;
    if (nB == 0) { c->stats_partials = 0; return PCR_OK; }   // huffman_hqs.h:137
    if (!c->async_upload && (rc = enqueue_transcode(c, true, c->stream))) return rc;   // normally only the provisional last batch of a stream that is still loading
    maybe_finalize(c);
    constexpr bool color_pass = MODE == MODE_HQS_COLOR || MODE == MODE_HQS_COLOR_BC7;
    const int win_pixel_bytes = color_pass ? WIN_PIXEL_BYTES_HQS : WIN_PIXEL_BYTES;
    const unsigned my_slot = color_pass ? 2u : 1u;
    const int variant_hqs = MODE != MODE_BASIC;
    const uint32_t dyn_lds = frame_dyn_lds(c, nB);
    const int parts = frame_parts(c);
    const bool have_prepass = c->prepass_ready && c->prepass_batches == nB && c->prepass_variant_hqs == variant_hqs &&
                              (c->prepass_slots & my_slot) && c->prepass_dyn_lds == dyn_lds && c->prepass_parts == parts &&
                              std::memcmp(&c->prepass_params, p, sizeof *p) == 0;
    c->prepass_ready = false;
    RenderArgs a = make_args(c, p, variant_hqs);
    WinPlan *const plan_slot[2] = { c->d_win, c->d_win + c->hdr.num_batches * MAX_PARTS };
    a.win = plan_slot[color_pass ? 1 : 0];
    a.win_pixel_bytes = win_pixel_bytes;
    a.dyn_lds_bytes = dyn_lds;
    if (dyn_lds > (uint32_t)(parts == 2 ? DYN_LDS_BYTES_HALF : DYN_LDS_BYTES) && (rc = enable_big_lds(c))) return rc;
    unsigned slots = c->prepass_slots;
    if (!have_prepass) {
        // the depth pass's prepass writes the colour pass's window plan as well: cull, LOD and the batch lists are the same
        // for both passes of a frame (huffman_hqs/depth.cu:197-227 == render.cu:362-392), only the LDS pixel size differs
        if (MODE == MODE_HQS_DEPTH) a.win_hqs = plan_slot[1];
        slots = MODE == MODE_HQS_DEPTH ? 3u : my_slot;
        c->stats_partials = (int)((nB + PREPASS_BATCHES - 1) / PREPASS_BATCHES);
        hipLaunchKernelGGL(k_lod_prepass, dim3((unsigned)c->stats_partials * PREPASS_WGS_PER_CHUNK), dim3(PREPASS_THREADS), 0, c->stream, a);
        a.win_hqs = nullptr;
    }
    if (MODE == MODE_HQS_DEPTH && (slots & 2u)) {
        // what the colour pass of this frame needs is in place: it runs no prepass of its own if it is given the same parameters
        c->prepass_ready = true; c->prepass_slots = 2u;
        c->prepass_params = *p; c->prepass_variant_hqs = variant_hqs; c->prepass_dyn_lds = dyn_lds; c->prepass_batches = nB;
        c->prepass_parts = parts;
    }
    const bool timed = c->kt_sample_now();
    const int slot = (int)(c->kt_samples % pcr_ctx::KT_PAIRS);
    if (timed) HIP_TRY(c, hipEventRecord(c->kt_begin[slot], c->stream));
    // A stream loaded with PCR_LAYOUT_BOTH can be drawn by either variant. The windows variant trades bytes for
    // instructions, which pays while the scatter runs in the LDS framebuffer windows; when the batches' rectangles outgrow
    // those (more pixels per batch than a window holds: 4096x4096 over 1526 batches), the frame is bound by global
    // framebuffer traffic and the 3 B per point of the packed words win.
    const bool have_windows = c->layout != PCR_LAYOUT_WORDS, have_words = c->layout != PCR_LAYOUT_POINT_WINDOWS;
    if ((c->variant == PCR_VARIANT_WORDS && !have_words) || (c->variant == PCR_VARIANT_POINT_WINDOWS && !have_windows))
        return set_err(c, PCR_E_ARG, "render variant %d needs a stream loaded with that layout (or PCR_LAYOUT_BOTH); this one has layout %d",
                       c->variant, c->layout);
    const bool windows = have_windows && (c->variant == PCR_VARIANT_POINT_WINDOWS || !have_words ||
                                          (c->variant == PCR_VARIANT_AUTO && (int64_t)c->width * c->height <= nB * (int64_t)WIN_PIXELS));
    // The grid is sized for "every batch visible"; workgroups beyond the prepass's dense list return at once. The checked
    // kernel (second list) is launched only if the stream may hold a flagged batch.
    const bool generic = maybe_generic_batches(c, parts);
    // (half-batches: workgroup x draws part (x >> 3) & 1 of list entry (x >> 4) * 8 + (x & 7) -- both halves on one XCD, see k_render)
    const dim3 grid(parts == 2 ? (unsigned)((nB + 7) / 8) * 16u : (unsigned)nB), block((unsigned)(PCR_WORKGROUP_SIZE / parts));
#define PCR_LAUNCH(L, G) do { if (parts == 2) hipLaunchKernelGGL((k_render<MODE, L, G, 2>), grid, block, dyn_lds, c->stream, a); \
                              else hipLaunchKernelGGL((k_render<MODE, L, G, 1>), grid, block, dyn_lds, c->stream, a); } while (0)
    if (windows) {
        PCR_LAUNCH(LAYOUT_POINT_WINDOWS, false);
        if (generic) PCR_LAUNCH(LAYOUT_POINT_WINDOWS, true);
    } else {
        PCR_LAUNCH(LAYOUT_WORDS, false);
        if (generic) PCR_LAUNCH(LAYOUT_WORDS, true);
    }
#undef PCR_LAUNCH
    if (timed) { HIP_TRY(c, hipEventRecord(c->kt_end[slot], c->stream)); ++c->kt_samples; }
    if (c->kt_every > 0) ++c->kt_launches;
    HIP_TRY(c, hipGetLastError());
    return PCR_OK;
}

} // namespace

extern "C" {

const char *pcr_last_error(const pcr_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

// Bumped with every change to k_render / k_transcode that can move a measured number: what a stored profile (HBM traffic
// from PMC counters, profiles/pmc_traffic_latest.json) was measured on is compared with this before it is quoted.
const char *pcr_kernel_version(void) { return "r04.v111"; }

int pcr_create(int device, pcr_ctx **out)
{
    if (!out) return set_err(nullptr, PCR_E_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return set_err(nullptr, PCR_E_NODEVICE, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return set_err(nullptr, PCR_E_ARG, "device %d out of range [0,%d)", device, ndev);
    HIP_TRY(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(nullptr, PCR_E_NODEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    pcr_ctx *c = new (std::nothrow) pcr_ctx();
    if (!c) return set_err(nullptr, PCR_E_NOMEM, "out of host memory");
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev_begin) != hipSuccess || hipEventCreate(&c->ev_end) != hipSuccess ||
        hipEventCreateWithFlags(&c->arena_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->arena_done[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->any_generic_ev, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void **)&c->h_any_generic, 64, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&c->d_stats, PCR_STATS_PARTIALS * sizeof(pcr_render_stats)) != hipSuccess) {
        pcr_destroy(c);
        return set_err(nullptr, PCR_E_HIP, "could not create stream/events");
    }
    *c->h_any_generic = 0;
    c->stream = c->own_stream;
    *out = c;
    return PCR_OK;
}

void pcr_destroy(pcr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_stream_buffers(c);
    free_las_buffers(c);
    free_frame_buffers(c);
    dfree(c->d_stats);
    for (int i = 0; i < 2; ++i) {
        if (c->arena[i]) (void)hipHostFree(c->arena[i]);
        if (c->arena_done[i]) (void)hipEventDestroy(c->arena_done[i]);
    }
    for (int i = 0; i < pcr_ctx::KT_PAIRS; ++i) {
        if (c->kt_begin[i]) (void)hipEventDestroy(c->kt_begin[i]);
        if (c->kt_end[i]) (void)hipEventDestroy(c->kt_end[i]);
    }
    if (c->any_generic_ev) (void)hipEventDestroy(c->any_generic_ev);
    if (c->h_any_generic) (void)hipHostFree(c->h_any_generic);
    if (c->h_wave_rows) (void)hipHostFree(c->h_wave_rows);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    for (hipEvent_t e : c->loader_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->fence) if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int pcr_set_stream(pcr_ctx *c, void *hip_stream)
{
    if (!c) return PCR_E_ARG;
    // No synchronisation here: work already enqueued stays on the stream it was enqueued on, and ordering between
    // the old and the new stream is the caller's (events), which is what lets a merge overlap the next render.
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return PCR_OK;
}

int pcr_synchronize(pcr_ctx *c)
{
    if (!c) return PCR_E_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PCR_OK;
}

// ---- resource ----------------------------------------------------------------------------------
int pcr_stream_begin(pcr_ctx *c, const pcr_file_header *h, int64_t batch_index_base)
{
    if (!c) return PCR_E_ARG;
    if (!h) return set_err(c, PCR_E_ARG, "header is NULL");
    if (h->num_batches <= 0 || h->num_points != h->num_batches * PCR_POINTS_PER_BATCH)
        return set_err(c, PCR_E_FORMAT, "header: numPoints %lld != numBatches %lld * 65536", (long long)h->num_points, (long long)h->num_batches);
    if (h->encoded_bytes < 0 || h->separate_bytes < 0 || (h->encoded_bytes & 3) || (h->separate_bytes & 3))
        return set_err(c, PCR_E_FORMAT, "header: bad stream byte counts");
    if (h->num_batches > 0xFFFF)
        return set_err(c, PCR_E_FORMAT, "at most 65535 batches (4.29e9 points) per context; shard larger files");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_stream_buffers(c);
    c->hdr = *h; c->batch_index_base = batch_index_base;
    const size_t nB = (size_t)h->num_batches;
    c->enc_words = h->encoded_bytes / 4 + PCR_ENCODED_PAD_WORDS;      // HuffmanLasLoader.cpp:39-41
    c->sep_words = h->separate_bytes / 4 + PCR_SEPARATE_PAD_WORDS;
    int rc;
    size_t *acc = &c->stream_bytes;
    int layout = c->next_layout;
    if (layout == PCR_LAYOUT_AUTO) {
        // what the stream would take as point windows (5 B per point + side data), against the budget the caller set
        const int64_t as_windows = (int64_t)nB * (int64_t)(PW_BATCH_BYTES + 4 * 4096 + 12288 + 4096 + PCR_COLOR_BYTES_PER_BATCH) + h->separate_bytes;
        layout = c->hbm_budget > 0 && as_windows > c->hbm_budget ? PCR_LAYOUT_WORDS : PCR_LAYOUT_POINT_WINDOWS;
    }
    const bool windows = layout != PCR_LAYOUT_WORDS, words = layout != PCR_LAYOUT_POINT_WINDOWS;
    c->lane_words_scratch = true;
    c->keep_words = words;
    const size_t lw_batches = (size_t)std::min<int64_t>(TRANSCODE_CHUNK, (int64_t)nB);
    if (words && !c->h_wave_rows)       // rows per wave of a chunk, then the staging of its block pointers and row offsets
        HIP_TRY(c, hipHostMalloc((void **)&c->h_wave_rows, (size_t)TRANSCODE_CHUNK * (LWC_WAVES + 2 + LWC_WAVES + 1) * 4, hipHostMallocDefault));
    if ((rc = dalloc_zero(c, c->d_batches, nB, acc)) || (rc = dalloc_zero(c, c->d_start, nB * 3072, acc)) ||
        (rc = dalloc_zero(c, c->d_encoded, (size_t)c->enc_words + PCR_GUARD_WORDS, acc)) ||
        (rc = dalloc_zero(c, c->d_separate, (size_t)c->sep_words + PCR_GUARD_WORDS, acc)) ||
        (rc = dalloc_zero(c, c->d_sep_sizes, nB * 1024, acc)) || (rc = dalloc_zero(c, c->d_table_values, nB * 4096, acc)) ||
        (rc = dalloc_zero(c, c->d_table_lens, nB * 4096, acc)) || (rc = dalloc_zero(c, c->d_cluster_sizes, nB * 32, acc)) ||
        (rc = dalloc_zero(c, c->d_colors, nB * PCR_COLOR_BYTES_PER_BATCH, acc)) || (rc = dalloc_zero(c, c->d_colors_t, nB * PCR_COLOR_BYTES_PER_BATCH, acc)) ||
        (rc = dalloc_zero(c, c->d_lod, nB, acc)) ||
        (rc = dalloc_zero(c, c->d_win, 2 * MAX_PARTS * nB, acc)) ||
        (rc = dalloc_zero(c, c->d_lane_words, lw_batches * LW_ROWS * PCR_WORKGROUP_SIZE, acc)) || (rc = dalloc_zero(c, c->d_batch_flags, nB, acc)) ||
        (rc = dalloc_zero(c, c->d_packed_table, nB * PCR_HUFFMAN_TABLE_SIZE, acc)) ||
        (rc = dalloc_zero(c, c->d_batch_runs, nB * RUN_RECORDS * RUN_WORDS, acc)) ||
        (rc = dalloc_zero(c, c->d_order, 2 * ((nB + PREPASS_BATCHES - 1) / PREPASS_BATCHES) * PREPASS_BATCHES, acc)) ||
        (rc = dalloc_zero(c, c->d_chunk_count, (WORK_CLASSES + 1) * PCR_MAX_PREPASS_WORKGROUPS, acc)) || (rc = dalloc_zero(c, c->d_any_generic, 1, acc)) ||
        (windows && (rc = dalloc_zero(c, c->d_point_windows, nB * PW_BATCH_BYTES + PW_GUARD_BYTES, acc))) ||
        (words && ((rc = dalloc_zero(c, c->d_lw_block, nB, acc)) || (rc = dalloc_zero(c, c->d_lw_wave_row, nB * (LWC_WAVES + 1), acc)) ||
                   (rc = dalloc_zero(c, c->d_wave_rows, (size_t)TRANSCODE_CHUNK * LWC_WAVES, acc)) ||
                   (rc = dalloc_zero(c, c->d_lw_prov, (size_t)(LWC_WAVES * LW_ROWS + LWC_PAD_ROWS) * 64, acc))))) {
        free_stream_buffers(c);
        return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->layout = layout;
    c->stream_open = true;
    return PCR_OK;
}

int pcr_upload_batch(pcr_ctx *c, int64_t batch_index, const void *blob, size_t n)
{
    return pcr_upload_batches(c, batch_index, 1, &blob, &n);
}

// One loader task (the reference hands over <= 100 records at a time, HuffmanLasLoader.cpp:106): every record is
// validated, the nine per-batch arrays of the whole task are packed contiguously into one of two pinned staging
// arenas, and nine large asynchronous copies move them; the arenas alternate, so packing task k+1 on the host
// overlaps the copies of task k. The reference issues nine synchronous cuMemcpyHtoD per record instead
// (HuffmanLasLoader.cpp:229-289).
int pcr_upload_batches(pcr_ctx *c, int64_t first_index, int64_t count, const void *const *blobs, const size_t *sizes)
{
    if (!c) return PCR_E_ARG;
    if (!c->stream_open) return set_err(c, PCR_E_ARG, "pcr_stream_begin has not been called");
    if (count <= 0 || !blobs || !sizes) return set_err(c, PCR_E_ARG, "no batch records given");
    if (first_index != c->batches_loaded)
        return set_err(c, PCR_E_ARG, "batches must be uploaded in order: expected %lld, got %lld", (long long)c->batches_loaded, (long long)first_index);
    if (first_index + count > c->hdr.num_batches) return set_err(c, PCR_E_ARG, "batch index %lld beyond header", (long long)(first_index + count - 1));

    const size_t fixed = PCR_BATCH_FIXED_HEADER + 4u * (3072 + 1024 + 4096 + 4096 + 32);
    struct View { const uint8_t *start, *sepsz, *tv, *tl, *cl, *enc, *sep, *col; int32_t ne, ns; };
    std::vector<View> views((size_t)count);
    int64_t sum_ne = 0, sum_ns = 0;
    size_t color_bytes = c->color_bytes;
    // ---- pass 1: validate (nothing is modified if any record is bad) --------------------------------------------
    for (int64_t k = 0; k < count; ++k) {
        const int64_t bi = first_index + k;
        const uint8_t *r = (const uint8_t *)blobs[k];
        const size_t n = sizes[k];
        if (!r) return set_err(c, PCR_E_ARG, "blob is NULL");
        if (n < fixed) return set_err(c, PCR_E_FORMAT, "batch %lld: record of %zu bytes is too short", (long long)bi, n);
        int32_t hdr[5]; std::memcpy(hdr, r, 20);                      // include/BatchDumpData.h:60-107
        int32_t dt_size, num_clusters;
        std::memcpy(&dt_size, r + 116, 4); std::memcpy(&num_clusters, r + 120, 4);
        if (hdr[1] != PCR_POINTS_PER_BATCH || hdr[2] != PCR_WORKGROUP_SIZE || hdr[3] != PCR_POINTS_PER_THREAD ||
            hdr[4] != PCR_CLUSTERS_PER_THREAD || dt_size != PCR_HUFFMAN_TABLE_SIZE || num_clusters != PCR_CLUSTERS_PER_BATCH)
            return set_err(c, PCR_E_FORMAT, "batch %lld: unsupported geometry (points %d threads %d ppt %d cpt %d table %d clusters %d)",
                           (long long)bi, hdr[1], hdr[2], hdr[3], hdr[4], dt_size, num_clusters);
        View &v = views[(size_t)k];
        v.start = r + PCR_BATCH_FIXED_HEADER;
        v.sepsz = v.start + 3072 * 4;
        v.tv = v.sepsz + 1024 * 4;
        v.tl = v.tv + 4096 * 4;
        v.cl = v.tl + 4096 * 4;
        v.enc = v.cl + 32 * 4;
        std::memcpy(&v.ne, v.cl + 31 * 4, 4);
        std::memcpy(&v.ns, v.sepsz + 1023 * 4, 4);
        if (v.ne < 64 || v.ns < 0) return set_err(c, PCR_E_FORMAT, "batch %lld: stream lengths %d / %d", (long long)bi, v.ne, v.ns);
        // BatchDumpData.h:130-136: 8 (BC1) or 16 (BC7 mode 6) colour bytes per 16 points, fixed when the reference is built
        // (COLOR_COMPRESSION); here the first record of a stream tells which, and the others have to agree
        const size_t streams = fixed + 4u * ((size_t)v.ne + (size_t)v.ns);
        if (color_bytes == 0) {
            if (n == streams + PCR_COLOR_BYTES_PER_BATCH) color_bytes = PCR_COLOR_BYTES_PER_BATCH;
            else if (n == streams + PCR_COLOR_BYTES_PER_BATCH_BC7) color_bytes = PCR_COLOR_BYTES_PER_BATCH_BC7;
        }
        if (color_bytes == 0 || n != streams + color_bytes)                                                          // BatchDumpData.h:148
            return set_err(c, PCR_E_FORMAT, "batch %lld: record size %zu does not match its stream lengths", (long long)bi, n);
        v.sep = v.enc + (size_t)v.ne * 4;
        v.col = v.sep + (size_t)v.ns * 4;
        sum_ne += v.ne; sum_ns += v.ns;
        if (c->enc_ptr + sum_ne > c->enc_words - PCR_ENCODED_PAD_WORDS || c->sep_ptr + sum_ns > c->sep_words - PCR_SEPARATE_PAD_WORDS)
            return set_err(c, PCR_E_FORMAT, "batch %lld: streams exceed the header's byte counts", (long long)bi);
        for (int i = 0; i < 4096; ++i) {                              // code lengths (narrowed to int8 below)
            int32_t l; std::memcpy(&l, v.tl + 4 * i, 4);
            if (l == 0 || l > PCR_MAX_CW_LEN || l < -PCR_MAX_CW_LEN)
                return set_err(c, PCR_E_FORMAT, "batch %lld: decoder table entry %d has code length %d", (long long)bi, i, l);
        }
        int32_t prev = 0;                                             // monotone prefixes keep in-range reads inside the batch
        for (int i = 0; i < 32; ++i) { int32_t x; std::memcpy(&x, v.cl + 4 * i, 4); if (x < prev) return set_err(c, PCR_E_FORMAT, "batch %lld: cluster sizes not monotone", (long long)bi); prev = x; }
        prev = 0;
        for (int i = 0; i < 1024; ++i) { int32_t x; std::memcpy(&x, v.sepsz + 4 * i, 4); if (x < prev) return set_err(c, PCR_E_FORMAT, "batch %lld: separate sizes not monotone", (long long)bi); prev = x; }
    }

    // ---- pass 2: pack into the free pinned arena ------------------------------------------------------------------
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->color_bytes == 0) {
        // the first records of the stream: its colour format is known now (the colour array was allocated for BC1)
        if (color_bytes != PCR_COLOR_BYTES_PER_BATCH) {
            const size_t nB = (size_t)c->hdr.num_batches;
            dfree_counted(c, c->d_colors, nB * PCR_COLOR_BYTES_PER_BATCH);
            dfree_counted(c, c->d_colors_t, nB * PCR_COLOR_BYTES_PER_BATCH);
            int rc = dalloc_zero(c, c->d_colors, nB * color_bytes, &c->stream_bytes);
            if (!rc) rc = dalloc_zero(c, c->d_colors_t, nB * color_bytes, &c->stream_bytes);
            if (rc) return rc;
            // the zero fill was enqueued on the context's stream; the copies and k_transcode's colors_t writes below may run on
            // the loader stream (pcr_set_async_upload), which nothing orders behind it: drain it first, as pcr_stream_begin does
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        c->color_bytes = color_bytes;
    }
    const size_t nb = (size_t)count;
    const size_t o_batches = 0;
    const size_t o_start = o_batches + nb * sizeof(pcr_gpu_batch);
    const size_t o_sepsz = o_start + nb * 3072 * 4;
    const size_t o_tv = o_sepsz + nb * 1024 * 4;
    const size_t o_tl = o_tv + nb * 4096 * 4;
    const size_t o_cl = o_tl + nb * 4096;
    const size_t o_col = o_cl + nb * 32 * 4;
    const size_t o_enc = o_col + nb * color_bytes;
    const size_t o_sep = o_enc + (size_t)sum_ne * 4;
    const size_t total = o_sep + (size_t)sum_ns * 4;
    const int slot = c->arena_next;
    c->arena_next ^= 1;
    if (c->arena_busy[slot]) { HIP_TRY(c, hipEventSynchronize(c->arena_done[slot])); c->arena_busy[slot] = false; }
    if (c->arena_size[slot] < total) {
        if (c->arena[slot]) { (void)hipHostFree(c->arena[slot]); c->arena[slot] = nullptr; c->arena_size[slot] = 0; }
        const size_t want = total + total / 4 + 4096;
        HIP_TRY(c, hipHostMalloc((void **)&c->arena[slot], want, hipHostMallocDefault));
        c->arena_size[slot] = want;
    }
    uint8_t *A = c->arena[slot];
    int64_t enc_ptr = c->enc_ptr, sep_ptr = c->sep_ptr;
    size_t eo = 0, so = 0;
    for (int64_t k = 0; k < count; ++k) {
        const View &v = views[(size_t)k];
        const uint8_t *r = (const uint8_t *)blobs[k];
        double sc[3], of[3]; float bmin[3], bmax[3], lmin[3], lmax[3];
        std::memcpy(sc, r + 20, 24); std::memcpy(of, r + 44, 24);
        std::memcpy(bmin, r + 68, 12); std::memcpy(bmax, r + 80, 12);
        std::memcpy(lmin, r + 92, 12); std::memcpy(lmax, r + 104, 12);
        pcr_gpu_batch g;                                  // HuffmanLasLoader.cpp:188-211
        g.min_x = bmin[0]; g.min_y = bmin[1]; g.min_z = bmin[2];
        g.max_x = bmax[0]; g.max_y = bmax[1]; g.max_z = bmax[2];
        g.scale_x = sc[0]; g.scale_y = sc[1]; g.scale_z = sc[2];
        g.offset_x = of[0]; g.offset_y = of[1]; g.offset_z = of[2];
        g.las_min_x = lmin[0]; g.las_min_y = lmin[1]; g.las_min_z = lmin[2];
        g.las_max_x = lmax[0]; g.las_max_y = lmax[1]; g.las_max_z = lmax[2];
        g.encoding_batch_offset = enc_ptr;
        g.separate_batch_offset = sep_ptr;
        g.decoder_table_offset = (first_index + k) * 4096;
        g.cluster_sizes_offset = (first_index + k) * 32;
        g.max_cw_len = PCR_MAX_CW_LEN;
        const size_t kk = (size_t)k;
        std::memcpy(A + o_batches + kk * sizeof g, &g, sizeof g);
        std::memcpy(A + o_start + kk * 3072 * 4, v.start, 3072 * 4);
        std::memcpy(A + o_sepsz + kk * 1024 * 4, v.sepsz, 1024 * 4);
        std::memcpy(A + o_tv + kk * 4096 * 4, v.tv, 4096 * 4);
        int8_t *tl8 = (int8_t *)(A + o_tl + kk * 4096);
        for (int i = 0; i < 4096; ++i) { int32_t l; std::memcpy(&l, v.tl + 4 * i, 4); tl8[i] = (int8_t)l; }   // render.cu:393
        std::memcpy(A + o_cl + kk * 32 * 4, v.cl, 32 * 4);
        std::memcpy(A + o_col + kk * color_bytes, v.col, color_bytes);
        std::memcpy(A + o_enc + eo, v.enc, (size_t)v.ne * 4); eo += (size_t)v.ne * 4;
        std::memcpy(A + o_sep + so, v.sep, (size_t)v.ns * 4); so += (size_t)v.ns * 4;
        enc_ptr += v.ne; sep_ptr += v.ns;
    }

    // ---- nine copies for the whole task ---------------------------------------------------------------------------
    const size_t b0 = (size_t)first_index;
    hipStream_t st = c->async_upload ? c->copy_stream : c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->d_batches + b0, A + o_batches, nb * sizeof(pcr_gpu_batch), hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_start + b0 * 3072, A + o_start, nb * 3072 * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_sep_sizes + b0 * 1024, A + o_sepsz, nb * 1024 * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_table_values + b0 * 4096, A + o_tv, nb * 4096 * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_table_lens + b0 * 4096, A + o_tl, nb * 4096, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_cluster_sizes + b0 * 32, A + o_cl, nb * 32 * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_colors + b0 * color_bytes, A + o_col, nb * color_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_encoded + c->enc_ptr, A + o_enc, (size_t)sum_ne * 4, hipMemcpyHostToDevice, st));
    if (sum_ns) HIP_TRY(c, hipMemcpyAsync(c->d_separate + c->sep_ptr, A + o_sep, (size_t)sum_ns * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipEventRecord(c->arena_done[slot], st));
    c->arena_busy[slot] = true;
    // the caller's records were copied into the arena: they may be released on return; the device copies complete in
    // stream order before any later render call
    c->enc_ptr = enc_ptr; c->sep_ptr = sep_ptr;
    for (const View &v : views) c->h_stream_words.push_back(v.ne + v.ns);
    c->batches_loaded += count; c->points_loaded += count * PCR_POINTS_PER_BATCH;   // HuffmanLasLoader.cpp:294-295
    { int trc = enqueue_transcode(c, false, st); if (trc) return trc; }    // this context's HBM layout of the stream is part of loading it
    HIP_TRY(c, hipGetLastError());
    if (c->async_upload) {
        // a batch becomes drawable with the task that brings the words behind it (its chains' tail over-reads)
        hipEvent_t ev;
        if (c->loader_events.empty()) HIP_TRY(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        else { ev = c->loader_events.back(); c->loader_events.pop_back(); }
        HIP_TRY(c, hipEventRecord(ev, st));
        c->loader_tasks.push_back({ev, c->transcoded});
    }
    return PCR_OK;
}

int pcr_upload_tail(pcr_ctx *c, const uint32_t *enc, size_t n_enc, const int32_t *sep, size_t n_sep)
{
    if (!c) return PCR_E_ARG;
    c->prepass_ready = false;        // what pcr_frame_begin prepared no longer matches the context's state
    if (!c->stream_open) return set_err(c, PCR_E_ARG, "no stream");
    if (c->finalized) {
        if (n_enc == 0 && n_sep == 0) return PCR_OK;
        return set_err(c, PCR_E_ARG, "the stream was finalised by a frame (raw words released): upload the shard tail before the first render call");
    }
    if (n_enc > PCR_ENCODED_PAD_WORDS || n_sep > PCR_SEPARATE_PAD_WORDS)
        return set_err(c, PCR_E_ARG, "tail larger than the pads (%d / %d words)", PCR_ENCODED_PAD_WORDS, PCR_SEPARATE_PAD_WORDS);
    if ((size_t)(c->enc_words - c->enc_ptr) < n_enc || (size_t)(c->sep_words - c->sep_ptr) < n_sep)
        return set_err(c, PCR_E_ARG, "tail does not fit behind the uploaded batches");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->async_upload) poll_loader(c, true);
    if (n_enc) HIP_TRY(c, hipMemcpyAsync(c->d_encoded + c->enc_ptr, enc, n_enc * 4, hipMemcpyHostToDevice, c->stream));
    if (n_sep) HIP_TRY(c, hipMemcpyAsync(c->d_separate + c->sep_ptr, sep, n_sep * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->batches_loaded > 0 && c->transcoded >= c->batches_loaded) c->transcoded = c->batches_loaded - 1;   // its over-reads see these words
    c->provisional_for = -1;
    if (c->async_upload) {           // no render-time transcode in this mode: redo the last batch now
        { int trc = enqueue_transcode(c, true, c->stream); if (trc) return trc; }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->batches_resident = c->transcoded;
    }
    return PCR_OK;
}

int pcr_stream_unload(pcr_ctx *c)
{
    if (!c) return PCR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_stream_buffers(c);
    return PCR_OK;
}

int pcr_set_stream_layout(pcr_ctx *c, int layout)
{
    if (!c) return PCR_E_ARG;
    if (layout != PCR_LAYOUT_WORDS && layout != PCR_LAYOUT_POINT_WINDOWS && layout != PCR_LAYOUT_BOTH && layout != PCR_LAYOUT_AUTO)
        return set_err(c, PCR_E_ARG, "unknown stream layout %d", layout);
    c->next_layout = layout;         // the stream that is loaded keeps the layout it was loaded with
    return PCR_OK;
}

int pcr_set_hbm_budget(pcr_ctx *c, int64_t bytes)
{
    if (!c || bytes < 0) return PCR_E_ARG;
    c->hbm_budget = bytes;
    return PCR_OK;
}

int pcr_stream_layout(const pcr_ctx *c) { return c && c->stream_open ? c->layout : -1; }

int pcr_set_workgroup_parts(pcr_ctx *c, int parts)
{
    if (!c) return PCR_E_ARG;
    if (parts < 0 || parts > MAX_PARTS) return set_err(c, PCR_E_ARG, "workgroups per batch: 0 (chosen per frame), 1 or 2, not %d", parts);
    c->parts_mode = parts;
    return PCR_OK;
}

int pcr_set_render_variant(pcr_ctx *c, int variant)
{
    if (!c) return PCR_E_ARG;
    if (variant != PCR_VARIANT_AUTO && variant != PCR_VARIANT_WORDS && variant != PCR_VARIANT_POINT_WINDOWS)
        return set_err(c, PCR_E_ARG, "unknown render variant %d", variant);
    c->variant = variant;
    return PCR_OK;
}

int64_t pcr_batches_loaded(const pcr_ctx *c) { return c ? c->batches_loaded : 0; }

int pcr_set_async_upload(pcr_ctx *c, int on)
{
    if (!c) return PCR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((on != 0) == c->async_upload) return PCR_OK;
    if (on) {
        if (!c->copy_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipStreamSynchronize(c->stream));         // earlier loader work (zero fill, copies, transcode) is behind us
        c->batches_resident = c->transcoded;                 // a provisional last batch is drawn again once its successor arrives
        c->async_upload = true;
    } else {
        poll_loader(c, true);
        HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
        c->async_upload = false;
    }
    return PCR_OK;
}

int64_t pcr_last_frame_batches(const pcr_ctx *c) { return c ? c->last_frame_batches : 0; }

int64_t pcr_batches_resident(pcr_ctx *c)
{
    if (!c) return 0;
    if (c->async_upload) poll_loader(c, false);
    return c->visible_batches();
}
int64_t pcr_points_loaded(const pcr_ctx *c) { return c ? c->points_loaded : 0; }

int64_t pcr_stream_resident_bytes(const pcr_ctx *c) { return c && c->stream_open ? (int64_t)c->stream_bytes : 0; }
int pcr_stream_color_format(const pcr_ctx *c)
{
    if (!c || !c->stream_open || c->color_bytes == 0) return 0;
    return c->color_bytes == PCR_COLOR_BYTES_PER_BATCH_BC7 ? PCR_COLOR_BC7 : PCR_COLOR_BC1;
}

int64_t pcr_stream_algorithmic_bytes(const pcr_ctx *c)
{
    if (!c || !c->stream_open) return 0;
    // SURVEY 8d, B_dec: every byte of the compressed representation once = encoded + separate + cluster prefix (128 B per
    // batch) + per-batch side data as the file holds it (GPUBatch 160 + start values 12 288 + escape prefix 4 096 + the
    // two int32 decoder-table arrays 32 768). k_render itself reads the table as 16 KiB of packed entries.
    const int64_t per_batch = 128 + 160 + 12288 + 4096 + 32768;
    return c->enc_ptr * 4 + c->sep_ptr * 4 + c->batches_loaded * per_batch;
}

int64_t pcr_last_frame_algorithmic_bytes(pcr_ctx *c)
{
    if (!c || !c->stream_open || c->last_frame_batches <= 0) return 0;
    const size_t nB = (size_t)std::min<int64_t>(c->last_frame_batches, (int64_t)c->h_stream_words.size());
    std::vector<uint32_t> lod(nB);
    if (hipSetDevice(c->device) != hipSuccess || hipMemcpyAsync(lod.data(), c->d_lod, nB * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    const int64_t per_batch = 128 + 160 + 12288 + 4096 + 32768;
    int64_t bytes = 0;
    for (size_t b = 0; b < nB; ++b) {
        const uint32_t npr = lod[b] & LOD_NPR_MASK;
        if ((lod[b] & LOD_CULLED) || npr == 0) continue;
        bytes += per_batch + (4 * (int64_t)c->h_stream_words[b] * (int64_t)npr) / PCR_POINTS_PER_THREAD;
    }
    return bytes;
}

// ---- method ------------------------------------------------------------------------------------
int pcr_set_image_size(pcr_ctx *c, int w, int h)
{
    if (!c) return PCR_E_ARG;
    c->prepass_ready = false;        // what pcr_frame_begin prepared no longer matches the context's state
    if (w <= 0 || h <= 0 || (int64_t)w * (h + 1) + 1 > 0x7FFFFFFF) return set_err(c, PCR_E_ARG, "bad image size %dx%d", w, h);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const bool external = c->fb && c->fb != c->own_fb;
    if (external) return set_err(c, PCR_E_ARG, "release external buffers before resizing");
    free_frame_buffers(c);
    c->width = w; c->height = h; c->fb_elems = pcr_fb_elems(w, h);
    // PCR_FRAME_PAD_ELEMS behind every buffer: a frame cut into N equal slices for the sliced multi-GPU exchange
    // (include/pcr_dist.h) reaches up to 2 N words past fb_elems. The pads are written here once -- identity of min / sum --
    // and by nothing else but those collectives, which leave them as they were.
    c->fb_alloc = c->fb_elems + PCR_FRAME_PAD_ELEMS;
    HIP_TRY(c, hipMalloc((void **)&c->own_fb, c->fb_alloc * 8));
    HIP_TRY(c, hipMalloc((void **)&c->own_rg, c->fb_alloc * 8));
    HIP_TRY(c, hipMalloc((void **)&c->own_ba, c->fb_alloc * 8));
    HIP_TRY(c, hipMalloc((void **)&c->d_rgba, c->fb_alloc * 4));
    HIP_TRY(c, hipMemsetAsync(c->own_fb + c->fb_elems, 0xFF, PCR_FRAME_PAD_ELEMS * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->own_rg + c->fb_elems, 0, PCR_FRAME_PAD_ELEMS * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->own_ba + c->fb_elems, 0, PCR_FRAME_PAD_ELEMS * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_rgba, 0, c->fb_alloc * 4, c->stream));
    c->tiles_x = (((uint32_t)w - 1u) >> TILE_W_SHIFT) + 1u;
    c->ntiles = c->tiles_x * ((((uint32_t)h + 1u) >> TILE_H_SHIFT) + 1u);        // rows 0 .. h + 1: pixel ids reach w * (h + 1)
    c->tiles_stride = (c->ntiles + 15u) & ~15u;
    HIP_TRY(c, hipMalloc((void **)&c->d_tiles, 3 * (size_t)c->tiles_stride + 16));
    HIP_TRY(c, hipMemsetAsync(c->d_tiles, 0, 3 * (size_t)c->tiles_stride + 16, c->stream));
    c->tile_cur = 0; c->tile_prev = 1; c->tile_spare = 2; c->tile_e_cur = 1; c->tile_e_prev = 0; c->tile_epochs = 1; c->tiles_tracked = false;
    c->fb = c->own_fb; c->rg = c->own_rg; c->ba = c->own_ba;
    c->accum_dirty = true;
    return pcr_clear(c);
}

// Tile flags as a clearing kernel has to leave them (clear_tile_flags); afterwards the context keeps track again. The array that was
// being marked is folded into the image-state array and zeroed by that kernel -- and retired to "spare": the frame that follows marks
// the all-zero third array, so a prepass running in the SAME launch as the clear (pcr_frame_begin, pcr_frame_turn) does not race with
// the zeroing. Call before make_args.
static TileFlags tiles_at_clear(pcr_ctx *c)
{
    TileFlags t = { nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0 };
    if (c->tiles_usable()) {
        t.cur = c->tiles_half(c->tile_cur); t.prev = c->tiles_half(c->tile_prev); t.cur_all = c->tiles_all(c->tile_cur); t.prev_all = c->tiles_all(c->tile_prev);
        t.ntiles = c->ntiles; t.tracked = c->tiles_tracked ? 1 : 0;
        if (c->tile_e_prev == 0) c->tile_e_prev = ++c->tile_epochs;     // (never used yet: any value no word holds)
        t.e_cur = c->tile_e_cur; t.e_prev = c->tile_e_prev;
        std::swap(c->tile_cur, c->tile_spare);
        c->tile_e_cur = ++c->tile_epochs;       // (a fresh epoch: the new array's "everything" word cannot hold it yet)
        c->tiles_tracked = true;
    } else {
        c->tiles_tracked = false;
    }
    return t;
}

int pcr_clear(pcr_ctx *c)
{
    if (!c) return PCR_E_ARG;
    c->prepass_ready = false;        // what pcr_frame_begin prepared no longer matches the context's state
    if (!c->fb) return set_err(c, PCR_E_ARG, "no framebuffer");
    HIP_TRY(c, hipSetDevice(c->device));
    // huffman_hqs.h:267-269. HuffmanMemIter clears only fb (huffman_mem_iter_cuda.h:250-252); RG/BA are re-zeroed only
    // after a colour pass (or when the buffers changed hands), which saves two 16.6 MB fills per basic frame at 1080p
    uint64_t *rg = c->accum_dirty ? c->rg : nullptr, *ba = c->accum_dirty ? c->ba : nullptr;
    if (((uintptr_t)c->fb | (uintptr_t)rg | (uintptr_t)ba) & 15) return set_err(c, PCR_E_ARG, "framebuffers must be 16-byte aligned");
    hipLaunchKernelGGL(k_clear, dim3(2048), dim3(256), 0, c->stream, c->fb, rg, ba, c->fb_elems, c->empty_key, tiles_at_clear(c));
    HIP_TRY(c, hipGetLastError());
    c->accum_dirty = false;
    return PCR_OK;
}

int pcr_frame_begin(pcr_ctx *c, const pcr_render_params *p, int method)
{
    int rc = check_params(c, p);
    if (rc) return rc;
    if (method != PCR_METHOD_BASIC && method != PCR_METHOD_HQS) return set_err(c, PCR_E_ARG, "unknown method %d", method);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->async_upload) poll_loader(c, false);
    const int64_t nB = c->visible_batches();
    if (nB == 0) return pcr_clear(c);
    uint64_t *rg = c->accum_dirty ? c->rg : nullptr, *ba = c->accum_dirty ? c->ba : nullptr;
    if (((uintptr_t)c->fb | (uintptr_t)rg | (uintptr_t)ba) & 15) return set_err(c, PCR_E_ARG, "framebuffers must be 16-byte aligned");
    if (!c->async_upload && (rc = enqueue_transcode(c, true, c->stream))) return rc;   // the prepass sorts batches by what k_transcode found out about them
    maybe_finalize(c);
    const TileFlags tflags = tiles_at_clear(c);
    RenderArgs a = make_args(c, p, method != PCR_METHOD_BASIC);
    a.win_pixel_bytes = WIN_PIXEL_BYTES;         // first pass of either method (basic / HQS depth)
    if (method == PCR_METHOD_HQS) a.win_hqs = c->d_win + c->hdr.num_batches * MAX_PARTS;     // ... and the colour pass's plan with it
    a.dyn_lds_bytes = frame_dyn_lds(c, nB);
    c->stats_partials = (int)((nB + PREPASS_BATCHES - 1) / PREPASS_BATCHES);
    const uint32_t prepass_wgs = (uint32_t)c->stats_partials * PREPASS_WGS_PER_CHUNK;       // (two workgroups per chunk: lists / plans)
    hipLaunchKernelGGL(k_frame_begin, dim3(prepass_wgs + 2048u), dim3(256), 0, c->stream, a,
                       prepass_wgs, c->fb, rg, ba, c->fb_elems, c->empty_key, tflags);
    HIP_TRY(c, hipGetLastError());
    c->accum_dirty = false;
    c->prepass_ready = true;
    c->prepass_params = *p; c->prepass_variant_hqs = a.variant_hqs; c->prepass_slots = a.win_hqs ? 3u : 1u;
    c->prepass_dyn_lds = a.dyn_lds_bytes; c->prepass_parts = a.parts;
    c->prepass_batches = nB;
    return PCR_OK;
}

static int launch_resolve(pcr_ctx *c, const pcr_render_params *p, bool hqs);

// The end of one frame and the start of the next in one launch: resolve (flags of p_done) -> d_rgba, CLEAR, and the cull/LOD
// prepass for p_next. Equivalent to pcr_resolve_* followed by pcr_frame_begin(p_next); the framebuffer is empty afterwards
// (read it before, if it is wanted: the depth dump does), the image is in pcr_read_rgba as after pcr_resolve_*.
int pcr_frame_turn(pcr_ctx *c, const pcr_render_params *p_done, const pcr_render_params *p_next, int method)
{
    int rc = check_params(c, p_next);
    if (rc) return rc;
    if (!p_done) return set_err(c, PCR_E_ARG, "params of the finished frame are NULL");
    if (p_done->width != c->width || p_done->height != c->height) return set_err(c, PCR_E_ARG, "params image size != framebuffer");
    if (method != PCR_METHOD_BASIC && method != PCR_METHOD_HQS) return set_err(c, PCR_E_ARG, "unknown method %d", method);
    const bool hqs = method == PCR_METHOD_HQS;
    if (hqs && (!c->rg || !c->ba)) return set_err(c, PCR_E_ARG, "no RG/BA accumulation buffers");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->async_upload) poll_loader(c, false);
    const int64_t nB = c->visible_batches();
    if (nB == 0) {                                   // nothing to prepare: the two separate steps
        if ((rc = launch_resolve(c, p_done, hqs))) return rc;
        return pcr_clear(c);
    }
    if (!hqs && c->accum_dirty) {                    // a basic frame behind an HQS one: RG/BA are zeroed by the ordinary CLEAR
        if ((rc = launch_resolve(c, p_done, false))) return rc;
        return pcr_frame_begin(c, p_next, method);
    }
    if (!c->async_upload && (rc = enqueue_transcode(c, true, c->stream))) return rc;
    maybe_finalize(c);
    RenderArgs a = make_args(c, p_next, hqs);
    a.win_pixel_bytes = WIN_PIXEL_BYTES;
    if (hqs) a.win_hqs = c->d_win + c->hdr.num_batches * MAX_PARTS;
    a.dyn_lds_bytes = frame_dyn_lds(c, nB);
    c->stats_partials = (int)((nB + PREPASS_BATCHES - 1) / PREPASS_BATCHES);
    const uint32_t prepass_wgs = (uint32_t)c->stats_partials * PREPASS_WGS_PER_CHUNK;       // (two workgroups per chunk: lists / plans)
    const uint32_t pixels = (uint32_t)((size_t)c->width * c->height);
    if (c->tiles_usable() && c->tiles_tracked) {
        // only the tiles something was written in (and those the image still holds something in): FrameView::tiles
        // what the finished frame marked (cur) and what the image may still hold (prev) are read, prev is left zeroed; this launch's
        // own prepass marks the NEXT frame's tiles in the third array (all zero since the turn before)
        TileFlags tf = { c->tiles_half(c->tile_cur), c->tiles_half(c->tile_prev), c->tiles_all(c->tile_cur), c->tiles_all(c->tile_prev),
                         c->ntiles, c->tile_e_cur, c->tile_e_prev, 1 };
        const int cur = c->tile_cur, prev = c->tile_prev, spare = c->tile_spare;
        c->tile_cur = spare; c->tile_prev = cur; c->tile_spare = prev;
        c->tile_e_prev = c->tile_e_cur; c->tile_e_cur = ++c->tile_epochs;
        a.f.tiles = c->tiles_half(c->tile_cur); a.f.tiles_all = c->tiles_all(c->tile_cur); a.f.tiles_epoch = c->tile_e_cur;      // (the prepass of the next frame)
        const unsigned grid = prepass_wgs + std::min<unsigned>(c->ntiles, 4096u);
        if (hqs) hipLaunchKernelGGL(k_frame_turn_tiles<true>, dim3(grid), dim3(256), 0, c->stream, a, prepass_wgs, p_done->show_num_points,
                                    p_done->colorize_chunks, (uint32_t)c->width, pixels, c->fb, c->rg, c->ba, c->d_rgba, (uint32_t)c->fb_elems, c->empty_key,
                                    tf, c->tiles_x);
        else     hipLaunchKernelGGL(k_frame_turn_tiles<false>, dim3(grid), dim3(256), 0, c->stream, a, prepass_wgs, p_done->show_num_points,
                                    p_done->colorize_chunks, (uint32_t)c->width, pixels, c->fb, c->rg, c->ba, c->d_rgba, (uint32_t)c->fb_elems, c->empty_key,
                                    tf, c->tiles_x);
    } else {
        const unsigned grid = prepass_wgs + 2048u;
        const TileFlags tflags = tiles_at_clear(c);
        a.f.tiles = c->tiles_tracked ? c->tiles_half(c->tile_cur) : nullptr;     // (the prepass of the next frame marks the fresh array)
        a.f.tiles_all = c->d_tiles ? c->tiles_all(c->tile_cur) : nullptr; a.f.tiles_epoch = c->tile_e_cur;
        if (hqs) hipLaunchKernelGGL(k_frame_turn<true>, dim3(grid), dim3(256), 0, c->stream, a, prepass_wgs, p_done->show_num_points,
                                    p_done->colorize_chunks, pixels, c->fb, c->rg, c->ba, c->d_rgba, (uint32_t)c->fb_elems, c->empty_key, tflags);
        else     hipLaunchKernelGGL(k_frame_turn<false>, dim3(grid), dim3(256), 0, c->stream, a, prepass_wgs, p_done->show_num_points,
                                    p_done->colorize_chunks, pixels, c->fb, c->rg, c->ba, c->d_rgba, (uint32_t)c->fb_elems, c->empty_key, tflags);
    }
    HIP_TRY(c, hipGetLastError());
    c->accum_dirty = false;
    c->prepass_ready = true;
    c->prepass_params = *p_next; c->prepass_variant_hqs = a.variant_hqs; c->prepass_slots = a.win_hqs ? 3u : 1u;
    c->prepass_dyn_lds = a.dyn_lds_bytes; c->prepass_parts = a.parts;
    c->prepass_batches = nB;
    return PCR_OK;
}

int pcr_set_int64_mergeable(pcr_ctx *c, int on)
{
    if (!c) return PCR_E_ARG;
    c->empty_key = on ? 0x7FFFFFFFFFFFFFFFull : ~0ull;
    c->tiles_tracked = false;       // a tile turn rewrites the empty word in the dirty tiles only: the next clear has to be a full one
    return PCR_OK;
}

int pcr_render_basic(pcr_ctx *c, const pcr_render_params *p)
{
    // COLOR_COMPRESSION == 7 is not a defined configuration of the reference's basic method: its rasterize decodes the colour
    // array as BC1 whatever the setting (huffman_mem_iter_cuda/render.cu:299) and its resolve then indexes that array with
    // the resulting COLOUR as if it were a point id (resolve.cu:183) -- an out-of-bounds read. Only the HQS method draws BC7.
    if (c && c->stream_open && c->color_bytes == PCR_COLOR_BYTES_PER_BATCH_BC7)
        return set_err(c, PCR_E_ARG, "the basic method is not defined for a stream with BC7 colours (use the HQS method)");
    return launch_render<MODE_BASIC>(c, p);
}
int pcr_render_hqs_depth(pcr_ctx *c, const pcr_render_params *p) { return launch_render<MODE_HQS_DEPTH>(c, p); }
int pcr_render_hqs_color(pcr_ctx *c, const pcr_render_params *p)
{
    if (c && (!c->rg || !c->ba)) return set_err(c, PCR_E_ARG, "no RG/BA accumulation buffers");
    if (c) c->accum_dirty = true;
    if (c && c->stream_open && c->color_bytes == PCR_COLOR_BYTES_PER_BATCH_BC7) return launch_render<MODE_HQS_COLOR_BC7>(c, p);
    return launch_render<MODE_HQS_COLOR>(c, p);
}

// ---- 10-10-10 path ------------------------------------------------------------------------------
int pcr_las_begin(pcr_ctx *c, int64_t num_points)
{
    if (!c) return PCR_E_ARG;
    if (num_points <= 0) return set_err(c, PCR_E_ARG, "num_points must be > 0");
    const int64_t nB = (num_points + PCR_POINTS_PER_BATCH - 1) / PCR_POINTS_PER_BATCH;
    if (nB * PCR_POINTS_PER_BATCH >= 0x7FFFFFFFll)         // the framebuffer key holds a 31-bit point index (resolve.cu)
        return set_err(c, PCR_E_ARG, "%lld points exceed the 31-bit point index of the 10-10-10 framebuffer", (long long)num_points);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_las_buffers(c);
    const size_t slots = (size_t)nB * PCR_POINTS_PER_BATCH;
    int rc;
    if ((rc = dalloc_zero(c, c->d_xyzb, (size_t)nB)) || (rc = dalloc_zero(c, c->d_xyz12, slots)) ||
        (rc = dalloc_zero(c, c->d_xyz8, slots)) || (rc = dalloc_zero(c, c->d_xyz4, slots)) ||
        (rc = dalloc_zero(c, c->d_point_rgba, slots)) || (rc = dalloc_zero(c, c->d_las_level, (size_t)nB)) ||
        (rc = dalloc_zero(c, c->d_las_win, (size_t)nB)) ||
        (rc = dalloc_zero(c, c->d_las_order, (size_t)((nB + LAS_PREPASS_BATCHES - 1) / LAS_PREPASS_BATCHES) * LAS_PREPASS_BATCHES)) ||
        (rc = dalloc_zero(c, c->d_las_chunk_count, (size_t)LAS_CLASSES * (size_t)((nB + LAS_PREPASS_BATCHES - 1) / LAS_PREPASS_BATCHES)))) {
        free_las_buffers(c);
        return rc == PCR_E_HIP ? set_err(c, PCR_E_NOMEM, "out of device memory for %lld batches", (long long)nB) : rc;
    }
    c->las_capacity = nB; c->las_loaded = 0; c->las_open = true;
    return PCR_OK;
}

int pcr_las_upload(pcr_ctx *c, int64_t first_batch, int64_t count, const pcr_xyz_batch *batches, const uint32_t *xyz12,
                   const uint32_t *xyz8, const uint32_t *xyz4, const uint32_t *rgba)
{
    if (!c) return PCR_E_ARG;
    if (!c->las_open) return set_err(c, PCR_E_ARG, "pcr_las_begin has not been called");
    if (!batches || !xyz12 || !xyz8 || !xyz4 || !rgba) return set_err(c, PCR_E_ARG, "NULL input array");
    if (count <= 0 || first_batch != c->las_loaded || first_batch + count > c->las_capacity)
        return set_err(c, PCR_E_ARG, "batches [%lld,%lld) out of order or beyond the %lld allocated (next expected %lld)",
                       (long long)first_batch, (long long)(first_batch + count), (long long)c->las_capacity, (long long)c->las_loaded);
    for (int64_t i = 0; i < count; ++i) {
        const pcr_xyz_batch &g = batches[i];
        if (!(g.min_x <= g.max_x && g.min_y <= g.max_y && g.min_z <= g.max_z) || g.num_points < 0 || g.num_points > PCR_POINTS_PER_BATCH)
            return set_err(c, PCR_E_FORMAT, "batch %lld: bad bounding box or point count", (long long)(first_batch + i));
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t off = (size_t)first_batch * PCR_POINTS_PER_BATCH, n = (size_t)count * PCR_POINTS_PER_BATCH * 4;
    HIP_TRY(c, hipMemcpyAsync(c->d_xyzb + first_batch, batches, (size_t)count * sizeof(pcr_xyz_batch), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_xyz12 + off, xyz12, n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_xyz8 + off, xyz8, n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_xyz4 + off, xyz4, n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_point_rgba + off, rgba, n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));             // the arrays are borrowed for the call only
    c->las_loaded += count;
    return PCR_OK;
}

int pcr_las_unload(pcr_ctx *c)
{
    if (!c) return PCR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_las_buffers(c);
    return PCR_OK;
}

int64_t pcr_las_batches_loaded(const pcr_ctx *c) { return c ? c->las_loaded : 0; }

static int check_las(pcr_ctx *c, const pcr_render_params *p)
{
    if (!c) return PCR_E_ARG;
    if (!p) return set_err(c, PCR_E_ARG, "render params are NULL");
    if (!c->las_open) return set_err(c, PCR_E_ARG, "no 10-10-10 data loaded (call pcr_las_begin / pcr_las_upload)");
    if (!c->fb) return set_err(c, PCR_E_ARG, "no framebuffer (call pcr_set_image_size)");
    if (p->width != c->width || p->height != c->height)
        return set_err(c, PCR_E_ARG, "params image size %dx%d != framebuffer %dx%d", p->width, p->height, c->width, c->height);
    return PCR_OK;
}

int pcr_render_las(pcr_ctx *c, const pcr_render_params *p)
{
    int rc = check_las(c, p);
    if (rc) return rc;
    const int64_t nB = c->las_loaded;
    if (nB == 0) { c->stats_partials = 0; return PCR_OK; }     // compute_loop_las_cuda.h:107
    c->prepass_ready = false;        // d_stats is shared with the Huffman methods' prepass
    LasArgs a;
    a.p = *p;
    a.s.batches = c->d_xyzb; a.s.xyz12 = c->d_xyz12; a.s.xyz8 = c->d_xyz8; a.s.xyz4 = c->d_xyz4; a.s.num_batches = nB;
    a.f.fb = c->fb; a.f.rg = c->rg; a.f.ba = c->ba; a.f.fb_elems = (uint32_t)c->fb_elems;
    a.f.tiles = nullptr; a.f.tiles_all = nullptr; a.f.tiles_x = 0; a.f.tiles_epoch = 0; a.f.tiles_total = 0;
    c->tiles_tracked = false;        // (this method's kernels do not mark tiles)
    a.level = c->d_las_level; a.win = c->d_las_win; a.stats = c->d_stats; a.win_capacity = WIN_PIXELS;
    c->stats_partials = (int)((nB + PREPASS_THREADS - 1) / PREPASS_THREADS);
    a.order = c->d_las_order; a.chunk_count = c->d_las_chunk_count; a.chunks = (uint32_t)c->stats_partials;
    hipLaunchKernelGGL(k_las_prepass, dim3((unsigned)c->stats_partials), dim3(PREPASS_THREADS), 0, c->stream, a);
    const bool timed = c->kt_sample_now();
    const int slot = (int)(c->kt_samples % pcr_ctx::KT_PAIRS);
    if (timed) HIP_TRY(c, hipEventRecord(c->kt_begin[slot], c->stream));
    hipLaunchKernelGGL(k_las_render, dim3((unsigned)nB), dim3(PCR_WORKGROUP_SIZE), 0, c->stream, a);
    if (timed) { HIP_TRY(c, hipEventRecord(c->kt_end[slot], c->stream)); ++c->kt_samples; }
    if (c->kt_every > 0) ++c->kt_launches;
    HIP_TRY(c, hipGetLastError());
    return PCR_OK;
}

int pcr_resolve_las(pcr_ctx *c, const pcr_render_params *p)
{
    int rc = check_las(c, p);
    if (rc) return rc;
    dim3 grid((unsigned)((c->width + 15) / 16), (unsigned)((c->height + 15) / 16));
    hipLaunchKernelGGL(k_las_resolve, grid, dim3(256), 0, c->stream, c->width, c->height, c->fb, c->d_point_rgba, c->d_rgba);
    HIP_TRY(c, hipGetLastError());
    return PCR_OK;
}

int64_t pcr_las_algorithmic_bytes(pcr_ctx *c)
{
    if (!c || !c->las_open || c->las_loaded == 0) return 0;
    std::vector<int32_t> level((size_t)c->las_loaded);
    if (hipMemcpyAsync(level.data(), c->d_las_level, level.size() * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) return 0;
    int64_t bytes = 0;
    for (int64_t b = 0; b + 1 < c->las_loaded; ++b) {         // the last workgroup does not run
        const int l = level[(size_t)b];
        if (l < 0) continue;
        bytes += (int64_t)sizeof(pcr_xyz_batch) + (int64_t)PCR_POINTS_PER_BATCH * 4 * (l >= 2 ? 1 : l == 1 ? 2 : 3);
    }
    return bytes;
}

static int launch_resolve(pcr_ctx *c, const pcr_render_params *p, bool hqs)
{
    if (!c) return PCR_E_ARG;
    if (!p) return set_err(c, PCR_E_ARG, "params are NULL");
    if (!c->fb) return set_err(c, PCR_E_ARG, "no framebuffer");
    if (p->width != c->width || p->height != c->height) return set_err(c, PCR_E_ARG, "params image size != framebuffer");
    dim3 grid((unsigned)((c->width + 15) / 16), (unsigned)((c->height + 15) / 16));   // huffman_hqs.h:249-250
    if (hqs) hipLaunchKernelGGL(k_resolve<true>, grid, dim3(256), 0, c->stream, p->show_num_points, p->colorize_chunks, c->width, c->height, c->fb, c->rg, c->ba, c->d_rgba);
    else     hipLaunchKernelGGL(k_resolve<false>, grid, dim3(256), 0, c->stream, p->show_num_points, p->colorize_chunks, c->width, c->height, c->fb, c->rg, c->ba, c->d_rgba);
    HIP_TRY(c, hipGetLastError());
    return PCR_OK;
}
int pcr_resolve_basic(pcr_ctx *c, const pcr_render_params *p) { return launch_resolve(c, p, false); }
int pcr_resolve_hqs(pcr_ctx *c, const pcr_render_params *p) { return launch_resolve(c, p, true); }

int pcr_get_stats(pcr_ctx *c, pcr_render_stats *out)
{
    if (!c || !out) return PCR_E_ARG;
    std::vector<pcr_render_stats> part((size_t)PCR_STATS_PARTIALS);
    std::memset(out, 0, sizeof *out);
    if (c->stats_partials > 0) {
        HIP_TRY(c, hipMemcpyAsync(part.data(), c->d_stats, (size_t)c->stats_partials * sizeof(pcr_render_stats), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < c->stats_partials; ++i) {
            out->batches_total += part[i].batches_total; out->batches_culled += part[i].batches_culled;
            out->points_iterated += part[i].points_iterated; out->batches_double += part[i].batches_double;
        }
    }
    return PCR_OK;
}

int pcr_read_framebuffer(pcr_ctx *c, uint64_t *host, size_t n)
{
    if (!c || !host) return PCR_E_ARG;
    if (!c->fb || n > c->fb_elems) return set_err(c, PCR_E_ARG, "bad framebuffer read of %zu elements", n);
    HIP_TRY(c, hipMemcpyAsync(host, c->fb, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->empty_key != ~0ull)                   // the caller always sees the reference's empty word
        for (size_t i = 0; i < n; ++i) if (host[i] == c->empty_key) host[i] = ~0ull;
    return PCR_OK;
}

int pcr_read_accum(pcr_ctx *c, uint64_t *hrg, uint64_t *hba, size_t n)
{
    if (!c || !hrg || !hba) return PCR_E_ARG;
    if (!c->rg || !c->ba || n > c->fb_elems) return set_err(c, PCR_E_ARG, "bad accumulation read");
    HIP_TRY(c, hipMemcpyAsync(hrg, c->rg, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hba, c->ba, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PCR_OK;
}

int pcr_read_rgba(pcr_ctx *c, uint32_t *host, size_t n)
{
    if (!c || !host) return PCR_E_ARG;
    if (!c->d_rgba || n > (size_t)c->width * c->height) return set_err(c, PCR_E_ARG, "bad rgba read");
    HIP_TRY(c, hipMemcpyAsync(host, c->d_rgba, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return PCR_OK;
}

// ---- multi-GPU plumbing ------------------------------------------------------------------------
void *pcr_get_stream(pcr_ctx *c) { return c ? (void *)c->stream : nullptr; }
int pcr_get_device(const pcr_ctx *c) { return c ? c->device : -1; }
size_t pcr_framebuffer_elems(const pcr_ctx *c) { return c ? c->fb_elems : 0; }
size_t pcr_framebuffer_capacity(const pcr_ctx *c)
{
    if (!c || !c->fb) return 0;
    return (c->fb == c->own_fb && c->rg == c->own_rg && c->ba == c->own_ba) ? c->fb_alloc : c->fb_elems;
}
void *pcr_device_rgba(pcr_ctx *c) { return c ? c->d_rgba : nullptr; }
// (whoever asks for these may write through them -- a collective merging partial frames in place: no tile flags until the next clear)
static void expose(pcr_ctx *c) { if (c) { c->tiles_tracked = false; c->fb_exposed = true; } }
void *pcr_device_framebuffer(pcr_ctx *c) { expose(c); return c ? c->fb : nullptr; }
void *pcr_device_rg(pcr_ctx *c) { expose(c); return c ? c->rg : nullptr; }
void *pcr_device_ba(pcr_ctx *c) { expose(c); return c ? c->ba : nullptr; }
int pcr_framebuffer_private(pcr_ctx *c)
{
    if (!c) return PCR_E_ARG;
    c->fb_exposed = false; c->tiles_tracked = false;        // (tracking resumes with the next full clear)
    return PCR_OK;
}

int pcr_use_external_buffers(pcr_ctx *c, void *fb, void *rg, void *ba)
{
    if (!c) return PCR_E_ARG;
    c->prepass_ready = false;        // what pcr_frame_begin prepared no longer matches the context's state
    if (!c->own_fb) return set_err(c, PCR_E_ARG, "call pcr_set_image_size first");
    // pointers are captured by value at enqueue time: switching them does not disturb work already enqueued
    uint64_t *nrg = rg ? (uint64_t *)rg : c->own_rg, *nba = ba ? (uint64_t *)ba : c->own_ba;
    if (nrg != c->rg || nba != c->ba) c->accum_dirty = true;     // unknown contents: the next pcr_clear zeroes them
    c->fb = fb ? (uint64_t *)fb : c->own_fb;
    c->rg = nrg;
    c->ba = nba;
    c->tiles_tracked = false;
    return PCR_OK;
}

int pcr_merge_min_slices(pcr_ctx *c, void *slices, int nslices, size_t slice_elems)
{
    if (!c || !slices || nslices < 1 || slice_elems == 0) return PCR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (nslices > 1) {
        // few workgroups: this runs next to a render, whose 1024-thread workgroups wait for whole CUs' worth of wave slots
        hipLaunchKernelGGL(k_merge_min_slices, dim3(128), dim3(256), 0, c->stream, (uint64_t *)slices, nslices, slice_elems);
        HIP_TRY(c, hipGetLastError());
    }
    return PCR_OK;
}

int pcr_resolve_basic_range(pcr_ctx *c, const pcr_render_params *p, const void *fb, size_t count, void *rgba)
{
    if (!c || !p || !fb || !rgba) return PCR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (count == 0) return PCR_OK;
    hipLaunchKernelGGL(k_resolve_range, dim3(128), dim3(256), 0, c->stream, p->show_num_points, p->colorize_chunks,
                       (const uint64_t *)fb, count, (uint32_t *)rgba);
    HIP_TRY(c, hipGetLastError());
    return PCR_OK;
}

int pcr_resolve_hqs_range(pcr_ctx *c, const pcr_render_params *p, const void *fb, const void *rg, const void *ba, size_t count, void *rgba)
{
    if (!c || !p || !fb || !rg || !ba || !rgba) return PCR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (count == 0) return PCR_OK;
    hipLaunchKernelGGL(k_resolve_range_hqs, dim3(128), dim3(256), 0, c->stream, p->show_num_points, p->colorize_chunks,
                       (const uint64_t *)fb, (const uint64_t *)rg, (const uint64_t *)ba, count, (uint32_t *)rgba);
    HIP_TRY(c, hipGetLastError());
    return PCR_OK;
}

int pcr_fence_record(pcr_ctx *c, int slot, void *hip_stream)
{
    if (!c || slot < 0 || slot >= pcr_ctx::FENCES) return PCR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->fence[slot]) HIP_TRY(c, hipEventCreateWithFlags(&c->fence[slot], hipEventDisableTiming | hipEventReleaseToDevice));
    HIP_TRY(c, hipEventRecord(c->fence[slot], hip_stream ? (hipStream_t)hip_stream : c->stream));
    return PCR_OK;
}

int pcr_fence_wait(pcr_ctx *c, int slot, void *hip_stream)
{
    if (!c || slot < 0 || slot >= pcr_ctx::FENCES) return PCR_E_ARG;
    if (!c->fence[slot]) return PCR_OK;          // never recorded: nothing to wait for
    HIP_TRY(c, hipStreamWaitEvent(hip_stream ? (hipStream_t)hip_stream : c->stream, c->fence[slot], 0));
    return PCR_OK;
}

int pcr_merge_min(pcr_ctx *c, const void *other)
{
    if (!c || !other) return PCR_E_ARG;
    c->tiles_tracked = false;
    if (!c->fb) return set_err(c, PCR_E_ARG, "no framebuffer");
    hipLaunchKernelGGL(k_merge_min, dim3(2048), dim3(256), 0, c->stream, c->fb, (const uint64_t *)other, (uint32_t)c->fb_elems);
    HIP_TRY(c, hipGetLastError());
    return PCR_OK;
}

int pcr_merge_sum(pcr_ctx *c, const void *org, const void *oba)
{
    if (!c) return PCR_E_ARG;
    c->tiles_tracked = false;
    if (!c->rg || !c->ba) return set_err(c, PCR_E_ARG, "no accumulation buffers");
    c->accum_dirty = true;
    if (org) hipLaunchKernelGGL(k_merge_sum, dim3(2048), dim3(256), 0, c->stream, c->rg, (const uint64_t *)org, (uint32_t)c->fb_elems);
    if (oba) hipLaunchKernelGGL(k_merge_sum, dim3(2048), dim3(256), 0, c->stream, c->ba, (const uint64_t *)oba, (uint32_t)c->fb_elems);
    HIP_TRY(c, hipGetLastError());
    return PCR_OK;
}

int pcr_flip_sign(pcr_ctx *c)
{
    if (!c) return PCR_E_ARG;
    c->tiles_tracked = false;
    if (!c->fb) return set_err(c, PCR_E_ARG, "no framebuffer");
    hipLaunchKernelGGL(k_flip_sign, dim3(2048), dim3(256), 0, c->stream, c->fb, (uint32_t)c->fb_elems);
    HIP_TRY(c, hipGetLastError());
    return PCR_OK;
}

// ---- measurement -------------------------------------------------------------------------------
int pcr_timing_begin(pcr_ctx *c)
{
    if (!c) return PCR_E_ARG;
    HIP_TRY(c, hipEventRecord(c->ev_begin, c->stream));
    return PCR_OK;
}

int pcr_timing_end(pcr_ctx *c, float *ms)
{
    if (!c || !ms) return PCR_E_ARG;
    HIP_TRY(c, hipEventRecord(c->ev_end, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->ev_end));
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev_begin, c->ev_end));
    return PCR_OK;
}

int pcr_measure_hbm(pcr_ctx *c, size_t bytes, int reps, float *read_gbps, float *copy_gbps)
{
    if (!c || !read_gbps || !copy_gbps || reps <= 0 || bytes < (1u << 20)) return PCR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n16 = bytes / 16;
    hbm_vec4 *a = nullptr, *b = nullptr;
    uint32_t *sink = nullptr;
    auto cleanup = [&] { if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (sink) (void)hipFree(sink); };
    if (hipMalloc((void **)&a, n16 * 16) != hipSuccess || hipMalloc((void **)&b, n16 * 16) != hipSuccess ||
        hipMalloc((void **)&sink, 4) != hipSuccess) {
        cleanup();
        return set_err(c, PCR_E_NOMEM, "pcr_measure_hbm: cannot allocate 2 x %zu bytes", n16 * 16);
    }
    int rc = PCR_OK;
    auto run = [&](bool copy, float *out) {
        const unsigned grid = 256 * 16;          // 16 workgroups of 256 threads per CU
        float best = 0.0f;
        for (int r = 0; r < reps + 1; ++r) {     // first pass untimed
            if (hipEventRecord(c->ev_begin, c->stream) != hipSuccess) { rc = PCR_E_HIP; return; }
            if (copy) hipLaunchKernelGGL(k_hbm_copy, dim3(grid), dim3(256), 0, c->stream, a, b, n16);
            else hipLaunchKernelGGL(k_hbm_read, dim3(grid), dim3(256), 0, c->stream, a, n16, sink);
            float ms = 0.0f;
            if (hipEventRecord(c->ev_end, c->stream) != hipSuccess || hipEventSynchronize(c->ev_end) != hipSuccess ||
                hipEventElapsedTime(&ms, c->ev_begin, c->ev_end) != hipSuccess) { rc = PCR_E_HIP; return; }
            const float gbps = (float)((copy ? 2.0 : 1.0) * (double)(n16 * 16) / (ms * 1e-3) / 1e9);
            if (r > 0 && gbps > best) best = gbps;
        }
        *out = best;
    };
    if (hipMemsetAsync(a, 0x5A, n16 * 16, c->stream) != hipSuccess || hipMemsetAsync(b, 0, n16 * 16, c->stream) != hipSuccess)
        rc = PCR_E_HIP;
    if (rc == PCR_OK) run(false, read_gbps);
    if (rc == PCR_OK) run(true, copy_gbps);
    (void)hipStreamSynchronize(c->stream);
    cleanup();
    if (rc != PCR_OK) return set_err(c, rc, "pcr_measure_hbm: %s", hipGetErrorString(hipGetLastError()));
    return PCR_OK;
}

int pcr_kernel_timing_enable(pcr_ctx *c, int every)
{
    if (!c || every < 0) return PCR_E_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (every > 0)
        for (int i = 0; i < pcr_ctx::KT_PAIRS; ++i) {
            // device-scope release: the default system-scope one writes the L2 back around the kernel being timed
            if (!c->kt_begin[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->kt_begin[i], hipEventReleaseToDevice));
            if (!c->kt_end[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->kt_end[i], hipEventReleaseToDevice));
        }
    c->kt_every = every;
    c->kt_launches = 0;
    c->kt_samples = 0;
    return PCR_OK;
}

int pcr_kernel_timing_read(pcr_ctx *c, float *avg_ms, int *launches)
{
    if (!c || !avg_ms || !launches) return PCR_E_ARG;
    *avg_ms = 0.0f; *launches = 0;
    const int n = (int)std::min<int64_t>(c->kt_samples, pcr_ctx::KT_PAIRS);
    if (n == 0) return PCR_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->kt_begin[i], c->kt_end[i]));
        sum += ms;
    }
    *avg_ms = (float)(sum / n); *launches = n;
    return PCR_OK;
}

#ifdef PCR_EXP_FAR_STATS
int pcr_exp_read_far(pcr_ctx *c, unsigned long long *out, int reset)
{
    if (!c || !out) return PCR_E_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_far), 8 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[8] = {0}; HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(pcr::g_far), z, sizeof z)); }
    return PCR_OK;
}
#endif
#ifdef PCR_EXP_TIMELINE
int pcr_exp_read_timeline(pcr_ctx *c, unsigned long long *out, size_t n)
{
    if (!c || !out) return PCR_E_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_timeline), n * sizeof(unsigned long long)));
    return PCR_OK;
}
int pcr_exp_read_wave_ends(pcr_ctx *c, unsigned long long *out, size_t n)
{
    if (!c || !out) return PCR_E_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(pcr::g_wave_end), n * sizeof(unsigned long long)));
    return PCR_OK;
}
#endif

} // extern "C"

#include "pcr_gpu_encoder.hip.h"   // pcr_gpu_encode_points (include/pcr_gpu_encode.h)
