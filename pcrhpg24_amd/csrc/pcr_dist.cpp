// pcr_dist.cpp — implementation of include/pcr_dist.h (libpcr_dist.so): batch-sharded rendering over the GPUs of a node,
// partial framebuffers merged with RCCL's native unsigned 64-bit min / sum, in place, on each context's own stream.
// No reference counterpart (the reference is single-GPU); this is north_star's multi-GPU step on the C++ side.
#include "pcr_dist.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

struct pcr_dist {
    pcr_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int exchange = PCR_DIST_EXCHANGE_AUTO;
    // PCR_DIST_EXCHANGE_SLICED_P2P: everyone's copy of the slice this rank owns, back to back (world x slice words)
    uint64_t *scratch = nullptr;
    size_t scratch_elems = 0;
    const void *merged_slice = nullptr;          // where the last pcr_dist_merge_min_sliced left this rank's merged slice
    size_t pending_min_slices = 0;               // P2P: words per slice of an all-to-all whose local min has not been enqueued yet
};

namespace {
thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define NCCL_TRY(call)                                                                                     \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess) return fail(PCR_E_HIP, "%s failed: %s", #call, ncclGetErrorString(r_));     \
    } while (0)
#define PCR_TRY(d, call)                                                                                   \
    do {                                                                                                   \
        int r_ = (call);                                                                                   \
        if (r_ != PCR_OK) return fail(r_, "rank %d: %s: %s", (d)->rank, #call, pcr_last_error((d)->ctx));  \
    } while (0)

struct Slice { size_t first, count; };
Slice slice_of(size_t elems, int world, int rank)
{
    const size_t S = ((elems + (size_t)world - 1) / (size_t)world + 1) & ~(size_t)1;
    return { (size_t)rank * S, S };
}

// the frame cut into d->world slices has to fit the context's buffers (own buffers carry PCR_FRAME_PAD_ELEMS of slack)
int check_sliced(pcr_dist *d, Slice *mine)
{
    const size_t n = pcr_framebuffer_elems(d->ctx), cap = pcr_framebuffer_capacity(d->ctx);
    if (!n) return fail(PCR_E_ARG, "rank %d has no framebuffer", d->rank);
    const Slice s = slice_of(n, d->world, d->rank);
    if ((size_t)d->world * s.count > cap)
        return fail(PCR_E_ARG, "rank %d: %d slices of %zu words need %zu words, the framebuffers hold %zu (external buffers carry no pad)",
                    d->rank, d->world, s.count, (size_t)d->world * s.count, cap);
    *mine = s;
    return PCR_OK;
}

int reduce_scatter_u64(pcr_dist *d, void *buf, Slice s, ncclRedOp_t op)
{
    hipStream_t st = (hipStream_t)pcr_get_stream(d->ctx);
    if (hipSetDevice(pcr_get_device(d->ctx)) != hipSuccess) return fail(PCR_E_HIP, "hipSetDevice failed");
    // in place: recvbuff == sendbuff + rank * recvcount
    NCCL_TRY(ncclReduceScatter(buf, (uint64_t *)buf + s.first, s.count, ncclUint64, op, d->comm, st));
    return PCR_OK;
}

int reduce_u64(pcr_dist *d, void *buf, size_t count, ncclRedOp_t op, int root)
{
    hipStream_t st = (hipStream_t)pcr_get_stream(d->ctx);
    if (hipSetDevice(pcr_get_device(d->ctx)) != hipSuccess) return fail(PCR_E_HIP, "hipSetDevice failed");
    if (root == PCR_DIST_ALL) NCCL_TRY(ncclAllReduce(buf, buf, count, ncclUint64, op, d->comm, st));
    else                      NCCL_TRY(ncclReduce(buf, buf, count, ncclUint64, op, root, d->comm, st));
    return PCR_OK;
}
} // namespace

extern "C" {

const char *pcr_dist_last_error(void) { return g_err.c_str(); }

void pcr_dist_shard_range(int64_t units, int world, int rank, int64_t *first, int64_t *count)
{
    const int64_t base = units / world, rem = units % world;
    if (first) *first = rank * base + (rank < rem ? rank : rem);
    if (count) *count = base + (rank < rem ? 1 : 0);
}

int pcr_dist_unique_id(unsigned char id[PCR_DIST_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == PCR_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return PCR_OK;
}

int pcr_dist_create(pcr_ctx *ctx, const unsigned char id[PCR_DIST_ID_BYTES], int rank, int world, pcr_dist **out)
{
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return fail(PCR_E_ARG, "bad arguments");
    *out = nullptr;
    if (hipSetDevice(pcr_get_device(ctx)) != hipSuccess) return fail(PCR_E_HIP, "hipSetDevice failed");
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    pcr_dist *d = new pcr_dist();
    d->ctx = ctx; d->rank = rank; d->world = world;
    ncclResult_t r = ncclCommInitRank(&d->comm, world, u, rank);
    if (r != ncclSuccess) { delete d; return fail(PCR_E_HIP, "ncclCommInitRank failed: %s", ncclGetErrorString(r)); }
    *out = d;
    return PCR_OK;
}

int pcr_dist_create_local(pcr_ctx *const *ctxs, int n, pcr_dist **out)
{
    if (!ctxs || !out || n < 1) return fail(PCR_E_ARG, "bad arguments");
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(PCR_E_ARG, "context %d is NULL", i);
        devs[(size_t)i] = pcr_get_device(ctxs[i]);
        for (int j = 0; j < i; ++j)
            if (devs[(size_t)j] == devs[(size_t)i]) return fail(PCR_E_ARG, "contexts %d and %d share device %d: one rank per GPU", j, i, devs[(size_t)i]);
    }
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    ncclResult_t r = ncclCommInitAll(comms.data(), n, devs.data());
    bool ok = r == ncclSuccess;
    for (int i = 0; ok && i < n; ++i) {
        out[i] = new (std::nothrow) pcr_dist();
        if (!out[i]) { ok = false; break; }
        out[i]->ctx = ctxs[i]; out[i]->comm = comms[(size_t)i]; out[i]->rank = i; out[i]->world = n;
        comms[(size_t)i] = nullptr;             // owned by out[i] from here on
    }
    if (!ok) {                                  // nothing half-made is handed back
        for (int i = 0; i < n; ++i) { pcr_dist_destroy(out[i]); out[i] = nullptr; }
        for (ncclComm_t c : comms) if (c) (void)ncclCommDestroy(c);
        return r != ncclSuccess ? fail(PCR_E_HIP, "ncclCommInitAll failed: %s", ncclGetErrorString(r)) : fail(PCR_E_NOMEM, "out of host memory");
    }
    return PCR_OK;
}

void pcr_dist_destroy(pcr_dist *d)
{
    if (!d) return;
    if (d->scratch) {
        (void)hipSetDevice(pcr_get_device(d->ctx));
        (void)hipStreamSynchronize((hipStream_t)pcr_get_stream(d->ctx));
        (void)hipFree(d->scratch);
    }
    if (d->comm) (void)ncclCommDestroy(d->comm);
    delete d;
}

int pcr_dist_set_exchange(pcr_dist *d, int mode)
{
    if (!d) return fail(PCR_E_ARG, "dist is NULL");
    if (mode < PCR_DIST_EXCHANGE_AUTO || mode > PCR_DIST_EXCHANGE_SLICED_P2P) return fail(PCR_E_ARG, "unknown exchange %d", mode);
    d->exchange = mode;
    return PCR_OK;
}

int pcr_dist_exchange(const pcr_dist *d)
{
    if (!d) return PCR_DIST_EXCHANGE_REDUCE;
    if (d->exchange != PCR_DIST_EXCHANGE_AUTO) return d->exchange;
    // AUTO is REDUCE, whatever the frame's size: the sliced exchange (in-place reduce-scatter over the padded frame, gather with
    // sendbuff == recvbuff + rank * count, the all-to-all into scratch) has only ever run on a one-rank communicator, where every
    // collective is a self-copy (no box with peers yet: SCALE_r01..r03 skipped). It stays an explicit choice
    // (pcr_dist_set_exchange / --merge sliced) until tests/test_dist_native.py's >= 2-GPU branch has passed on hardware.
    return PCR_DIST_EXCHANGE_REDUCE;
}

void pcr_dist_slice_range(size_t elems, int world, int rank, size_t *first, size_t *count)
{
    const Slice s = slice_of(elems, world < 1 ? 1 : world, rank);
    if (first) *first = s.first;
    if (count) *count = s.count;
}

const void *pcr_dist_merged_slice(pcr_dist *d)
{
    if (!d) return nullptr;
    if (d->pending_min_slices) {
        // behind the all-to-all in stream order (which, inside a group, is only enqueued by pcr_dist_group_end: hence here)
        const size_t S = d->pending_min_slices;
        d->pending_min_slices = 0;
        if (pcr_merge_min_slices(d->ctx, d->scratch, d->world, S) != PCR_OK) { fail(PCR_E_HIP, "rank %d: pcr_merge_min_slices: %s", d->rank, pcr_last_error(d->ctx)); return nullptr; }
    }
    return d->merged_slice;
}

int pcr_dist_merge_min_sliced(pcr_dist *d)
{
    if (!d) return fail(PCR_E_ARG, "dist is NULL");
    Slice s;
    int rc = check_sliced(d, &s);
    if (rc) return rc;
    uint64_t *fb = (uint64_t *)pcr_device_framebuffer(d->ctx);
    d->pending_min_slices = 0;
    if (d->exchange != PCR_DIST_EXCHANGE_SLICED_P2P) {
        d->merged_slice = fb + s.first;
        return reduce_scatter_u64(d, fb, s, ncclMin);
    }
    // point to point: slice j of my frame goes to rank j over the link between us, everyone's copy of my slice comes back;
    // the min over the copies is a local kernel (pcr_merge_min_slices)
    hipStream_t st = (hipStream_t)pcr_get_stream(d->ctx);
    if (hipSetDevice(pcr_get_device(d->ctx)) != hipSuccess) return fail(PCR_E_HIP, "hipSetDevice failed");
    const size_t need = (size_t)d->world * s.count;
    if (d->scratch_elems < need) {
        if (d->scratch) { (void)hipStreamSynchronize(st); (void)hipFree(d->scratch); d->scratch = nullptr; d->scratch_elems = 0; }
        if (hipMalloc((void **)&d->scratch, need * 8) != hipSuccess) return fail(PCR_E_NOMEM, "rank %d: no memory for %zu scratch words", d->rank, need);
        d->scratch_elems = need;
    }
    NCCL_TRY(ncclAllToAll(fb, d->scratch, s.count, ncclUint64, d->comm, st));
    d->merged_slice = d->scratch;
    d->pending_min_slices = s.count;             // the local min is enqueued by pcr_dist_merged_slice
    return PCR_OK;
}

int pcr_dist_merge_sum_sliced(pcr_dist *d)
{
    if (!d) return fail(PCR_E_ARG, "dist is NULL");
    Slice s;
    int rc = check_sliced(d, &s);
    if (rc) return rc;
    void *rg = pcr_device_rg(d->ctx), *ba = pcr_device_ba(d->ctx);
    if (!rg || !ba) return fail(PCR_E_ARG, "rank %d has no accumulation buffers", d->rank);
    NCCL_TRY(ncclGroupStart());
    rc = reduce_scatter_u64(d, rg, s, ncclSum);
    if (rc == PCR_OK) rc = reduce_scatter_u64(d, ba, s, ncclSum);
    NCCL_TRY(ncclGroupEnd());
    return rc;
}

int pcr_dist_gather_image(pcr_dist *d, int root)
{
    if (!d) return fail(PCR_E_ARG, "dist is NULL");
    if (root != PCR_DIST_ALL && (root < 0 || root >= d->world)) return fail(PCR_E_ARG, "root %d out of range", root);
    Slice s;
    int rc = check_sliced(d, &s);
    if (rc) return rc;
    uint32_t *rgba = (uint32_t *)pcr_device_rgba(d->ctx);
    if (!rgba) return fail(PCR_E_ARG, "rank %d has no image buffer", d->rank);
    hipStream_t st = (hipStream_t)pcr_get_stream(d->ctx);
    if (hipSetDevice(pcr_get_device(d->ctx)) != hipSuccess) return fail(PCR_E_HIP, "hipSetDevice failed");
    // in place: sendbuff == recvbuff + rank * sendcount
    if (root == PCR_DIST_ALL) NCCL_TRY(ncclAllGather(rgba + s.first, rgba, s.count, ncclUint32, d->comm, st));
    else                      NCCL_TRY(ncclGather(rgba + s.first, rgba, s.count, ncclUint32, root, d->comm, st));
    return PCR_OK;
}

// pixels of the image that lie in this rank's slice
static size_t slice_pixels(const pcr_render_params *p, Slice s)
{
    const size_t px = (size_t)p->width * (size_t)p->height;
    return s.first < px ? (s.count < px - s.first ? s.count : px - s.first) : 0;
}

// merge + resolve + gather of a basic frame, sliced form (everything but the render and the clear)
static int basic_exchange_sliced(pcr_dist *d, const pcr_render_params *p, int root)
{
    int rc = pcr_dist_merge_min_sliced(d);
    if (rc) return rc;
    Slice s;
    if ((rc = check_sliced(d, &s))) return rc;
    const void *merged = pcr_dist_merged_slice(d);
    if (!merged) return PCR_E_HIP;
    PCR_TRY(d, pcr_resolve_basic_range(d->ctx, p, merged, slice_pixels(p, s), (uint32_t *)pcr_device_rgba(d->ctx) + s.first));
    return pcr_dist_gather_image(d, root);
}

int pcr_dist_rank(const pcr_dist *d) { return d ? d->rank : -1; }
int pcr_dist_world(const pcr_dist *d) { return d ? d->world : 0; }
int pcr_dist_comm_ranks(const pcr_dist *d)
{
    if (!d || !d->comm) return 0;
    int n = 0;
    return ncclCommCount(d->comm, &n) == ncclSuccess ? n : -1;
}

int pcr_dist_group_begin(void) { NCCL_TRY(ncclGroupStart()); return PCR_OK; }
int pcr_dist_group_end(void) { NCCL_TRY(ncclGroupEnd()); return PCR_OK; }

int pcr_dist_merge_min(pcr_dist *d, int root)
{
    if (!d) return fail(PCR_E_ARG, "dist is NULL");
    if (root != PCR_DIST_ALL && (root < 0 || root >= d->world)) return fail(PCR_E_ARG, "root %d out of range", root);
    void *fb = pcr_device_framebuffer(d->ctx);
    if (!fb) return fail(PCR_E_ARG, "rank %d has no framebuffer", d->rank);
    return reduce_u64(d, fb, pcr_framebuffer_elems(d->ctx), ncclMin, root);
}

int pcr_dist_merge_sum(pcr_dist *d, int root)
{
    if (!d) return fail(PCR_E_ARG, "dist is NULL");
    if (root != PCR_DIST_ALL && (root < 0 || root >= d->world)) return fail(PCR_E_ARG, "root %d out of range", root);
    void *rg = pcr_device_rg(d->ctx), *ba = pcr_device_ba(d->ctx);
    if (!rg || !ba) return fail(PCR_E_ARG, "rank %d has no accumulation buffers", d->rank);
    const size_t n = pcr_framebuffer_elems(d->ctx);
    // two collectives inside one group: RCCL fuses them into one launch
    NCCL_TRY(ncclGroupStart());
    int rc = reduce_u64(d, rg, n, ncclSum, root);
    if (rc == PCR_OK) rc = reduce_u64(d, ba, n, ncclSum, root);
    NCCL_TRY(ncclGroupEnd());
    return rc;
}

int pcr_dist_frame_basic(pcr_dist *d, const pcr_render_params *p, int root)
{
    if (!d || !p) return fail(PCR_E_ARG, "bad arguments");
    PCR_TRY(d, pcr_frame_begin(d->ctx, p, PCR_METHOD_BASIC));      // CLEAR + cull/LOD prepass
    PCR_TRY(d, pcr_render_basic(d->ctx, p));
    if (pcr_dist_exchange(d) != PCR_DIST_EXCHANGE_REDUCE) return basic_exchange_sliced(d, p, root);
    int rc = pcr_dist_merge_min(d, root);
    if (rc) return rc;
    if (root == PCR_DIST_ALL || root == d->rank) PCR_TRY(d, pcr_resolve_basic(d->ctx, p));
    return PCR_OK;
}

// Steady-state form of pcr_dist_frame_basic: the frame's CLEAR + prepass were done by the previous step (or by one
// pcr_frame_begin before the first), so a step is shard render + merge + ONE launch that resolves the merged frame where it
// ended up, clears, and runs the next frame's prepass (pcr_frame_turn); ranks that hold no result only clear + prepass.
int pcr_dist_step_basic(pcr_dist *d, const pcr_render_params *p, int root)
{
    if (!d || !p) return fail(PCR_E_ARG, "bad arguments");
    PCR_TRY(d, pcr_render_basic(d->ctx, p));
    if (pcr_dist_exchange(d) != PCR_DIST_EXCHANGE_REDUCE) {
        // the image is resolved slice by slice on the ranks that own the slices; what is left of the frame's end is CLEAR + prepass
        int rcs = basic_exchange_sliced(d, p, root);
        if (rcs) return rcs;
        PCR_TRY(d, pcr_frame_begin(d->ctx, p, PCR_METHOD_BASIC));
        return PCR_OK;
    }
    int rc = pcr_dist_merge_min(d, root);
    if (rc) return rc;
    if (root == PCR_DIST_ALL || root == d->rank) PCR_TRY(d, pcr_frame_turn(d->ctx, p, p, PCR_METHOD_BASIC));
    else                                         PCR_TRY(d, pcr_frame_begin(d->ctx, p, PCR_METHOD_BASIC));
    return PCR_OK;
}

int pcr_dist_frame_hqs(pcr_dist *d, const pcr_render_params *p, int root)
{
    if (!d || !p) return fail(PCR_E_ARG, "bad arguments");
    PCR_TRY(d, pcr_frame_begin(d->ctx, p, PCR_METHOD_HQS));
    PCR_TRY(d, pcr_render_hqs_depth(d->ctx, p));
    int rc = pcr_dist_merge_min(d, PCR_DIST_ALL);                 // every rank needs the global depth for its 1 % test
    if (rc) return rc;
    PCR_TRY(d, pcr_render_hqs_color(d->ctx, p));
    if (pcr_dist_exchange(d) != PCR_DIST_EXCHANGE_REDUCE) {
        // sums reduce-scattered: every rank divides the pixels of its slice (it holds the global depth words already)
        if ((rc = pcr_dist_merge_sum_sliced(d))) return rc;
        Slice s;
        if ((rc = check_sliced(d, &s))) return rc;
        PCR_TRY(d, pcr_resolve_hqs_range(d->ctx, p, (const uint64_t *)pcr_device_framebuffer(d->ctx) + s.first,
                                         (const uint64_t *)pcr_device_rg(d->ctx) + s.first, (const uint64_t *)pcr_device_ba(d->ctx) + s.first,
                                         slice_pixels(p, s), (uint32_t *)pcr_device_rgba(d->ctx) + s.first));
        return pcr_dist_gather_image(d, root);
    }
    rc = pcr_dist_merge_sum(d, root);
    if (rc) return rc;
    if (root == PCR_DIST_ALL || root == d->rank) PCR_TRY(d, pcr_resolve_hqs(d->ctx, p));
    return PCR_OK;
}

} // extern "C"
