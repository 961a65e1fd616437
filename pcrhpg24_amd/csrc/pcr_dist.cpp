// pcr_dist.cpp — implementation of include/pcr_dist.h (libpcr_dist.so): batch-sharded rendering over the GPUs of a node,
// partial framebuffers merged with RCCL's native unsigned 64-bit min / sum, in place, on each context's own stream.
// No reference counterpart (the reference is single-GPU); this is north_star's multi-GPU step on the C++ side.
#include "pcr_dist.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

struct pcr_dist {
    pcr_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
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
    std::vector<ncclComm_t> comms((size_t)n);
    NCCL_TRY(ncclCommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) {
        out[i] = new pcr_dist();
        out[i]->ctx = ctxs[i]; out[i]->comm = comms[(size_t)i]; out[i]->rank = i; out[i]->world = n;
    }
    return PCR_OK;
}

void pcr_dist_destroy(pcr_dist *d)
{
    if (!d) return;
    if (d->comm) (void)ncclCommDestroy(d->comm);
    delete d;
}

int pcr_dist_rank(const pcr_dist *d) { return d ? d->rank : -1; }
int pcr_dist_world(const pcr_dist *d) { return d ? d->world : 0; }

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
    rc = pcr_dist_merge_sum(d, root);
    if (rc) return rc;
    if (root == PCR_DIST_ALL || root == d->rank) PCR_TRY(d, pcr_resolve_hqs(d->ctx, p));
    return PCR_OK;
}

} // extern "C"
