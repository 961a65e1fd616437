"""Multi-GPU layer: one process per GPU, static contiguous batch shards, and the one exchange step the
path has — merging the partial framebuffers. No reference counterpart exists (the reference
is single-GPU: src/main.cpp:61); correctness rests on `min` / `+` being associative and commutative on
the packed 64-bit words, which makes the merged result bit-identical to a single-GPU render.

    basic :  render shard -> reduce MIN of the u64 framebuffer to the display rank (0) -> resolve there
    HQS   :  depth pass  -> all-reduce MIN (every rank then tests against the GLOBAL depth)
             colour pass -> reduce SUM of RG|BA (one buffer) to the display rank -> resolve there
A reduce moves half the bytes of an all-reduce over the point-to-point xGMI links; `merge="allreduce"` keeps the
finished frame on every rank instead; `merge="sliced"` cuts the frame into N slices (all-to-all, local min, resolve of the
own slice, all-gather of the image): 1/N of the frame per link. The C++ layer (include/pcr_dist.h, NativeDist below) has the
same three forms on RCCL's own unsigned 64-bit min; this module is the torch.distributed transport of the tests and of
bench.py's default at N > 1, and the gloo-runnable statement of the exchange arithmetic (tests/test_dist_cpu.py).

torch.distributed is the transport (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU
tests). It exposes signed int64 only. On the GPU path the framebuffer is made int64-mergeable instead of being
sign-flipped around every collective: empty pixels are cleared to INT64_MAX rather than all-ones
(`pcr_set_int64_mergeable`), every key a point produces has a clear top bit, so signed and unsigned order agree. The
numpy forms used by the gloo tests take arbitrary u64 words and flip the sign bit (x ^ 1<<63 is an order isomorphism
u64 -> i64).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

SIGN = np.uint64(0x8000000000000000)


def shard_range(num_units: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous [first, first+count) of `num_units` for `rank`; sizes differ by at most one."""
    base, rem = divmod(num_units, world_size)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def allreduce_min_u64_numpy(fb: np.ndarray, group=None) -> np.ndarray:
    """CPU path (gloo): in-place u64 min all-reduce of a numpy framebuffer. Used by the CPU tests."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy((fb ^ SIGN).view(np.int64))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    fb[:] = t.numpy().view(np.uint64) ^ SIGN
    return fb


def reduce_min_u64_numpy(fb: np.ndarray, dst: int = 0, group=None) -> np.ndarray:
    """CPU path (gloo): u64 min reduce to rank `dst`; other ranks' buffers are left unspecified (as NCCL does)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy((fb ^ SIGN).view(np.int64))
    dist.reduce(t, dst=dst, op=dist.ReduceOp.MIN, group=group)
    fb[:] = t.numpy().view(np.uint64) ^ SIGN
    return fb


def reduce_sum_u64_numpy(acc: np.ndarray, dst: int = 0, group=None) -> np.ndarray:
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(acc.view(np.int64))
    dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return acc


def allreduce_sum_u64_numpy(acc: np.ndarray, group=None) -> np.ndarray:
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(acc.view(np.int64))       # two's complement add == u64 add
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return acc


def exchange_shard_heads(head_enc: np.ndarray, head_sep: np.ndarray, device, group=None):
    """Every rank contributes the first words of ITS first batch (HuffmanFile.head_words(first)); returns the head words
    of the rank that follows this one in the global stream — what pcr_upload_tail needs so that the reference's tail
    over-reads see the same bytes as on one GPU — or None on the last rank. One small all_gather."""
    import torch
    import torch.distributed as dist
    from .host import ENCODED_PAD_WORDS, SEPARATE_PAD_WORDS
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ne, ns = min(len(head_enc), ENCODED_PAD_WORDS), min(len(head_sep), SEPARATE_PAD_WORDS)
    head = np.zeros(2 + ENCODED_PAD_WORDS + SEPARATE_PAD_WORDS, np.int32)
    head[0], head[1] = ne, ns
    head[2:2 + ne] = np.asarray(head_enc[:ne], np.uint32).view(np.int32)
    head[2 + ENCODED_PAD_WORDS:2 + ENCODED_PAD_WORDS + ns] = np.asarray(head_sep[:ns], np.int32)
    mine = torch.from_numpy(head).to(device)
    heads = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(heads, mine, group=group)
    if rank + 1 >= world:
        return None
    nx = heads[rank + 1].cpu().numpy()
    return (nx[2:2 + nx[0]].copy().view(np.uint32), nx[2 + ENCODED_PAD_WORDS:2 + ENCODED_PAD_WORDS + nx[1]].copy())


def _shared_stream(frame, stream):
    """The ONE stream the context's kernels and torch's collectives are enqueued on. torch's default stream has the handle 0,
    which pcr_set_stream reads as "the context's own stream" -- a non-blocking stream nothing orders against torch's: the
    collective then starts while the frame is still being drawn (found in round 3: a warmed-up RCCL launches fast enough to
    show it). So if torch's current stream is the default one, a stream of the frame's own becomes torch's current stream
    until release(). NOTE: that switch is process-wide (torch.cuda.set_stream) -- torch work the caller enqueues between bind()
    and release() lands on the frame's stream, ordered with the frame; pass an explicit non-default `stream` to avoid it.
    Idempotent: binding a frame that is already bound keeps the stream saved by the first bind."""
    import torch
    s = stream if stream is not None else torch.cuda.current_stream(frame.device)
    own = getattr(frame, "_own_stream", None)
    if own is not None and s.cuda_stream == own.cuda_stream:
        return own                                   # bound before: torch's current stream already is the frame's own
    if s.cuda_stream == 0:
        if own is None:
            own = frame._own_stream = torch.cuda.Stream(frame.device)
        own.wait_stream(s)                           # the frame's tensors were filled on the default stream
        if getattr(frame, "_prev_stream", None) is None:
            frame._prev_stream = s
        torch.cuda.set_stream(own)
        s = own
    return s


def _restore_stream(frame):
    import torch
    prev = getattr(frame, "_prev_stream", None)
    if prev is not None:
        prev.wait_stream(frame._own_stream)
        torch.cuda.set_stream(prev)
        frame._prev_stream = None


class DeviceFrame:
    """Framebuffers owned by torch (so RCCL can reduce them in place) and lent to a pcr context."""

    def __init__(self, ctx, width: int, height: int, device, accum: bool = True):
        import torch
        from ._native import fb_elems
        n = fb_elems(width, height)
        n2 = (n + 1) & ~1                          # keeps BA 16-byte aligned behind RG
        self.ctx = ctx
        self.device = device
        self.fb = torch.empty(n, dtype=torch.int64, device=device)
        # RG | BA: one collective for both. A frame of the basic method has none (accum=False): the context keeps its own,
        # which pcr_clear then does not have to zero again every time the frame is bound
        self.acc = torch.zeros(2 * n2, dtype=torch.int64, device=device) if accum else None
        self.rg = self.acc[:n] if accum else None
        self.ba = self.acc[n2:n2 + n] if accum else None

    def bind(self, stream=None):
        """Make this frame the context's render target; pcr work goes to `stream` (default: torch's current)."""
        s = _shared_stream(self, stream)
        self.ctx.set_stream(s.cuda_stream)
        self.ctx.use_external_buffers(self.fb.data_ptr(), self.rg.data_ptr() if self.rg is not None else 0,
                                      self.ba.data_ptr() if self.ba is not None else 0)
        # torch's NCCL/RCCL binding has no uint64: the frame is reduced as int64, which orders it correctly once empty
        # pixels are INT64_MAX instead of all-ones (every real key has a clear top bit) -- no sign-flip passes
        self.ctx.set_int64_mergeable(True)

    def allreduce_min(self, group=None):
        import torch.distributed as dist
        dist.all_reduce(self.fb, op=dist.ReduceOp.MIN, group=group)

    def allreduce_sum(self, group=None):
        import torch.distributed as dist
        dist.all_reduce(self.acc, op=dist.ReduceOp.SUM, group=group)

    def reduce_min(self, dst: int = 0, group=None) -> bool:
        """u64 min reduce to rank `dst`; returns whether this rank holds the merged frame afterwards."""
        import torch.distributed as dist
        dist.reduce(self.fb, dst=dst, op=dist.ReduceOp.MIN, group=group)
        return dist.get_rank(group) == dst

    def reduce_sum(self, dst: int = 0, group=None) -> bool:
        import torch.distributed as dist
        dist.reduce(self.acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
        return dist.get_rank(group) == dst

    def release(self):
        self.ctx.synchronize()
        self.ctx.use_external_buffers(0, 0, 0)
        self.ctx.set_int64_mergeable(False)
        self.ctx.set_stream(0)
        _restore_stream(self)


def slice_elems(n: int, world_size: int) -> int:
    """Words per slice when a frame of n words is cut into world_size slices (even, so every slice stays 16-byte aligned)."""
    return (-(-n // world_size) + 1) & ~1


def a2a_merge(fb, recv, rgba_slice, rgba, world_size: int, min_slices, resolve_range, group=None):
    """The all-to-all form of the basic method's merge, on torch tensors of either backend (RCCL on the GPU, gloo in the
    CPU tests). fb: this rank's frame, int64[world_size * S], empty pixels INT64_MAX, padding included. After the call
    recv[:S] holds the merged slice this rank owns, rgba_slice its resolved pixels and rgba (int32[world_size * S]) the
    whole image on every rank. Point-to-point xGMI carries 1/N of the frame per link instead of the whole frame around a
    ring; the element-wise work (min_slices, resolve_range) is the caller's: HIP kernels on the GPU path."""
    import torch.distributed as dist
    S = fb.numel() // world_size
    dist.all_to_all_single(recv, fb, group=group)           # recv[r*S:(r+1)*S] = rank r's copy of my slice
    min_slices(recv, world_size, S)                         # -> recv[:S]
    resolve_range(recv, S, rgba_slice)
    dist.all_gather_into_tensor(rgba, rgba_slice, group=group)


class SlicedFrame:
    """A frame for merge="sliced": torch-owned like DeviceFrame, padded to world_size equal slices, plus the receive buffer
    and the image tensors of a2a_merge."""

    def __init__(self, ctx, width: int, height: int, device, world_size: int):
        import torch
        from ._native import fb_elems
        self.ctx, self.device, self.world = ctx, device, world_size
        self.n = fb_elems(width, height)
        self.S = slice_elems(self.n, world_size)
        big = torch.iinfo(torch.int64).max
        self.fb = torch.full((world_size * self.S,), big, dtype=torch.int64, device=device)   # the padding stays empty
        self.recv = torch.empty(world_size * self.S, dtype=torch.int64, device=device)
        self.rgba_slice = torch.empty(self.S, dtype=torch.int32, device=device)
        self.rgba = torch.empty(world_size * self.S, dtype=torch.int32, device=device)
        self.rg = self.ba = self.acc = None

    def bind(self, stream=None):
        s = _shared_stream(self, stream)
        self.ctx.set_stream(s.cuda_stream)
        self.ctx.use_external_buffers(self.fb.data_ptr(), 0, 0)
        self.ctx.set_int64_mergeable(True)

    def merge_and_resolve(self, params, group=None):
        """Enqueue on torch's current stream (which must be the stream the context is bound to)."""
        ctx = self.ctx
        a2a_merge(self.fb, self.recv, self.rgba_slice, self.rgba, self.world,
                  lambda t, ns, S: ctx.merge_min_slices(t.data_ptr(), ns, S),
                  lambda t, S, out: ctx.resolve_basic_range(params, t.data_ptr(), S, out.data_ptr()), group)

    def image(self):
        """RGBA8 words of the merged frame (valid on every rank once the stream has drained)."""
        return self.rgba[:self.n]

    def gather_merged_framebuffer(self, group=None):
        """u64 words of the merged frame, assembled from the slices the ranks own (tests / dumps; not part of a frame)."""
        import torch
        import torch.distributed as dist
        out = torch.empty_like(self.fb)
        dist.all_gather_into_tensor(out, self.recv[:self.S].contiguous(), group=group)
        return out[:self.n]

    def release(self):
        self.ctx.synchronize()
        self.ctx.use_external_buffers(0, 0, 0)
        self.ctx.set_int64_mergeable(False)
        self.ctx.set_stream(0)
        _restore_stream(self)


def render_basic_sharded(ctx, frame, params, world_size: int, group=None, merge: str = "reduce"):
    """One frame of the basic method on this rank's shard + the merge, all on one stream. Enqueue only.
    merge="reduce": the finished frame (and its resolve) live on rank 0; "allreduce": on every rank; "a2a" (frame is a
    SlicedFrame): every rank merges and resolves the slice it owns, the image is all-gathered."""
    ctx.frame_begin(params)
    ctx.render_basic(params)
    final = True
    if frame is not None:
        if merge == "sliced":
            frame.merge_and_resolve(params, group)
            return
        if merge == "reduce":
            final = frame.reduce_min(0, group)
        else:
            frame.allreduce_min(group)
    if final:
        ctx.resolve_basic(params)


def render_hqs_sharded(ctx, frame: Optional[DeviceFrame], params, world_size: int, group=None, merge: str = "reduce"):
    ctx.frame_begin(params, hqs=True)
    ctx.render_hqs_depth(params)
    if frame is not None:
        frame.allreduce_min(group)          # global depth before the 1 % test: every rank needs it
    ctx.render_hqs_color(params)
    final = True
    if frame is not None:
        if merge in ("reduce", "sliced"):      # the all-to-all form exists for the basic method only
            final = frame.reduce_sum(0, group)
        else:
            frame.allreduce_sum(group)
    if final:
        ctx.resolve_hqs(params)


class NativeDist:
    """include/pcr_dist.h through ctypes: the C++ multi-GPU layer (libpcr_dist.so, RCCL's own ncclUint64 min / sum on the
    context's framebuffers and stream). torch.distributed only carries the 128-byte communicator id from rank 0 to the other
    ranks -- the launcher's job, whatever launcher it is."""

    def __init__(self, ctx, rank: int, world: int, device, group=None):
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _native as N
        from . import build
        N.hip_lib()                                  # torch's HIP / RCCL copies first (one runtime per process)
        lib = C.CDLL(build.DIST_LIB)
        lib.pcr_dist_last_error.restype = C.c_char_p
        lib.pcr_dist_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.pcr_dist_destroy.argtypes = [C.c_void_p]
        lib.pcr_dist_destroy.restype = None
        for n in ("pcr_dist_merge_min", "pcr_dist_merge_sum"):
            getattr(lib, n).argtypes = [C.c_void_p, C.c_int]
        for n in ("pcr_dist_frame_basic", "pcr_dist_frame_hqs", "pcr_dist_step_basic"):
            getattr(lib, n).argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.pcr_dist_set_exchange.argtypes = [C.c_void_p, C.c_int]
        lib.pcr_dist_exchange.argtypes = [C.c_void_p]
        self.lib, self.ctx, self.rank, self.world, self._C = lib, ctx, rank, world, C
        ident = C.create_string_buffer(128)
        if rank == 0 and lib.pcr_dist_unique_id(ident) != 0:
            raise RuntimeError("pcr_dist_unique_id: " + lib.pcr_dist_last_error().decode())
        t = torch.frombuffer(bytearray(ident.raw), dtype=torch.uint8).to(device)
        if world > 1:
            dist.broadcast(t, 0, group=group)
        ident = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
        self.h = C.c_void_p()
        if lib.pcr_dist_create(ctx.h, ident, rank, world, C.byref(self.h)) != 0:
            raise RuntimeError("pcr_dist_create: " + lib.pcr_dist_last_error().decode())

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(what + ": " + self.lib.pcr_dist_last_error().decode())

    def frame_basic(self, params, root: int = 0):
        """clear + prepass + shard render + u64 min merge (root, or -1 for every rank) + resolve where the frame ends up."""
        self._chk(self.lib.pcr_dist_frame_basic(self.h, self._C.byref(params), root), "pcr_dist_frame_basic")

    def frame_hqs(self, params, root: int = 0):
        self._chk(self.lib.pcr_dist_frame_hqs(self.h, self._C.byref(params), root), "pcr_dist_frame_hqs")

    def step_basic(self, params, root: int = 0):
        """render + merge + (resolve, clear, next prepass in one launch); prime with ctx.frame_begin(params) once."""
        self._chk(self.lib.pcr_dist_step_basic(self.h, self._C.byref(params), root), "pcr_dist_step_basic")

    def comm_ranks(self) -> int:
        """Ranks of the communicator as RCCL reports them (ncclCommCount)."""
        self.lib.pcr_dist_comm_ranks.argtypes = [self._C.c_void_p]
        return int(self.lib.pcr_dist_comm_ranks(self.h))

    EXCHANGE = {"auto": 0, "reduce": 1, "sliced": 2, "sliced_p2p": 3}

    def set_exchange(self, mode: str) -> str:
        """Which exchange the frame calls use (include/pcr_dist.h: PCR_DIST_EXCHANGE_*); returns what it resolves to for the
        context's image size ("auto" = reduce until the sliced forms have met a peer on hardware)."""
        self._chk(self.lib.pcr_dist_set_exchange(self.h, self.EXCHANGE[mode]), "pcr_dist_set_exchange")
        got = self.lib.pcr_dist_exchange(self.h)
        return {v: k for k, v in self.EXCHANGE.items()}[got]

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.pcr_dist_destroy(self.h)
            self.h = self._C.c_void_p()
