"""Multi-GPU layer: one process per GPU, static contiguous batch shards, and the one exchange step the
path has — merging the partial framebuffers. No reference counterpart exists (the reference
is single-GPU: src/main.cpp:61); correctness rests on `min` / `+` being associative and commutative on
the packed 64-bit words, which makes the merged result bit-identical to a single-GPU render.

    basic :  render shard -> all-reduce MIN over the u64 framebuffer
    HQS   :  depth pass  -> all-reduce MIN (every rank then tests against the GLOBAL depth)
             colour pass -> all-reduce SUM over RG and BA -> resolve

torch.distributed is the transport (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU
tests). It exposes signed int64 only, so the u64 min is computed as a signed min on sign-flipped words
(x ^ 1<<63 is an order isomorphism u64 -> i64); the flips run in libpcr_hip.so on the GPU path.
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


def allreduce_sum_u64_numpy(acc: np.ndarray, group=None) -> np.ndarray:
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(acc.view(np.int64))       # two's complement add == u64 add
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return acc


class DeviceFrame:
    """Framebuffers owned by torch (so RCCL can reduce them in place) and lent to a pcr context."""

    def __init__(self, ctx, width: int, height: int, device):
        import torch
        from ._native import fb_elems
        n = fb_elems(width, height)
        self.ctx = ctx
        self.device = device
        self.fb = torch.empty(n, dtype=torch.int64, device=device)
        self.rg = torch.zeros(n, dtype=torch.int64, device=device)
        self.ba = torch.zeros(n, dtype=torch.int64, device=device)

    def bind(self, stream=None):
        """Make this frame the context's render target; pcr work goes to `stream` (default: torch's current)."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        self.ctx.set_stream(s.cuda_stream)
        self.ctx.use_external_buffers(self.fb.data_ptr(), self.rg.data_ptr(), self.ba.data_ptr())

    def allreduce_min(self, group=None):
        import torch.distributed as dist
        self.ctx.flip_sign()
        dist.all_reduce(self.fb, op=dist.ReduceOp.MIN, group=group)
        self.ctx.flip_sign()

    def allreduce_sum(self, group=None):
        import torch.distributed as dist
        dist.all_reduce(self.rg, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(self.ba, op=dist.ReduceOp.SUM, group=group)

    def release(self):
        self.ctx.synchronize()
        self.ctx.use_external_buffers(0, 0, 0)
        self.ctx.set_stream(0)


def render_basic_sharded(ctx, frame: Optional[DeviceFrame], params, world_size: int, group=None):
    """One frame of the basic method on this rank's shard + the merge, all on one stream. Enqueue only."""
    ctx.clear()
    ctx.render_basic(params)
    if frame is not None:
        frame.allreduce_min(group)
    ctx.resolve_basic(params)


def render_hqs_sharded(ctx, frame: Optional[DeviceFrame], params, world_size: int, group=None):
    ctx.clear()
    ctx.render_hqs_depth(params)
    if frame is not None:
        frame.allreduce_min(group)          # global depth before the 1 % test
    ctx.render_hqs_color(params)
    if frame is not None:
        frame.allreduce_sum(group)
    ctx.resolve_hqs(params)


class PipelinedBasicRenderer:
    """Basic method over shards with the exchange step overlapped: frame k is merged (sign flip, RCCL min
    all-reduce, flip back, resolve) on a communication stream while frame k+1 is already being decoded and
    rasterized into the other of two framebuffers on the compute stream. HIP events order the two streams; results
    are the same as the one-stream form, one frame later."""

    def __init__(self, ctx, width: int, height: int, device, group=None):
        import torch
        self.ctx, self.device, self.group = ctx, device, group
        self.frames = [DeviceFrame(ctx, width, height, device), DeviceFrame(ctx, width, height, device)]
        self.compute = torch.cuda.Stream(device)
        self.comm = torch.cuda.Stream(device)
        self.rendered = [torch.cuda.Event(), torch.cuda.Event()]
        self.merged = [torch.cuda.Event(), torch.cuda.Event()]
        self.k = 0
        for e in self.merged:
            e.record(self.comm)

    def step(self, params):
        import torch
        i = self.k & 1
        f = self.frames[i]
        self.compute.wait_event(self.merged[i])        # this framebuffer's previous merge + resolve are done
        f.bind(self.compute)
        self.ctx.clear()
        self.ctx.render_basic(params)
        self.rendered[i].record(self.compute)
        self.comm.wait_event(self.rendered[i])
        with torch.cuda.stream(self.comm):             # RCCL orders itself against torch's current stream
            f.bind(self.comm)
            f.allreduce_min(self.group)
            self.ctx.resolve_basic(params)
            self.merged[i].record(self.comm)
        self.k += 1

    def last_frame(self) -> DeviceFrame:
        return self.frames[(self.k - 1) & 1]

    def finish(self):
        self.compute.synchronize()
        self.comm.synchronize()

    def release(self):
        self.finish()
        self.frames[0].release()
