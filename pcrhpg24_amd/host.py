"""Host-side mirror of the reference's plugin surface for the Huffman path, over the C ABI.

The names, call order and error behaviour follow the reference so the parity tests read like a session
of the reference viewer without a window:

    Method   {name, description, group, update(renderer), render(renderer)}     include/Method.h:10-23
    Resource {state, load(renderer), unload(renderer), process(renderer)}        modules/compute/Resources.h:27-35
    HuffmanLasData.create(path)                                                  modules/compute/HuffmanLasLoader.h:87-92
    HuffmanMemIter  ("huffman_mem_iter_cuda")                                    modules/huffman_mem_iter_cuda/huffman_mem_iter_cuda.h
    HuffmanHQS      ("huffman_hqs")                                              modules/huffman_hqs/huffman_hqs.h
    ComputeLasData.create(path) / ComputeLoopLasCUDA ("loop_las_cuda")           modules/compute/ComputeLasLoader.{h,cpp},
                                                                                 modules/compute_loop_las_cuda/compute_loop_las_cuda.h
    Runtime.addMethod / setSelectedMethod / resource                             include/Runtime.h:15-55
    Debug.LOD / frustumCullingEnabled / colorizeChunks / showNumPoints           include/Debug.h:14-31

Python here is plumbing only (ctypes calls, byte slicing); all compute is in libpcr_hip.so /
libpcr_host.so. A C++ twin of these adapters lives in csrc/pcr_methods.hpp.
"""
from __future__ import annotations

import ctypes as C
import mmap
import os
import struct
from typing import Iterator, Optional

import numpy as np

from . import _native as N
from ._native import EncodeStats, FileHeader, LasInfo, RenderParams, RenderStats, XyzBatch, fb_elems

POINTS_PER_BATCH = 65536
ENCODED_PAD_WORDS = 1024
SEPARATE_PAD_WORDS = 256
BATCH_FIXED_HEADER = 124
_BATCH_FIXED = BATCH_FIXED_HEADER + 4 * (3072 + 1024 + 4096 + 4096 + 32)
PCR_COLOR_BYTES = 32768             # BC1: 8 bytes per 16 points


class PcrError(RuntimeError):
    pass


# --------------------------------------------------------------------------------------------------
# native buffers / encoder front-ends (libpcr_host.so)
# --------------------------------------------------------------------------------------------------
class NativeBytes:
    """A malloc'ed byte buffer owned by libpcr_host.so, exposed through the buffer protocol."""

    def __init__(self, ptr: int, length: int):
        self._ptr, self._len = ptr, length
        self._arr = (C.c_uint8 * length).from_address(ptr)

    def __len__(self) -> int:
        return self._len

    def view(self) -> memoryview:
        return memoryview(self._arr).cast("B")

    def free(self) -> None:
        if self._ptr:
            self._arr = None
            N.host_lib().pcr_host_free(C.c_void_p(self._ptr))
            self._ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def synth_encode(total_points: int, seed: int = 0x5EED, first: int = 0, count: Optional[int] = None,
                 chunk_points: int = 0, nthreads: int = 0) -> tuple[NativeBytes, dict]:
    """Generate + encode points [first, first+count) of the synthetic scene -> (.huffman image, stats)."""
    lib = N.host_lib()
    if count is None:
        count = total_points - first
    out, ln, st = C.c_void_p(), C.c_size_t(), EncodeStats()
    rc = lib.pcr_synth_encode(total_points, seed, first, count, chunk_points, nthreads, C.byref(out), C.byref(ln), C.byref(st))
    if rc:
        raise PcrError(f"pcr_synth_encode: {N.host_error()}")
    return NativeBytes(out.value, ln.value), st.as_dict()


def kernel_version() -> str:
    """Version tag of the render/transcode kernels in libpcr_hip.so (pcr_kernel_version): stored measurements name theirs."""
    return N.hip_lib().pcr_kernel_version().decode()


def synth_points(total_points: int, seed: int, first: int, count: int):
    x = np.empty(count, np.int32); y = np.empty(count, np.int32); z = np.empty(count, np.int32)
    c = np.empty(count, np.uint32)
    rc = N.host_lib().pcr_synth_points(total_points, seed, first, count, x.ctypes.data, y.ctypes.data, z.ctypes.data, c.ctypes.data)
    if rc:
        raise PcrError(f"pcr_synth_points: {N.host_error()}")
    return x, y, z, c


def synth_las_info(total_points: int, seed: int = 0x5EED) -> LasInfo:
    las = LasInfo()
    N.host_lib().pcr_synth_las_info(total_points, seed, C.byref(las))
    return las


def encode_points(x, y, z, color, las: LasInfo, morton_sort: bool = True, chunk_points: int = 0,
                  nthreads: int = 0, pad_tails: bool = False, bc7: bool = False) -> tuple[NativeBytes, dict]:
    """`preprocess in.las out.huffman <sort>` on in-memory points (src/preprocess.cpp:1167-1279). pad_tails: the
    PCR_ENCODE_PAD_TAILS variant (not in the reference) whose streams decode without the tail artefact. bc7: colours as BC7
    mode-6 blocks, the file of a reference built with COLOR_COMPRESSION == 7 (PCR_ENCODE_BC7)."""
    x = np.ascontiguousarray(x, np.int32); y = np.ascontiguousarray(y, np.int32); z = np.ascontiguousarray(z, np.int32)
    color = np.ascontiguousarray(color, np.uint32)
    out, ln, st = C.c_void_p(), C.c_size_t(), EncodeStats()
    rc = N.host_lib().pcr_encode_points(x.ctypes.data, y.ctypes.data, z.ctypes.data, color.ctypes.data, len(x), C.byref(las),
                                        int(bool(morton_sort)) | (2 if pad_tails else 0) | (4 if bc7 else 0), chunk_points, nthreads,
                                        C.byref(out), C.byref(ln), C.byref(st))
    if rc:
        raise PcrError(f"pcr_encode_points: {N.host_error()}")
    return NativeBytes(out.value, ln.value), st.as_dict()


def camera_orbit(yaw: float, pitch: float, radius: float, target, width: int, height: int,
                 fovy: float = 60.0, near: float = 0.1, far: float = 200000.0) -> RenderParams:
    """OrbitControls + Camera + the ChangingRenderData setup of HuffmanHQS::render (huffman_hqs.h:157-183)."""
    p = RenderParams()
    t = (C.c_double * 3)(*target)
    rc = N.host_lib().pcr_camera_orbit(yaw, pitch, radius, t, width, height, fovy, near, far, C.byref(p))
    if rc:
        raise PcrError(f"pcr_camera_orbit: {N.host_error()}")
    return p


# --------------------------------------------------------------------------------------------------
# .huffman container (HuffmanLasData::loadHeader, HuffmanLasLoader.h:57-85)
# --------------------------------------------------------------------------------------------------
def las_quantize(x, y, z, color, las: LasInfo, nthreads: int = 0):
    """10-10-10 three-level quantisation of points in input order (pcr_las_quantize): returns
    (batches[nB] XyzBatch array, xyz12, xyz8, xyz4, rgba) with nB*65536 slots each."""
    x, y, z = (np.ascontiguousarray(a, np.int32) for a in (x, y, z))
    color = np.ascontiguousarray(color, np.uint32)
    n = len(x)
    if not (len(y) == len(z) == len(color) == n) or n == 0:
        raise ValueError("x, y, z, color must be non-empty and of equal length")
    nb = (n + POINTS_PER_BATCH - 1) // POINTS_PER_BATCH
    batches = (XyzBatch * nb)()
    arrs = [np.empty(nb * POINTS_PER_BATCH, np.uint32) for _ in range(4)]
    rc = N.host_lib().pcr_las_quantize(x.ctypes.data, y.ctypes.data, z.ctypes.data, color.ctypes.data, n, C.byref(las),
                                       C.addressof(batches), *(a.ctypes.data for a in arrs), nthreads)
    if rc:
        raise PcrError(f"pcr_las_quantize failed: {N.host_error()}")
    return (batches, *arrs)


def read_las(path: str):
    """Minimal LAS 1.x reader for the loaders (header fields and point layout as ComputeLasData::loadHeader reads
    them, modules/compute/ComputeLasLoader.h:55-95, and getPoint, computeLasLoader.cs:147-190): returns
    (x, y, z int32, color 0x00BBGGRR uint32, LasInfo)."""
    with open(path, "rb") as f:
        hdr = f.read(375)
        if len(hdr) < 227 or hdr[:4] != b"LASF":
            raise PcrError(f"{path}: not a LAS file")
        major, minor = hdr[24], hdr[25]
        n = struct.unpack_from("<I", hdr, 107)[0] if (major == 1 and minor < 4) else struct.unpack_from("<Q", hdr, 247)[0]
        n = min(n, 1_000_000_000)                                     # ComputeLasLoader.h:69
        off, fmt, bpp = struct.unpack_from("<I", hdr, 96)[0], hdr[104], struct.unpack_from("<H", hdr, 105)[0]
        las = LasInfo()
        sc = struct.unpack_from("<6d", hdr, 131)
        las.scale[:], las.offset[:] = sc[:3], sc[3:]
        mx_x, mn_x, mx_y, mn_y, mx_z, mn_z = struct.unpack_from("<6d", hdr, 179)
        las.min[:], las.max[:] = (mn_x, mn_y, mn_z), (mx_x, mx_y, mx_z)
        f.seek(off)
        raw = np.frombuffer(f.read(n * bpp), np.uint8)
    if len(raw) != n * bpp:
        raise PcrError(f"{path}: truncated point data")
    raw = raw.reshape(n, bpp)
    xyz = np.ascontiguousarray(raw[:, :12]).view("<i4")
    off_rgb = {2: 20, 3: 28, 7: 30, 8: 30}.get(fmt % 128, 0)         # computeLasLoader.cs:151-160 (0: reads X's bytes)
    rgb = np.ascontiguousarray(raw[:, off_rgb:off_rgb + 6]).view("<u2").astype(np.uint32)
    rgb = np.where(rgb > 255, rgb // 256, rgb)                       # :170-172
    color = rgb[:, 0] | (rgb[:, 1] << 8) | (rgb[:, 2] << 16)
    return (np.ascontiguousarray(xyz[:, 0]), np.ascontiguousarray(xyz[:, 1]), np.ascontiguousarray(xyz[:, 2]),
            color.astype(np.uint32), las)


class HuffmanFile:
    """Header + batch-record slicing of a .huffman image held in memory or memory-mapped from disk."""

    def __init__(self, data):
        self._mm = None
        if isinstance(data, (str, os.PathLike)):
            f = open(data, "rb")
            self._mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            f.close()
            data = self._mm
        elif isinstance(data, NativeBytes):
            self._keep = data
            data = data.view()
        self.buf = memoryview(data).cast("B")
        if len(self.buf) < 40:
            raise PcrError("file shorter than its 40-byte header")
        (self.numPoints, self.numBatches, self.encodedBytes, self.separateBytes, self.clusterBytes) = \
            struct.unpack_from("<5q", self.buf, 0)
        if self.numBatches < 0 or len(self.buf) < 40 + 8 * self.numBatches:
            raise PcrError("file shorter than its batch size table")
        self.batch_data_sizes = np.frombuffer(self.buf, np.int64, self.numBatches, 40)
        if self.numBatches and int(self.batch_data_sizes.min()) < _BATCH_FIXED + PCR_COLOR_BYTES:
            raise PcrError("a batch record is shorter than its fixed part (size table entry %d)" % int(self.batch_data_sizes.min()))
        self.offsetToBatchData = 40 + 8 * self.numBatches
        self.batch_offsets = self.offsetToBatchData + np.concatenate([[0], np.cumsum(self.batch_data_sizes)])
        if int(self.batch_offsets[-1]) > len(self.buf):
            raise PcrError("batch records exceed the file")

    def header(self, first: int = 0, count: Optional[int] = None) -> FileHeader:
        """Header of the whole file, or of the sub-stream of batches [first, first+count)."""
        if count is None:
            count = self.numBatches - first
        if first == 0 and count == self.numBatches:
            return FileHeader(self.numPoints, self.numBatches, self.encodedBytes, self.separateBytes, self.clusterBytes)
        enc = sep = 0
        for b in range(first, first + count):
            ne, ns = self.stream_lengths(b)
            enc += 4 * ne; sep += 4 * ns
        return FileHeader(count * POINTS_PER_BATCH, count, enc, sep, 128 * count)

    def blob(self, b: int) -> memoryview:
        return self.buf[int(self.batch_offsets[b]):int(self.batch_offsets[b + 1])]

    def stream_lengths(self, b: int) -> tuple[int, int]:
        """(#encoded words, #escape words) of batch b, read from its inclusive prefixes."""
        o = int(self.batch_offsets[b])
        ns = struct.unpack_from("<i", self.buf, o + BATCH_FIXED_HEADER + 4 * 3072 + 4 * 1023)[0]
        ne = struct.unpack_from("<i", self.buf, o + BATCH_FIXED_HEADER + 4 * (3072 + 1024 + 4096 + 4096) + 4 * 31)[0]
        return ne, ns

    def head_words(self, b: int) -> tuple[np.ndarray, np.ndarray]:
        """First words of batch b's encoded / escape streams: the tail a shard ending before b must carry."""
        ne, ns = self.stream_lengths(b)
        o = int(self.batch_offsets[b]) + _BATCH_FIXED
        enc = np.frombuffer(self.buf, np.uint32, min(ne, ENCODED_PAD_WORDS), o)
        sep = np.frombuffer(self.buf, np.int32, min(ns, SEPARATE_PAD_WORDS), o + 4 * ne)
        return enc, sep

    def blobs(self, first: int = 0, count: Optional[int] = None) -> Iterator[memoryview]:
        if count is None:
            count = self.numBatches - first
        for b in range(first, first + count):
            yield self.blob(b)


# --------------------------------------------------------------------------------------------------
# C-ABI context wrapper
# --------------------------------------------------------------------------------------------------
class Context:
    """Owns one pcr_ctx (one GPU, one stream)."""

    def __init__(self, device: int = 0):
        self.lib = N.hip_lib()
        h = C.c_void_p()
        rc = self.lib.pcr_create(device, C.byref(h))
        if rc:
            raise PcrError(f"pcr_create({device}) -> {rc}: {(self.lib.pcr_last_error(None) or b'').decode()}")
        self.h = h
        self.device = device

    def _chk(self, rc: int, what: str) -> None:
        if rc:
            raise PcrError(f"{what} -> {rc}: {(self.lib.pcr_last_error(self.h) or b'').decode()}")

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.pcr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # resource side
    def stream_begin(self, hdr: FileHeader, batch_index_base: int = 0):
        self._chk(self.lib.pcr_stream_begin(self.h, C.byref(hdr), batch_index_base), "pcr_stream_begin")

    def upload_batch(self, index: int, blob) -> None:
        mv = memoryview(blob).cast("B")
        arr = np.frombuffer(mv, np.uint8)
        self._chk(self.lib.pcr_upload_batch(self.h, index, arr.ctypes.data, len(mv)), "pcr_upload_batch")

    def upload_batches(self, first: int, blobs) -> None:
        """One loader task: `blobs` are the records of batches first, first+1, ... (buffer-protocol objects)."""
        arrs = [np.frombuffer(memoryview(b).cast("B"), np.uint8) for b in blobs]
        n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        sizes = (C.c_size_t * n)(*[a.size for a in arrs])
        self._chk(self.lib.pcr_upload_batches(self.h, first, n, ptrs, sizes), "pcr_upload_batches")

    def upload_tail(self, enc: np.ndarray, sep: np.ndarray) -> None:
        enc = np.ascontiguousarray(enc, np.uint32); sep = np.ascontiguousarray(sep, np.int32)
        self._chk(self.lib.pcr_upload_tail(self.h, enc.ctypes.data, len(enc), sep.ctypes.data, len(sep)), "pcr_upload_tail")

    def stream_unload(self):
        self._chk(self.lib.pcr_stream_unload(self.h), "pcr_stream_unload")

    @property
    def batches_loaded(self) -> int:
        return int(self.lib.pcr_batches_loaded(self.h))

    @property
    def points_loaded(self) -> int:
        return int(self.lib.pcr_points_loaded(self.h))

    @property
    def algorithmic_bytes(self) -> int:
        return int(self.lib.pcr_stream_algorithmic_bytes(self.h))

    @property
    def last_frame_algorithmic_bytes(self) -> int:
        """Algorithmic bytes of the frame the last render call drew: culled batches left out, a drawn batch's words by the
        share of its chains' points its level of detail decodes (pcr_last_frame_algorithmic_bytes; synchronises)."""
        return int(self.lib.pcr_last_frame_algorithmic_bytes(self.h))

    # method side
    # -- GPU encoder (include/pcr_gpu_encode.h) ---------------------------------------------------------
    @property
    def resident_bytes(self) -> int:
        """Device bytes the loaded stream occupies right now (pcr_stream_resident_bytes)."""
        return int(self.lib.pcr_stream_resident_bytes(self.h))

    def stream_color_format(self) -> int:
        """1 (BC1) / 7 (BC7 mode 6) / 0 before the first record (pcr_stream_color_format)."""
        return int(self.lib.pcr_stream_color_format(self.h))

    def gpu_encode_points(self, x, y, z, color, las: LasInfo, morton_sort: bool = True, chunk_points: int = 0,
                          pad_tails: bool = False) -> tuple[NativeBytes, dict]:
        """The encoder of `encode_points`, run on the GPU; same file image byte for byte."""
        x = np.ascontiguousarray(x, np.int32); y = np.ascontiguousarray(y, np.int32); z = np.ascontiguousarray(z, np.int32)
        color = np.ascontiguousarray(color, np.uint32)
        out, ln, st = C.c_void_p(), C.c_size_t(), EncodeStats()
        self._chk(self.lib.pcr_gpu_encode_points(self.h, x.ctypes.data, y.ctypes.data, z.ctypes.data, color.ctypes.data, len(x),
                                                 C.byref(las), int(bool(morton_sort)) | (2 if pad_tails else 0), chunk_points,
                                                 C.byref(out), C.byref(ln), C.byref(st)), "pcr_gpu_encode_points")
        return NativeBytes(out.value, ln.value), st.as_dict()

    # -- 10-10-10 resource / method ----------------------------------------------------------------
    def las_begin(self, num_points: int):
        self._chk(self.lib.pcr_las_begin(self.h, num_points), "pcr_las_begin")

    def las_upload(self, first_batch: int, batches, xyz12, xyz8, xyz4, rgba) -> None:
        count = len(batches)
        for a in (xyz12, xyz8, xyz4, rgba):
            if a.dtype != np.uint32 or not a.flags.c_contiguous or len(a) != count * POINTS_PER_BATCH:
                raise ValueError("level arrays must be contiguous uint32 with 65536 slots per batch")
        self._chk(self.lib.pcr_las_upload(self.h, first_batch, count, C.addressof(batches), xyz12.ctypes.data,
                                           xyz8.ctypes.data, xyz4.ctypes.data, rgba.ctypes.data), "pcr_las_upload")

    def las_unload(self):
        self._chk(self.lib.pcr_las_unload(self.h), "pcr_las_unload")

    @property
    def las_batches_loaded(self) -> int:
        return int(self.lib.pcr_las_batches_loaded(self.h))

    @property
    def las_algorithmic_bytes(self) -> int:
        return int(self.lib.pcr_las_algorithmic_bytes(self.h))

    def render_las(self, p: RenderParams):
        self._chk(self.lib.pcr_render_las(self.h, C.byref(p)), "pcr_render_las")

    def resolve_las(self, p: RenderParams):
        self._chk(self.lib.pcr_resolve_las(self.h, C.byref(p)), "pcr_resolve_las")

    def set_image_size(self, w: int, h: int):
        self._chk(self.lib.pcr_set_image_size(self.h, w, h), "pcr_set_image_size")
        self.width, self.height = w, h

    def clear(self):
        self._chk(self.lib.pcr_clear(self.h), "pcr_clear")

    def frame_begin(self, p: RenderParams, hqs: bool = False):
        """pcr_clear + the prepass of the render call that follows, in one launch (pcr_hip.h)."""
        self._chk(self.lib.pcr_frame_begin(self.h, C.byref(p), 1 if hqs else 0), "pcr_frame_begin")

    def frame_turn(self, done: RenderParams, nxt: RenderParams, hqs: bool = False):
        """Resolve the finished frame, clear, and run the next frame's prepass in one launch (pcr_frame_turn)."""
        self._chk(self.lib.pcr_frame_turn(self.h, C.byref(done), C.byref(nxt), 1 if hqs else 0), "pcr_frame_turn")

    def render_basic(self, p: RenderParams):
        self._chk(self.lib.pcr_render_basic(self.h, C.byref(p)), "pcr_render_basic")

    def render_hqs_depth(self, p: RenderParams):
        self._chk(self.lib.pcr_render_hqs_depth(self.h, C.byref(p)), "pcr_render_hqs_depth")

    def render_hqs_color(self, p: RenderParams):
        self._chk(self.lib.pcr_render_hqs_color(self.h, C.byref(p)), "pcr_render_hqs_color")

    def resolve_basic(self, p: RenderParams):
        self._chk(self.lib.pcr_resolve_basic(self.h, C.byref(p)), "pcr_resolve_basic")

    def resolve_hqs(self, p: RenderParams):
        self._chk(self.lib.pcr_resolve_hqs(self.h, C.byref(p)), "pcr_resolve_hqs")

    def synchronize(self):
        self._chk(self.lib.pcr_synchronize(self.h), "pcr_synchronize")

    def stats(self) -> dict:
        st = RenderStats()
        self._chk(self.lib.pcr_get_stats(self.h, C.byref(st)), "pcr_get_stats")
        return st.as_dict()

    def read_framebuffer(self, full: bool = False) -> np.ndarray:
        n = fb_elems(self.width, self.height) if full else self.width * self.height
        out = np.empty(n, np.uint64)
        self._chk(self.lib.pcr_read_framebuffer(self.h, out.ctypes.data, n), "pcr_read_framebuffer")
        return out

    def read_accum(self, full: bool = False):
        n = fb_elems(self.width, self.height) if full else self.width * self.height
        rg, ba = np.empty(n, np.uint64), np.empty(n, np.uint64)
        self._chk(self.lib.pcr_read_accum(self.h, rg.ctypes.data, ba.ctypes.data, n), "pcr_read_accum")
        return rg, ba

    def read_rgba(self) -> np.ndarray:
        n = self.width * self.height
        out = np.empty(n, np.uint32)
        self._chk(self.lib.pcr_read_rgba(self.h, out.ctypes.data, n), "pcr_read_rgba")
        return out

    # multi-GPU plumbing / measurement
    def device_framebuffer(self) -> int:
        return int(self.lib.pcr_device_framebuffer(self.h) or 0)

    def framebuffer_private(self) -> None:
        """Pointers handed out by device_framebuffer() are no longer written through: dirty tiles are used again (pcr_hip.h)."""
        self._chk(self.lib.pcr_framebuffer_private(self.h), "pcr_framebuffer_private")

    def use_external_buffers(self, fb: int = 0, rg: int = 0, ba: int = 0):
        self._chk(self.lib.pcr_use_external_buffers(self.h, C.c_void_p(fb or None), C.c_void_p(rg or None), C.c_void_p(ba or None)),
                  "pcr_use_external_buffers")

    def set_stream(self, hip_stream: int):
        self._chk(self.lib.pcr_set_stream(self.h, C.c_void_p(hip_stream or None)), "pcr_set_stream")

    def merge_min(self, other_fb: int):
        self._chk(self.lib.pcr_merge_min(self.h, C.c_void_p(other_fb)), "pcr_merge_min")

    def merge_sum(self, other_rg: int, other_ba: int):
        self._chk(self.lib.pcr_merge_sum(self.h, C.c_void_p(other_rg or None), C.c_void_p(other_ba or None)), "pcr_merge_sum")

    def flip_sign(self):
        self._chk(self.lib.pcr_flip_sign(self.h), "pcr_flip_sign")

    LAYOUT_WORDS, LAYOUT_POINT_WINDOWS, LAYOUT_BOTH, LAYOUT_AUTO = 0, 1, 2, 3

    def set_stream_layout(self, layout: int) -> None:
        """HBM layout of the next stream this context loads (pcr_hip.h: PCR_LAYOUT_*)."""
        self._chk(self.lib.pcr_set_stream_layout(self.h, int(layout)), "pcr_set_stream_layout")

    def set_hbm_budget(self, nbytes: int) -> None:
        """LAYOUT_AUTO: a stream whose point windows would take more than this many bytes is loaded as packed words."""
        self._chk(self.lib.pcr_set_hbm_budget(self.h, int(nbytes)), "pcr_set_hbm_budget")

    @property
    def stream_layout(self) -> int:
        return int(self.lib.pcr_stream_layout(self.h))

    VARIANT_AUTO, VARIANT_WORDS, VARIANT_POINT_WINDOWS = 0, 1, 2

    def set_render_variant(self, variant: int) -> None:
        """Which decode variant draws a stream that has both layouts resident (pcr_hip.h: PCR_VARIANT_*)."""
        self._chk(self.lib.pcr_set_render_variant(self.h, int(variant)), "pcr_set_render_variant")

    def set_workgroup_parts(self, parts: int) -> None:
        """Workgroups per batch of the render kernels: 0 = library's choice, 1 = whole batches (1024 threads), 2 = half-batches."""
        self._chk(self.lib.pcr_set_workgroup_parts(self.h, int(parts)), "pcr_set_workgroup_parts")

    def merge_min_slices(self, slices_ptr: int, nslices: int, slice_elems: int) -> None:
        self._chk(self.lib.pcr_merge_min_slices(self.h, slices_ptr, nslices, slice_elems), "pcr_merge_min_slices")

    def resolve_basic_range(self, p: RenderParams, fb_ptr: int, count: int, rgba_ptr: int) -> None:
        self._chk(self.lib.pcr_resolve_basic_range(self.h, C.byref(p), fb_ptr, count, rgba_ptr), "pcr_resolve_basic_range")

    def fence_record(self, slot: int, hip_stream: int = 0) -> None:
        self._chk(self.lib.pcr_fence_record(self.h, slot, hip_stream), "pcr_fence_record")

    def fence_wait(self, slot: int, hip_stream: int = 0) -> None:
        self._chk(self.lib.pcr_fence_wait(self.h, slot, hip_stream), "pcr_fence_wait")

    def set_int64_mergeable(self, on: bool) -> None:
        """Empty pixels as INT64_MAX so that a signed 64-bit MIN collective orders the framebuffer (pcr_hip.h)."""
        self._chk(self.lib.pcr_set_int64_mergeable(self.h, int(on)), "pcr_set_int64_mergeable")

    def set_async_upload(self, on: bool) -> None:
        """Loader copies + transcode on the context's loader stream; frames draw what has arrived (pcr_hip.h)."""
        self._chk(self.lib.pcr_set_async_upload(self.h, int(on)), "pcr_set_async_upload")

    @property
    def batches_resident(self) -> int:
        return int(self.lib.pcr_batches_resident(self.h))

    @property
    def last_frame_batches(self) -> int:
        return int(self.lib.pcr_last_frame_batches(self.h))

    def measure_hbm(self, nbytes: int = 2 << 30, reps: int = 5) -> tuple[float, float]:
        """(streaming read GB/s, streaming copy GB/s) of this device: the practical HBM ceiling."""
        r, w = C.c_float(), C.c_float()
        self._chk(self.lib.pcr_measure_hbm(self.h, nbytes, reps, C.byref(r), C.byref(w)), "pcr_measure_hbm")
        return float(r.value), float(w.value)

    def kernel_timing(self, every: int) -> None:
        """Bracket every `every`-th decode+rasterize launch with HIP events (0/False = off)."""
        self._chk(self.lib.pcr_kernel_timing_enable(self.h, int(every)), "pcr_kernel_timing_enable")

    def kernel_timing_read(self) -> tuple[float, int]:
        """(average ms of the decode+rasterize kernel over the most recent launches, how many)."""
        ms, n = C.c_float(), C.c_int()
        self._chk(self.lib.pcr_kernel_timing_read(self.h, C.byref(ms), C.byref(n)), "pcr_kernel_timing_read")
        return float(ms.value), int(n.value)

    def timing_begin(self):
        self._chk(self.lib.pcr_timing_begin(self.h), "pcr_timing_begin")

    def timing_end(self) -> float:
        ms = C.c_float()
        self._chk(self.lib.pcr_timing_end(self.h, C.byref(ms)), "pcr_timing_end")
        return float(ms.value)


# --------------------------------------------------------------------------------------------------
# the reference's plugin surface, headless
# --------------------------------------------------------------------------------------------------
class Debug:                      # include/Debug.h:14-31 (the flags the Huffman methods read)
    LOD = 0.1
    frustumCullingEnabled = True
    colorizeChunks = False
    showNumPoints = False


class Renderer:
    """Headless stand-in for the reference Renderer: window size + orbit camera + one GPU context."""

    def __init__(self, width: int = 1920, height: int = 1080, device: int = 0):   # Renderer.cpp:142-143
        self.width, self.height = width, height
        self.ctx = Context(device)
        self.ctx.set_image_size(width, height)
        self.yaw, self.pitch, self.radius, self.target = 0.0, 0.0, 10.0, (0.0, 0.0, 0.0)
        self.params_override: Optional[RenderParams] = None

    def set_camera(self, yaw, pitch, radius, target):
        self.yaw, self.pitch, self.radius, self.target = yaw, pitch, radius, tuple(target)

    def render_params(self) -> RenderParams:
        p = self.params_override.copy() if self.params_override is not None else \
            camera_orbit(self.yaw, self.pitch, self.radius, self.target, self.width, self.height)
        p.lod_percent = int(Debug.LOD * 100)                       # huffman_hqs.h:177
        p.enable_frustum_culling = int(Debug.frustumCullingEnabled)
        p.colorize_chunks = int(Debug.colorizeChunks)
        p.show_num_points = int(Debug.showNumPoints)
        return p


class Resource:                   # modules/compute/Resources.h:20-35
    UNLOADED, LOADING, LOADED, UNLOADING = range(4)

    def __init__(self):
        self.state = Resource.UNLOADED

    def load(self, renderer): raise NotImplementedError
    def unload(self, renderer): raise NotImplementedError
    def process(self, renderer): raise NotImplementedError


class Method:                     # include/Method.h:10-23
    name = "no name"
    description = ""
    group = "no group"

    def update(self, renderer): raise NotImplementedError
    def render(self, renderer): raise NotImplementedError


class Runtime:                    # include/Runtime.h:15-55
    methods: list = []
    selectedMethod: Optional[Method] = None
    resource: Optional[Resource] = None

    @staticmethod
    def addMethod(m: Method):
        Runtime.methods.append(m)

    @staticmethod
    def setSelectedMethod(name: str):
        for m in Runtime.methods:
            if m.name == name:
                Runtime.selectedMethod = m

    @staticmethod
    def getSelectedMethod():
        return Runtime.selectedMethod

    @staticmethod
    def reset():
        Runtime.methods, Runtime.selectedMethod, Runtime.resource = [], None, None


class HuffmanLasData(Resource):
    """modules/compute/HuffmanLasLoader.{h,cpp}. `first_batch`/`num_batches` select a contiguous shard
    (multi-GPU); the reference always loads the whole file."""

    BATCHES_PER_TASK = 100        # HuffmanLasLoader.cpp:106

    def __init__(self):
        super().__init__()
        self.path = ""
        self.file: Optional[HuffmanFile] = None
        self.numBatches = self.numPoints = 0
        self.numBatchesLoaded = self.numPointsLoaded = 0
        self.first_batch = 0
        self._next = 0

    @staticmethod
    def create(path_or_bytes, first_batch: int = 0, num_batches: Optional[int] = None) -> "HuffmanLasData":
        d = HuffmanLasData()
        d.path = path_or_bytes if isinstance(path_or_bytes, (str, os.PathLike)) else "<memory>"
        d.file = HuffmanFile(path_or_bytes)                          # loadHeader()
        d.first_batch = first_batch
        d.numBatches = d.file.numBatches - first_batch if num_batches is None else num_batches
        d.numPoints = d.numBatches * POINTS_PER_BATCH
        return d

    def load(self, renderer: Renderer):
        if self.state != Resource.UNLOADED:                          # HuffmanLasLoader.cpp:25-31
            return
        self.state = Resource.LOADING
        hdr = self.file.header(self.first_batch, self.numBatches)
        renderer.ctx.stream_begin(hdr, self.first_batch)
        self._next = 0
        self.numBatchesLoaded = self.numPointsLoaded = 0

    def process(self, renderer: Renderer):
        """Upload the next task of <= 100 batches (HuffmanLasLoader.cpp:301-313; the reference's reader
        thread hands them over one task per frame — here the file is already mapped)."""
        if self.state not in (Resource.LOADING,):
            return
        end = min(self.numBatches, self._next + self.BATCHES_PER_TASK)
        renderer.ctx.upload_batches(self._next, [self.file.blob(self.first_batch + i) for i in range(self._next, end)])
        self._next = end
        self.numBatchesLoaded = renderer.ctx.batches_loaded
        self.numPointsLoaded = renderer.ctx.points_loaded
        if self._next == self.numBatches:
            nxt = self.first_batch + self.numBatches
            if nxt < self.file.numBatches:                            # shard boundary: carry the follower's head words
                renderer.ctx.upload_tail(*self.file.head_words(nxt))
            self.state = Resource.LOADED

    def load_all(self, renderer: Renderer):
        self.load(renderer)
        while self.state == Resource.LOADING:
            self.process(renderer)

    def unload(self, renderer: Renderer):
        self.numBatchesLoaded = 0
        renderer.ctx.stream_unload()
        self.state = Resource.UNLOADED


class _HuffmanMethod(Method):
    group = "none"

    def __init__(self, renderer: Renderer, las: HuffmanLasData):
        self.renderer, self.las = renderer, las

    def update(self, renderer: Renderer):                            # huffman_hqs.h:116-124
        if Runtime.resource is not self.las:
            if Runtime.resource is not None:
                Runtime.resource.unload(renderer)
            self.las.load(renderer)
            Runtime.resource = self.las


class HuffmanMemIter(_HuffmanMethod):
    """modules/huffman_mem_iter_cuda/huffman_mem_iter_cuda.h:122-254: decode + {depth,BC1 colour} atomicMin."""
    name = "huffman_mem_iter_cuda"
    description = "- Decodes Huffman Encoded values on the GPU"

    def render(self, renderer: Renderer):
        """One frame. The reference clears at the END of render() for the next frame (:250-252); headless
        callers want to read the result, so the clear happens at the START of the next frame instead."""
        self.las.process(renderer)
        if self.las.numPointsLoaded == 0:
            return
        p = renderer.render_params()
        ctx = renderer.ctx
        ctx.frame_begin(p)            # CLEAR (of the previous frame) + the cull/LOD prepass, one launch
        ctx.render_basic(p)           # RENDER
        ctx.resolve_basic(p)          # RESOLVE
        self.last_params = p


class ComputeHuffman(HuffmanMemIter):
    """modules/huffman_cuda/huffman_cuda.h:60-75: the reference's first Huffman method, registered as "huffman_cuda" (commented
    out in its main.cpp:19, 265 in favour of huffman_mem_iter_cuda: same decode + atomicMin raster). The name north_star lists."""
    name = "huffman_cuda"


class HuffmanHQS(_HuffmanMethod):
    """modules/huffman_hqs/huffman_hqs.h:126-273: depth pass, 1 % colour accumulation pass, averaging resolve."""
    name = "huffman_hqs"
    description = "- Decodes Huffman Encoded values on the GPU"

    def render(self, renderer: Renderer):
        self.las.process(renderer)
        if self.las.numPointsLoaded == 0:
            return
        p = renderer.render_params()
        ctx = renderer.ctx
        ctx.frame_begin(p, hqs=True)
        ctx.render_hqs_depth(p)
        ctx.render_hqs_color(p)
        ctx.resolve_hqs(p)
        self.last_params = p


class ComputeLasData(Resource):
    """modules/compute/ComputeLasLoader.{h,cpp}: a LAS file quantised batch by batch into three 10-10-10 levels.
    The reference uploads raw LAS bytes and quantises in a compute shader (computeLasLoader.cs); here the host library
    quantises (pcr_las_quantize) and the four arrays are uploaded."""

    POINTS_PER_TASK = 100 * POINTS_PER_BATCH      # MAX_POINTS_PER_BATCH, Resources.h:10

    def __init__(self):
        super().__init__()
        self.path = ""
        self.numPoints = self.numPointsLoaded = self.numBatchesLoaded = 0
        self._pts = None
        self.las: Optional[LasInfo] = None

    @staticmethod
    def create(path: str) -> "ComputeLasData":                        # ComputeLasLoader.h:97-103
        return ComputeLasData.from_points(*read_las(path), path=path)

    @staticmethod
    def from_points(x, y, z, color, las: LasInfo, path: str = "<memory>") -> "ComputeLasData":
        d = ComputeLasData()
        d.path, d.las = path, las
        d._pts = tuple(np.ascontiguousarray(a, t) for a, t in ((x, np.int32), (y, np.int32), (z, np.int32), (color, np.uint32)))
        d.numPoints = len(d._pts[0])
        return d

    def load(self, renderer: Renderer):                               # ComputeLasLoader.cpp:14-38
        if self.state != Resource.UNLOADED:
            return
        self.state = Resource.LOADING
        renderer.ctx.las_begin(self.numPoints)
        self.numPointsLoaded = self.numBatchesLoaded = 0

    def process(self, renderer: Renderer):                            # ComputeLasLoader.cpp:140-262
        if self.state != Resource.LOADING:
            return
        a, b = self.numPointsLoaded, min(self.numPoints, self.numPointsLoaded + self.POINTS_PER_TASK)
        q = las_quantize(*(v[a:b] for v in self._pts), self.las)
        renderer.ctx.las_upload(self.numBatchesLoaded, *q)
        self.numPointsLoaded = b
        self.numBatchesLoaded = renderer.ctx.las_batches_loaded
        if b == self.numPoints:
            self.state = Resource.LOADED

    def load_all(self, renderer: Renderer):
        self.load(renderer)
        while self.state == Resource.LOADING:
            self.process(renderer)

    def unload(self, renderer: Renderer):                             # ComputeLasLoader.cpp:114-131
        self.numPointsLoaded = self.numBatchesLoaded = 0
        renderer.ctx.las_unload()
        self.state = Resource.UNLOADED


class ComputeLoopLasCUDA(Method):
    """modules/compute_loop_las_cuda/compute_loop_las_cuda.h:52-222: per-batch level of detail picks how many of the
    three 10-bit levels a workgroup reads; {depth, point index} atomicMin; resolve looks the colour up by index."""
    name = "loop_las_cuda"
    description = "- Each thread renders X points.\n- Loads points from LAS file\n- encodes point coordinates in 10+10+10 bits"
    group = "10-10-10 bit encoded"

    def __init__(self, renderer: Renderer, las: ComputeLasData):
        self.renderer, self.las = renderer, las

    def update(self, renderer: Renderer):                             # empty in the reference (:92-93); resource switch as huffman_hqs.h:116-124
        if Runtime.resource is not self.las:
            if Runtime.resource is not None:
                Runtime.resource.unload(renderer)
            self.las.load(renderer)
            Runtime.resource = self.las

    def render(self, renderer: Renderer):                             # compute_loop_las_cuda.h:99-222
        self.las.process(renderer)
        if self.las.numPointsLoaded == 0:
            return
        p = renderer.render_params()
        ctx = renderer.ctx
        ctx.clear()
        ctx.render_las(p)
        ctx.resolve_las(p)
        self.last_params = p
