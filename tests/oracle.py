"""ctypes binding of oracle/libpcr_oracle.so — the CPU checker. Test infrastructure only: nothing under
pcrhpg24_amd/ imports this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from pcrhpg24_amd._native import GpuBatch, RenderParams, RenderStats, XyzBatch, fb_elems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libpcr_oracle.so")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libpcr_ref.so")

MEM_ITER, HQS = 0, 1


def build() -> None:
    src = [os.path.join(ORACLE_DIR, f) for f in ("pcr_oracle.c", "pcr_oracle.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in src):
        subprocess.run(["make", "-C", ORACLE_DIR, "libpcr_oracle.so"], check=True, stdout=subprocess.DEVNULL)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.pcr_oracle_file_parse.restype = C.c_void_p
        L.pcr_oracle_file_parse.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.pcr_oracle_file_stream.restype = C.c_void_p
        L.pcr_oracle_file_stream.argtypes = [C.c_void_p]
        L.pcr_oracle_file_free.argtypes = [C.c_void_p]
        L.pcr_oracle_file_free.restype = None
        L.pcr_oracle_decode_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.pcr_oracle_decode_batch.restype = None
        L.pcr_oracle_batch_lod.argtypes = [C.POINTER(GpuBatch), C.POINTER(RenderParams), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.pcr_oracle_decode_bc1.restype = C.c_uint32
        L.pcr_oracle_decode_bc1.argtypes = [C.c_uint64, C.c_void_p]
        L.pcr_oracle_decode_bc7.restype = C.c_uint32
        L.pcr_oracle_decode_bc7.argtypes = [C.c_uint64, C.c_void_p]
        for n in ("pcr_oracle_render_basic", "pcr_oracle_render_hqs_depth"):
            getattr(L, n).argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int64, C.c_int64, C.c_void_p, C.POINTER(RenderStats)]
            getattr(L, n).restype = None
        L.pcr_oracle_render_basic_mt.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.POINTER(RenderStats)]
        L.pcr_oracle_render_hqs_color.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RenderStats)]
        L.pcr_oracle_render_hqs_color.restype = None
        L.pcr_oracle_count_depth_ties.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int64, C.c_int64, C.c_void_p,
                                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.pcr_oracle_resolve_basic.argtypes = [C.POINTER(RenderParams), C.c_void_p, C.c_void_p]
        L.pcr_oracle_resolve_basic.restype = None
        L.pcr_oracle_resolve_hqs.argtypes = [C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pcr_oracle_resolve_hqs.restype = None
        L.pcr_oracle_decode_chain.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.pcr_oracle_decode_chain.restype = None
        L.pcr_oracle_las_level.argtypes = [C.POINTER(XyzBatch), C.POINTER(RenderParams)]
        L.pcr_oracle_render_las.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RenderParams),
                                            C.c_void_p, C.POINTER(RenderStats)]
        L.pcr_oracle_render_las.restype = None
        L.pcr_oracle_resolve_las.argtypes = [C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p]
        L.pcr_oracle_resolve_las.restype = None
        L.pcr_oracle_lane_words.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        L.pcr_oracle_decode_chain_from_lane_words.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.pcr_oracle_decode_chain_from_lane_words.restype = None
        _lib = L
    return _lib


class OracleStreamStruct(C.Structure):
    _fields_ = [("num_batches", C.c_int64), ("batches", C.POINTER(GpuBatch)), ("start_values", C.c_void_p),
                ("encoded", C.c_void_p), ("encoded_words", C.c_int64), ("separate", C.c_void_p),
                ("separate_words", C.c_int64), ("separate_sizes", C.c_void_p), ("dt_values", C.c_void_p),
                ("dt_cwlen", C.c_void_p), ("cluster_sizes", C.c_void_p), ("colors", C.c_void_p),
                ("batch_index_base", C.c_int64), ("color_format", C.c_int64)]


class OracleFile:
    """A .huffman image parsed into the reference loader's flat arrays (HuffmanLasLoader.cpp:176-299)."""

    def __init__(self, data):
        mv = memoryview(data).cast("B")
        self._src = np.frombuffer(mv, np.uint8)
        err = C.create_string_buffer(256)
        self._f = lib().pcr_oracle_file_parse(self._src.ctypes.data, len(mv), err, 256)
        if not self._f:
            raise ValueError("oracle parse failed: " + err.value.decode())
        self.stream = lib().pcr_oracle_file_stream(self._f)
        self.s = OracleStreamStruct.from_address(self.stream)
        self.num_batches = int(self.s.num_batches)

    def __del__(self):
        if getattr(self, "_f", None):
            lib().pcr_oracle_file_free(self._f)
            self._f = None

    # -- raw arrays -------------------------------------------------------------------------------
    def encoded(self) -> np.ndarray:
        return np.ctypeslib.as_array(C.cast(self.s.encoded, C.POINTER(C.c_uint32)), (int(self.s.encoded_words),))

    def separate(self) -> np.ndarray:
        return np.ctypeslib.as_array(C.cast(self.s.separate, C.POINTER(C.c_int32)), (int(self.s.separate_words),))

    def batch(self, b: int) -> GpuBatch:
        return self.s.batches[b]

    # -- oracle calls -----------------------------------------------------------------------------
    def decode_batch(self, b: int, npr: int = 64) -> np.ndarray:
        out = np.zeros((1024, 64, 3), np.int32)
        lib().pcr_oracle_decode_batch(self.stream, b, npr, out.ctypes.data)
        return out

    def lane_words(self, b: int, rows: int = 80):
        """(words[rows, 1024], counts[1024]) of the lockstep walk: the lane-major form the HIP path decodes from."""
        out = np.zeros((rows, 1024), np.uint32)
        counts = np.zeros(1024, np.int32)
        rc = lib().pcr_oracle_lane_words(self.stream, b, rows, out.ctypes.data, counts.ctypes.data)
        assert rc == 0, "more words per chain than rows"
        return out, counts

    def decode_chain_from_lane_words(self, b: int, chain: int, words: np.ndarray, count: int, npr: int = 64) -> np.ndarray:
        out = np.zeros((npr, 3), np.int32)
        lib().pcr_oracle_decode_chain_from_lane_words(self.stream, b, chain, words[:, chain:].ctypes.data, count, npr, out.ctypes.data)
        return out

    def batch_lod(self, b: int, p: RenderParams, variant: int = MEM_ITER):
        npr, dbl = C.c_int(), C.c_int()
        vis = lib().pcr_oracle_batch_lod(C.byref(self.s.batches[b]), C.byref(p), variant, C.byref(npr), C.byref(dbl))
        return bool(vis), npr.value, bool(dbl.value)

    def new_fb(self, p: RenderParams) -> np.ndarray:
        return np.full(fb_elems(p.width, p.height), 0xFFFFFFFFFFFFFFFF, np.uint64)

    def render_basic(self, p: RenderParams, fb=None, first=0, count=None, nthreads=1):
        fb = self.new_fb(p) if fb is None else fb
        st = RenderStats()
        count = self.num_batches - first if count is None else count
        if nthreads > 1:
            lib().pcr_oracle_render_basic_mt(self.stream, C.byref(p), first, count, fb.ctypes.data, nthreads, C.byref(st))
        else:
            lib().pcr_oracle_render_basic(self.stream, C.byref(p), first, count, fb.ctypes.data, C.byref(st))
        return fb, st.as_dict()

    def count_depth_ties(self, p: RenderParams, fb: np.ndarray, first=0, count=None) -> tuple[int, int]:
        """(pixels of the finished basic frame `fb` whose winning depth several points reached, those among them where
        the tied points differ in colour: the pixels at which the reference's own result is schedule dependent)."""
        count = self.num_batches - first if count is None else count
        a, b = C.c_int64(), C.c_int64()
        rc = lib().pcr_oracle_count_depth_ties(self.stream, C.byref(p), first, count, fb.ctypes.data, C.byref(a), C.byref(b))
        assert rc == 0
        return a.value, b.value

    def render_hqs_depth(self, p: RenderParams, fb=None, first=0, count=None):
        fb = self.new_fb(p) if fb is None else fb
        st = RenderStats()
        count = self.num_batches - first if count is None else count
        lib().pcr_oracle_render_hqs_depth(self.stream, C.byref(p), first, count, fb.ctypes.data, C.byref(st))
        return fb, st.as_dict()

    def render_hqs_color(self, p: RenderParams, fb, rg=None, ba=None, first=0, count=None):
        n = fb_elems(p.width, p.height)
        rg = np.zeros(n, np.uint64) if rg is None else rg
        ba = np.zeros(n, np.uint64) if ba is None else ba
        st = RenderStats()
        count = self.num_batches - first if count is None else count
        lib().pcr_oracle_render_hqs_color(self.stream, C.byref(p), first, count, fb.ctypes.data, rg.ctypes.data, ba.ctypes.data, C.byref(st))
        return rg, ba, st.as_dict()


def resolve_basic(p: RenderParams, fb: np.ndarray) -> np.ndarray:
    out = np.zeros(p.width * p.height, np.uint32)
    lib().pcr_oracle_resolve_basic(C.byref(p), fb.ctypes.data, out.ctypes.data)
    return out


def resolve_hqs(p: RenderParams, fb, rg, ba) -> np.ndarray:
    out = np.zeros(p.width * p.height, np.uint32)
    lib().pcr_oracle_resolve_hqs(C.byref(p), fb.ctypes.data, rg.ctypes.data, ba.ctypes.data, out.ctypes.data)
    return out


def decode_bc1(index: int, colors: np.ndarray) -> int:
    return int(lib().pcr_oracle_decode_bc1(index, colors.ctypes.data))


def decode_bc7(index: int, colors: np.ndarray) -> int:
    return int(lib().pcr_oracle_decode_bc7(index, colors.ctypes.data))


def decode_chain(words, separate, dt_values, dt_cwlen, n: int) -> np.ndarray:
    words = np.ascontiguousarray(words, np.uint32)
    separate = np.ascontiguousarray(np.concatenate([np.asarray(separate, np.int32), np.zeros(1, np.int32)]), np.int32)
    out = np.zeros(n, np.int32)
    lib().pcr_oracle_decode_chain(words.ctypes.data, len(words), separate.ctypes.data,
                                  np.ascontiguousarray(dt_values, np.int32).ctypes.data,
                                  np.ascontiguousarray(dt_cwlen, np.int32).ctypes.data, n, out.ctypes.data)
    return out


# ---- 10-10-10 path (modules/compute_loop_las_cuda) -------------------------------------------------
def las_level(batch: XyzBatch, p: RenderParams) -> int:
    return int(lib().pcr_oracle_las_level(C.byref(batch), C.byref(p)))


def render_las(batches, xyz12, xyz8, xyz4, p: RenderParams, fb=None, num_batches=None):
    fb = np.full(fb_elems(p.width, p.height), 0xFFFFFFFFFFFFFFFF, np.uint64) if fb is None else fb
    st = RenderStats()
    nb = len(batches) if num_batches is None else num_batches
    lib().pcr_oracle_render_las(C.addressof(batches), nb, xyz12.ctypes.data, xyz8.ctypes.data, xyz4.ctypes.data,
                                C.byref(p), fb.ctypes.data, C.byref(st))
    return fb, st.as_dict()


def resolve_las(p: RenderParams, fb: np.ndarray, rgba_points: np.ndarray) -> np.ndarray:
    out = np.zeros(p.width * p.height, np.uint32)
    lib().pcr_oracle_resolve_las(C.byref(p), fb.ctypes.data, rgba_points.ctypes.data, out.ctypes.data)
    return out
