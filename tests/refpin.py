"""ctypes binding of oracle/_ref/libpcr_ref.so: the reference's own huffman.h / mymorton.h / rgbcx.cpp compiled
in place (oracle/ref_harness.cpp). Exists only where /root/reference does (and on the GPU box as a shipped .so).
Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from tests.oracle import REF_LIB

_ref = None


def available() -> bool:
    return os.path.exists(REF_LIB)


def ref_lib() -> C.CDLL:
    global _ref
    if _ref is None:
        L = C.CDLL(REF_LIB)
        L.ref_code_build.restype = C.c_void_p
        L.ref_code_build.argtypes = [C.c_void_p, C.c_int64]
        L.ref_code_build_sorted.restype = C.c_void_p
        L.ref_code_build_sorted.argtypes = [C.c_void_p, C.c_int64]
        L.ref_code_free.argtypes = [C.c_void_p]
        L.ref_code_dict_size.restype = C.c_int64
        L.ref_code_dict_size.argtypes = [C.c_void_p]
        L.ref_code_dict.argtypes = [C.c_void_p] * 4
        L.ref_code_table.argtypes = [C.c_void_p] * 3
        L.ref_code_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]
        L.ref_code_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_morton_key.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.ref_morton_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.ref_bc1_encode.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_bc1_unpack.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_bc7_encode.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_bc7_unpack.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_bc7_unpack.restype = C.c_int
        _ref = L
    return _ref


def _take(ptr, n, dtype):
    if n == 0:
        out = np.zeros(0, dtype)
    else:
        out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32 if dtype == np.uint32 else C.c_int32)), (n,)).copy().astype(dtype)
    ref_lib().ref_free(ptr)
    return out


class RefCode:
    """The reference's Huffman<int32_t> for one batch of symbols (Batch::calculate, src/preprocess.cpp:765-770)."""

    def __init__(self, symbols: np.ndarray, sorted_tree: bool = False):
        symbols = np.ascontiguousarray(symbols, np.int32)
        f = ref_lib().ref_code_build_sorted if sorted_tree else ref_lib().ref_code_build
        self.h = f(symbols.ctypes.data, len(symbols))

    def __del__(self):
        if getattr(self, "h", None):
            ref_lib().ref_code_free(self.h)
            self.h = None

    def dict(self):
        n = ref_lib().ref_code_dict_size(self.h)
        s, c, l = np.zeros(n, np.int32), np.zeros(n, np.uint32), np.zeros(n, np.int32)
        ref_lib().ref_code_dict(self.h, s.ctypes.data, c.ctypes.data, l.ctypes.data)
        return s, c, l

    def table(self):
        v, l = np.zeros(4096, np.int32), np.zeros(4096, np.int32)
        ref_lib().ref_code_table(self.h, v.ctypes.data, l.ctypes.data)
        return v, l

    def pack(self, chain: np.ndarray):
        chain = np.ascontiguousarray(chain, np.int32)
        w, s, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nw, ns = C.c_int32(), C.c_int32()
        ref_lib().ref_code_pack(self.h, chain.ctypes.data, len(chain), C.byref(w), C.byref(nw), C.byref(s), C.byref(ns), C.byref(n))
        return _take(w, nw.value, np.uint32), _take(s, ns.value, np.int32), _take(n, nw.value, np.int32)

    def unpack(self, words, separate, n: int):
        words = np.ascontiguousarray(words, np.uint32)
        separate = np.ascontiguousarray(separate, np.int32)
        out = np.zeros(n, np.int32)
        ref_lib().ref_code_unpack(self.h, words.ctypes.data, len(words), separate.ctypes.data, len(separate), n, out.ctypes.data)
        return out
