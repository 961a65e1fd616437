"""The C-ABI libraries load and export every symbol include/*.h declares (no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

from pcrhpg24_amd import _native as N
from pcrhpg24_amd import build

INC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
DECL = re.compile(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+(pcr_[a-z0-9_]+)\s*\(", re.M)


def declared(header):
    text = open(os.path.join(INC, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(DECL.findall(text))
    names.discard("pcr_fb_elems")       # static inline helper
    return names


def test_headers_and_binding_tables_agree():
    assert declared("pcr_hip.h") | declared("pcr_gpu_encode.h") == set(N.HIP_SYMBOLS)
    assert declared("pcr_encode.h") == set(N.HOST_SYMBOLS)


def test_hip_library_exports_every_declared_symbol():
    build.build_hip()
    lib = C.CDLL(build.HIP_LIB)
    for name in sorted(declared("pcr_hip.h") | declared("pcr_gpu_encode.h")):
        assert hasattr(lib, name), f"{name} is declared in include/pcr_hip.h / pcr_gpu_encode.h but not exported"


def test_host_library_exports_every_declared_symbol():
    lib = C.CDLL(build.build_host())
    for name in sorted(declared("pcr_encode.h")):
        assert hasattr(lib, name), f"{name} is declared in include/pcr_encode.h but not exported"


def test_struct_sizes_match_the_c_headers():
    assert C.sizeof(N.GpuBatch) == 160                  # GPUBatchSize, huffman_kernel_data.h:4
    assert C.sizeof(N.FileHeader) == 40
    assert C.sizeof(N.RenderParams) == 3 * 64 + 8 * 4
    assert C.sizeof(N.RenderStats) == 32


def test_create_without_gpu_fails_loudly():
    """No CPU fallback: on a machine without an MI355X pcr_create reports an error instead of rendering elsewhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = N.hip_lib()
    h = C.c_void_p()
    rc = lib.pcr_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert b"HIP" in lib.pcr_last_error(None) or b"device" in lib.pcr_last_error(None)
