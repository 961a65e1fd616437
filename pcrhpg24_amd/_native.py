"""ctypes bindings of the two native libraries. No compute happens in Python.

libpcr_hip.so is the product path: loading it fails loudly (RuntimeError) when the library is
missing — there is no CPU fallback for rendering.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

c_i64, c_i32, c_u64, c_u32, c_f32, c_f64 = C.c_int64, C.c_int32, C.c_uint64, C.c_uint32, C.c_float, C.c_double


class GpuBatch(C.Structure):                      # include/pcr_types.h: pcr_gpu_batch
    _fields_ = [(n, c_f32) for n in ("min_x", "min_y", "min_z", "max_x", "max_y", "max_z")] + \
               [(n, c_f64) for n in ("scale_x", "scale_y", "scale_z", "offset_x", "offset_y", "offset_z",
                                     "las_min_x", "las_min_y", "las_min_z", "las_max_x", "las_max_y", "las_max_z")] + \
               [(n, c_i64) for n in ("encoding_batch_offset", "separate_batch_offset", "decoder_table_offset",
                                     "cluster_sizes_offset", "max_cw_len")]


class XyzBatch(C.Structure):                      # pcr_xyz_batch (kernel_data.h:4-22)
    _fields_ = [("state", c_i32)] + [(n, c_f32) for n in ("min_x", "min_y", "min_z", "max_x", "max_y", "max_z")] + \
               [("num_points", c_i32), ("padding", c_i32 * 8)]


class FileHeader(C.Structure):                    # pcr_file_header
    _fields_ = [(n, c_i64) for n in ("num_points", "num_batches", "encoded_bytes", "separate_bytes", "cluster_bytes")]


class RenderParams(C.Structure):                  # pcr_render_params
    _fields_ = [("transform", c_f32 * 16), ("world_view", c_f32 * 16), ("proj", c_f32 * 16),
                ("width", c_i32), ("height", c_i32), ("points_per_thread", c_i32), ("lod_percent", c_i32),
                ("enable_frustum_culling", c_i32), ("show_num_points", c_i32), ("colorize_chunks", c_i32),
                ("reserved", c_i32)]

    def copy(self) -> "RenderParams":
        out = RenderParams()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(RenderParams))
        return out


class RenderStats(C.Structure):                   # pcr_render_stats
    _fields_ = [(n, c_i64) for n in ("batches_total", "batches_culled", "points_iterated", "batches_double")]

    def as_dict(self) -> dict:
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class LasInfo(C.Structure):                       # pcr_las_info
    _fields_ = [("scale", c_f64 * 3), ("offset", c_f64 * 3), ("min", c_f64 * 3), ("max", c_f64 * 3)]


class EncodeStats(C.Structure):                   # pcr_encode_stats
    _fields_ = [(n, c_i64) for n in ("num_points_in", "num_points", "num_batches", "encoded_bytes", "separate_bytes",
                                     "cluster_bytes", "escaped_symbols", "total_symbols", "file_bytes")]

    def as_dict(self) -> dict:
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


assert C.sizeof(XyzBatch) == 64 and C.sizeof(GpuBatch) == 160 and C.sizeof(FileHeader) == 40 and C.sizeof(RenderParams) == 224


def fb_elems(w: int, h: int) -> int:
    return w * (h + 1) + 1


HIP_SYMBOLS = [
    "pcr_create", "pcr_destroy", "pcr_last_error", "pcr_set_stream", "pcr_synchronize",
    "pcr_stream_begin", "pcr_upload_batch", "pcr_upload_batches", "pcr_upload_tail", "pcr_stream_unload", "pcr_batches_loaded",
    "pcr_points_loaded", "pcr_set_image_size", "pcr_clear", "pcr_render_basic", "pcr_render_hqs_depth",
    "pcr_render_hqs_color", "pcr_resolve_basic", "pcr_resolve_hqs", "pcr_get_stats", "pcr_read_framebuffer",
    "pcr_read_accum", "pcr_read_rgba", "pcr_device_framebuffer", "pcr_device_rg", "pcr_device_ba", "pcr_framebuffer_private",
    "pcr_use_external_buffers", "pcr_merge_min", "pcr_merge_sum", "pcr_flip_sign", "pcr_timing_begin",
    "pcr_timing_end", "pcr_kernel_timing_enable", "pcr_kernel_timing_read", "pcr_measure_hbm",
    "pcr_frame_begin", "pcr_frame_turn", "pcr_set_stream_layout", "pcr_set_hbm_budget", "pcr_stream_layout", "pcr_set_render_variant", "pcr_set_workgroup_parts", "pcr_set_int64_mergeable", "pcr_fence_record", "pcr_fence_wait", "pcr_merge_min_slices", "pcr_resolve_basic_range", "pcr_set_async_upload", "pcr_batches_resident", "pcr_last_frame_batches", "pcr_stream_algorithmic_bytes", "pcr_last_frame_algorithmic_bytes", "pcr_stream_resident_bytes", "pcr_stream_color_format", "pcr_kernel_version", "pcr_get_stream", "pcr_get_device", "pcr_framebuffer_elems", "pcr_framebuffer_capacity", "pcr_device_rgba", "pcr_resolve_hqs_range",
    "pcr_las_begin", "pcr_las_upload", "pcr_las_unload", "pcr_las_batches_loaded", "pcr_render_las", "pcr_resolve_las",
    "pcr_las_algorithmic_bytes", "pcr_gpu_encode_points", "pcr_gpu_encode_free",
]

HOST_SYMBOLS = [
    "pcr_host_last_error", "pcr_host_free", "pcr_encode_points", "pcr_synth_points", "pcr_synth_las_info",
    "pcr_synth_encode", "pcr_morton_key", "pcr_huffman_build", "pcr_pack_chain", "pcr_table_from_dict",
    "pcr_bc1_encode_block", "pcr_bc7_encode_block", "pcr_las_quantize", "pcr_camera_orbit",
]

_hip = None
_host = None


def hip_lib() -> C.CDLL:
    """The HIP library. Raises if it is not built — the product has no other rendering path."""
    global _hip
    if _hip is None:
        path = _build.HIP_LIB
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `python -m pcrhpg24_amd.build` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # One HIP runtime per process: torch ships its own libamdhip64/libhsa-runtime64 under torch/lib with the same
        # SONAMEs as /opt/rocm's. Whichever copy is mapped first serves both torch and this library; with ROCm's mapped
        # first, torch's own device initialisation later fails ("No HIP GPUs are available"). torch is the plumbing for
        # streams and RCCL on the multi-GPU path, so its copy goes first whenever torch is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(path)
        lib.pcr_last_error.restype = C.c_char_p
        lib.pcr_last_error.argtypes = [C.c_void_p]
        lib.pcr_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.pcr_destroy.argtypes = [C.c_void_p]
        lib.pcr_destroy.restype = None
        for n in ("pcr_batches_loaded", "pcr_points_loaded", "pcr_stream_algorithmic_bytes", "pcr_last_frame_algorithmic_bytes", "pcr_stream_resident_bytes", "pcr_las_batches_loaded",
                  "pcr_las_algorithmic_bytes"):
            getattr(lib, n).restype = c_i64
            getattr(lib, n).argtypes = [C.c_void_p]
        for n in ("pcr_device_framebuffer", "pcr_device_rg", "pcr_device_ba"):
            getattr(lib, n).restype = C.c_void_p
            getattr(lib, n).argtypes = [C.c_void_p]
        lib.pcr_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        lib.pcr_synchronize.argtypes = [C.c_void_p]
        lib.pcr_stream_begin.argtypes = [C.c_void_p, C.POINTER(FileHeader), c_i64]
        lib.pcr_upload_batch.argtypes = [C.c_void_p, c_i64, C.c_void_p, C.c_size_t]
        lib.pcr_upload_batches.argtypes = [C.c_void_p, c_i64, c_i64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        lib.pcr_upload_tail.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        lib.pcr_stream_unload.argtypes = [C.c_void_p]
        lib.pcr_set_image_size.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.pcr_clear.argtypes = [C.c_void_p]
        for n in ("pcr_render_basic", "pcr_render_hqs_depth", "pcr_render_hqs_color", "pcr_resolve_basic", "pcr_resolve_hqs"):
            getattr(lib, n).argtypes = [C.c_void_p, C.POINTER(RenderParams)]
        for n in ("pcr_render_las", "pcr_resolve_las"):
            getattr(lib, n).argtypes = [C.c_void_p, C.POINTER(RenderParams)]
        lib.pcr_gpu_encode_points.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.POINTER(LasInfo),
                                              C.c_int, c_i64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(EncodeStats)]
        lib.pcr_gpu_encode_free.argtypes = [C.c_void_p]
        lib.pcr_gpu_encode_free.restype = None
        lib.pcr_las_begin.argtypes = [C.c_void_p, c_i64]
        lib.pcr_las_upload.argtypes = [C.c_void_p, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pcr_las_unload.argtypes = [C.c_void_p]
        lib.pcr_get_stats.argtypes = [C.c_void_p, C.POINTER(RenderStats)]
        lib.pcr_read_framebuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.pcr_read_accum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.pcr_read_rgba.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        lib.pcr_use_external_buffers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pcr_merge_min.argtypes = [C.c_void_p, C.c_void_p]
        lib.pcr_merge_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pcr_flip_sign.argtypes = [C.c_void_p]
        lib.pcr_timing_begin.argtypes = [C.c_void_p]
        lib.pcr_timing_end.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        lib.pcr_measure_hbm.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.pcr_frame_begin.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_int]
        lib.pcr_frame_turn.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.POINTER(RenderParams), C.c_int]
        lib.pcr_kernel_version.restype = C.c_char_p
        lib.pcr_kernel_version.argtypes = []
        lib.pcr_stream_resident_bytes.restype = C.c_int64
        lib.pcr_stream_resident_bytes.argtypes = [C.c_void_p]
        lib.pcr_stream_color_format.restype = C.c_int
        lib.pcr_stream_color_format.argtypes = [C.c_void_p]
        lib.pcr_set_stream_layout.argtypes = [C.c_void_p, C.c_int]
        lib.pcr_set_render_variant.argtypes = [C.c_void_p, C.c_int]
        lib.pcr_set_workgroup_parts.argtypes = [C.c_void_p, C.c_int]
        lib.pcr_framebuffer_private.argtypes = [C.c_void_p]
        lib.pcr_set_int64_mergeable.argtypes = [C.c_void_p, C.c_int]
        lib.pcr_merge_min_slices.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]
        lib.pcr_resolve_basic_range.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_size_t, C.c_void_p]
        lib.pcr_resolve_hqs_range.argtypes = [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        lib.pcr_set_hbm_budget.argtypes = [C.c_void_p, C.c_int64]
        lib.pcr_stream_layout.argtypes = [C.c_void_p]
        lib.pcr_last_frame_algorithmic_bytes.argtypes = [C.c_void_p]
        lib.pcr_last_frame_algorithmic_bytes.restype = C.c_int64
        lib.pcr_framebuffer_capacity.argtypes = [C.c_void_p]
        lib.pcr_framebuffer_capacity.restype = C.c_size_t
        lib.pcr_device_rgba.argtypes = [C.c_void_p]
        lib.pcr_device_rgba.restype = C.c_void_p
        lib.pcr_fence_record.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.pcr_fence_wait.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.pcr_set_async_upload.argtypes = [C.c_void_p, C.c_int]
        lib.pcr_batches_resident.argtypes = [C.c_void_p]
        lib.pcr_batches_resident.restype = C.c_int64
        lib.pcr_last_frame_batches.argtypes = [C.c_void_p]
        lib.pcr_last_frame_batches.restype = C.c_int64
        lib.pcr_kernel_timing_enable.argtypes = [C.c_void_p, C.c_int]
        lib.pcr_kernel_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        _hip = lib
    return _hip


def host_lib() -> C.CDLL:
    global _host
    if _host is None:
        path = _build.HOST_LIB
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `python -m pcrhpg24_amd.build`")
        lib = C.CDLL(path)
        lib.pcr_host_last_error.restype = C.c_char_p
        lib.pcr_host_free.argtypes = [C.c_void_p]
        lib.pcr_host_free.restype = None
        lib.pcr_encode_points.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.POINTER(LasInfo),
                                          C.c_int, c_i64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                          C.POINTER(EncodeStats)]
        lib.pcr_synth_points.argtypes = [c_i64, c_u64, c_i64, c_i64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pcr_synth_las_info.argtypes = [c_i64, c_u64, C.POINTER(LasInfo)]
        lib.pcr_synth_encode.argtypes = [c_i64, c_u64, c_i64, c_i64, c_i64, C.c_int, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_size_t), C.POINTER(EncodeStats)]
        lib.pcr_morton_key.argtypes = [c_u32, c_u32, c_u32, C.POINTER(c_u32), C.POINTER(c_u64)]
        lib.pcr_morton_key.restype = None
        lib.pcr_huffman_build.argtypes = [C.c_void_p, c_i64, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_void_p, C.c_void_p]
        lib.pcr_pack_chain.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, c_i64,
                                       C.POINTER(C.c_void_p), C.POINTER(c_i32), C.POINTER(C.c_void_p), C.POINTER(c_i32),
                                       C.POINTER(C.c_void_p)]
        lib.pcr_table_from_dict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.c_void_p, C.c_void_p]
        lib.pcr_bc1_encode_block.argtypes = [C.c_void_p, C.c_void_p]
        lib.pcr_bc1_encode_block.restype = None
        lib.pcr_bc7_encode_block.argtypes = [C.c_void_p, C.c_void_p]
        lib.pcr_bc7_encode_block.restype = None
        lib.pcr_las_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_i64, C.POINTER(LasInfo), C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.pcr_camera_orbit.argtypes = [c_f64, c_f64, c_f64, C.POINTER(c_f64), C.c_int, C.c_int, c_f64, c_f64, c_f64,
                                         C.POINTER(RenderParams)]
        _host = lib
    return _host


def host_error() -> str:
    return (host_lib().pcr_host_last_error() or b"").decode()
