"""pcrhpg24_amd — MI355X-native Huffman point-cloud decode + rasterize path (hot path of rahul-goel/pcrhpg24).

Layout:  csrc/ (HIP kernels, C ABI, CPU encoder)  ·  host.py (the reference's Method/Resource surface over the C ABI)
         ·  dist.py (batch sharding + framebuffer merge over torch.distributed)  ·  build.py (in-tree builds)
"""
from .host import (ComputeHuffman, ComputeLasData, ComputeLoopLasCUDA, Context, Debug, HuffmanFile, HuffmanHQS, HuffmanLasData, HuffmanMemIter, Method, PcrError,  # noqa: F401
                   Renderer, Resource, Runtime, camera_orbit, encode_points, kernel_version, synth_encode, synth_las_info, synth_points, las_quantize, read_las)
from ._native import FileHeader, LasInfo, RenderParams, RenderStats, XyzBatch, fb_elems  # noqa: F401
