/*
 * pcr_types.h — plain-old-data types shared by the C ABI (pcr_hip.h), the HIP kernels,
 * the host-side method/resource adapters and the CPU oracle.
 *
 * Every struct is little-endian, naturally aligned, and free of pointers so it can be passed
 * through cgo / ctypes / JNI unchanged.
 *
 * Reference interfaces these mirror (paths relative to the reference checkout):
 *   pcr_gpu_batch      <- struct GPUBatch, modules/huffman_cuda/huffman_kernel_data.h:4-38 (160 B, same field order)
 *   pcr_render_params  <- struct ChangingRenderData + Mat, modules/compute_loop_las_cuda/kernel_data.h:24-26,54-74
 *                         (only the fields the Huffman kernels read; own packing, no CUDA int2 alignment quirk)
 *   pcr_file_header    <- 5 x int64 at the start of a .huffman file, src/preprocess.cpp:1205-1234,
 *                         modules/compute/HuffmanLasLoader.h:57-85
 *   geometry constants <- modules/compute/Resources.h:4-15
 */
#ifndef PCR_TYPES_H
#define PCR_TYPES_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* modules/compute/Resources.h:4-15 */
#define PCR_POINTS_PER_THREAD     64
#define PCR_WORKGROUP_SIZE        1024
#define PCR_CLUSTERS_PER_THREAD   1
#define PCR_POINTS_PER_BATCH      (PCR_POINTS_PER_THREAD * PCR_WORKGROUP_SIZE) /* 65536 */
#define PCR_HUFFMAN_TABLE_SIZE    4096
#define PCR_MAX_CW_LEN            12
#define PCR_CLUSTER_LANES         32     /* the stream is interleaved for 32-lane clusters (src/preprocess.cpp:540-587) */
#define PCR_CLUSTERS_PER_BATCH    (PCR_WORKGROUP_SIZE / PCR_CLUSTER_LANES)     /* 32 */
#define PCR_COLOR_BYTES_PER_BATCH (PCR_POINTS_PER_BATCH / 2)                   /* BC1: 8 B per 16 points */
#define PCR_COLOR_BYTES_PER_BATCH_BC7 PCR_POINTS_PER_BATCH                     /* BC7 mode 6: 16 B per 16 points (COLOR_COMPRESSION == 7) */
/* colour format of a stream (the reference's compile-time COLOR_COMPRESSION, BatchDumpData.h:130-136): told by the size of its records */
#define PCR_COLOR_BC1 1
#define PCR_COLOR_BC7 7
#define PCR_BATCH_FIXED_HEADER    124    /* bytes before start_values in a batch record (include/BatchDumpData.h:60-107) */

/* Zero padding (in 32-bit words) kept behind the encoded / escape streams so the reference's tail
 * over-reads stay inside the allocation. The reference pads EncodedData with
 * 4*1024 bytes (modules/compute/HuffmanLasLoader.cpp:39-41) and SeparateData with nothing. */
#define PCR_ENCODED_PAD_WORDS     1024
#define PCR_SEPARATE_PAD_WORDS    256
/* Zero words kept behind each pad that no upload ever writes: device loads clamp their index into them. */
#define PCR_GUARD_WORDS           8

/* u64 words allocated behind each of a context's own framebuffers (and u32 behind its RGBA8 image): room for the frame to
 * be cut into N equal slices (N <= 128) by the sliced multi-GPU exchange of include/pcr_dist.h. */
#define PCR_FRAME_PAD_ELEMS       256

#define PCR_BACKGROUND_COLOR      0x00443322u /* resolve.cu:166 */

/* struct GPUBatch (160 bytes) */
typedef struct pcr_gpu_batch {
    float   min_x, min_y, min_z;
    float   max_x, max_y, max_z;
    double  scale_x, scale_y, scale_z;
    double  offset_x, offset_y, offset_z;
    double  las_min_x, las_min_y, las_min_z;
    double  las_max_x, las_max_y, las_max_z;
    int64_t encoding_batch_offset;   /* word offset into EncodedData   */
    int64_t separate_batch_offset;   /* word offset into SeparateData  */
    int64_t decoder_table_offset;    /* entry offset into the tables   */
    int64_t cluster_sizes_offset;    /* entry offset into ClusterSizes */
    int64_t max_cw_len;              /* 12 */
} pcr_gpu_batch;

/* struct XYZBatch of the 10-10-10 path (modules/compute_loop_las_cuda/kernel_data.h:4-22, 64 bytes): bounding box of
 * the batch relative to the cloud's box minimum, and how many of its 65 536 slots hold real points. */
typedef struct pcr_xyz_batch {
    int32_t state;
    float   min_x, min_y, min_z;
    float   max_x, max_y, max_z;
    int32_t num_points;
    int32_t padding[8];
} pcr_xyz_batch;

/* First 40 bytes of a .huffman file. */
typedef struct pcr_file_header {
    int64_t num_points;      /* after padding; multiple of 65536 */
    int64_t num_batches;
    int64_t encoded_bytes;   /* sum over batches of 4*len(encoding)  */
    int64_t separate_bytes;  /* sum over batches of 4*len(separate)  */
    int64_t cluster_bytes;   /* 128 * num_batches */
} pcr_file_header;

/* Per-frame parameters (the fields of ChangingRenderData the Huffman kernels read).
 * Matrices are 4 rows of 4 floats: pos[r] = dot(row r, (x,y,z,1)). The reference host stores
 * glm::transpose(proj*view*world) there (modules/huffman_hqs/huffman_hqs.h:167-169), i.e. exactly rows. */
typedef struct pcr_render_params {
    float   transform[16];            /* uTransform  = proj * view * world */
    float   world_view[16];           /* uWorldView  = view * world        */
    float   proj[16];                 /* uProj                              */
    int32_t width, height;            /* uImageSize                         */
    int32_t points_per_thread;        /* uPointsPerThread, must be 64       */
    int32_t lod_percent;              /* uPointFormat = (int)(Debug::LOD*100), default 10 */
    int32_t enable_frustum_culling;   /* uEnableFrustumCulling, default 1   */
    int32_t show_num_points;          /* debug payload modes (render.cu:289-294) */
    int32_t colorize_chunks;
    int32_t reserved;
} pcr_render_params;

/* Counters a render call reports back (what the metric is computed from: SURVEY 8d). */
typedef struct pcr_render_stats {
    int64_t batches_total;
    int64_t batches_culled;
    int64_t points_iterated;   /* sum over non-culled batches of 1024 * NumPointsToRender */
    int64_t batches_double;    /* batches that took the double-precision dequantisation path */
} pcr_render_stats;

/* Number of u64 elements a framebuffer of w x h must hold: ndc == 1.0 maps to column w / row h
 * (SURVEY Appendix C.2), so pixel ids reach w*(h+1). */
static inline size_t pcr_fb_elems(int w, int h) { return (size_t)w * (size_t)(h + 1) + 1; }

#ifdef __cplusplus
}
#endif
#endif /* PCR_TYPES_H */
