// pcr_kernels.hip.h — hand-written HIP kernels for gfx950 (MI355X, wave64). Included by pcr_api.hip only.
//
// Compile flags that are part of the numeric contract: -ffp-contract=off (every FMA
// below is spelled __fmaf_rn/__fma_rn), default correctly-rounded f32 division and sqrt, no fast-math.
//
// Kernels
//   k_lod_prepass   eight lanes per batch: frustum cull + LOD (render.cu:333-379) -> lod word per batch + stats, the dense lists
//                   of the batches to draw (ballot + prefix compaction), and the LDS framebuffer windows of each (plan_windows)
//   k_transcode     once per loaded batch: the reference's lockstep walk over the cluster-interleaved stream
//                   (render.cu:404-451), recording per chain the words it receives -> lane-major stream, 40-bit point windows,
//                   packed decoder table, colour blocks in segment-major order
//   k_bounds        once per loaded batch: where the batch's chains fall apart into spatial clusters (runs of chains + boxes)
//   k_render<MODE>  every frame, one chain per lane -- two 512-thread workgroups per batch (PARTS = 2, the default) or the reference's
//                   one of 1024 (PARTS = 1): decodes the chain of <= 64 points from its own bits with the batch's decoder table in
//                   LDS, then projects and scatters every point (render.cu:383-540, huffman_hqs/depth.cu, huffman_hqs/render.cu;
//                   MODE 3: the colour pass over BC7 mode-6 colours)
//   k_las_*         the 10-10-10 method (modules/compute_loop_las_cuda)
//   k_resolve_*     framebuffer -> RGBA8 (resolve.cu:149-191, huffman_hqs/resolve.cu:2-47)
//   k_merge_* / k_flip_sign  multi-GPU partial-framebuffer merges
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "pcr_types.h"

namespace pcr {

// lod word written by the prepass
constexpr uint32_t LOD_NPR_MASK = 0xFFu;
constexpr uint32_t LOD_DOUBLE   = 0x100u;
constexpr uint32_t LOD_CULLED   = 0x200u;

enum { MODE_BASIC = 0, MODE_HQS_DEPTH = 1, MODE_HQS_COLOR = 2, MODE_HQS_COLOR_BC7 = 3 };   // (3: the colour pass over BC7 mode-6 colours)

// k_render's LDS plan (whole-batch workgroups: 76 KiB per 1024 threads -> two per CU; half-batch workgroups: see DYN_LDS_BYTES_HALF)
constexpr int CHUNK_WORDS    = 64;               // stream staging granule per cluster: 32 lanes x 2 words (8-byte loads)
constexpr int RING_WORDS     = 2 * CHUNK_WORDS;  // per cluster                                   -> 16 KiB
// 60 KiB of k_render's LDS are shared between the batch's escape words and its framebuffer window, divided per batch: the
// pool takes what the batch's escapes need (up to ESC_POOL_WORDS, which leaves 1024 window pixels), the window gets the rest.
// The benchmark stream's batches have 3831..7037 escape words: windows of 4090..5760 pixels. A fixed 24 + 36 KiB split put 6 %
// of those batches on the checked escape path, 1.5x slower each, and they finished last on their CUs (+17 % per launch).
// DYN_LDS_BYTES_BIG: the same kernels launched with 140 KiB of it -- one workgroup per CU, windows of up to ~17 000 pixels --
// for frames whose batches' rectangles outgrow the small windows (4096x4096 over 1526 batches: most points would go through
// global pre-reads and global atomics; with the big windows k_render 0.564 -> 0.48 ms). At 1080p it is 40 % slower: half
// the waves per SIMD. The host chooses per frame (launch_render); the size travels in RenderArgs::dyn_lds_bytes.
constexpr int DYN_LDS_BYTES  = 60 * 1024;
constexpr int DYN_LDS_BYTES_BIG = 140 * 1024;
constexpr int ESC_POOL_WORDS = 13312;            // most escape words of a batch the pool ever holds -> 52 KiB
constexpr int ESC_SLACK      = 64;               // words behind the batch's own escapes kept in the pool as well
// HALF-BATCH WORKGROUPS (round 4; k_render<..., PARTS = 2>): a workgroup of 512 threads draws chains [512 part, 512 part + 512) of
// a batch, four such workgroups per CU: 16 KiB table + 24 KiB of dynamic LDS each (the part's own escape words + windows for the
// part's rectangle, about half the batch's). Same eight waves per SIMD; what changes is the granularity at which LDS and wave
// slots turn over: a workgroup's waves leave their point loop up to 12 us apart (oldest-first arbiter), its LDS is held from
// set-up to merge, and a launch of 1526 whole batches is three lock-step generations. Departs from render.cu:328 (one block per
// batch) and :383-395 (the table staged once per block: here twice, the second read from L2 -- the two halves of a batch are
// mapped to the same XCD, see k_render).
constexpr int DYN_LDS_BYTES_HALF     = 24 * 1024;    // four workgroups per CU (4 x (16 + 24) KiB = 160 KiB)
constexpr int DYN_LDS_BYTES_HALF_BIG = 62 * 1024;    // two per CU: large images (the 140 KiB configuration's counterpart)
constexpr int ESC_POOL_WORDS_HALF    = 5120;         // most escape words of a half-batch its pool ever holds -> 20 KiB
constexpr int MAX_PARTS = 2;
__host__ __device__ constexpr int esc_pool_max(int parts) { return parts == 1 ? ESC_POOL_WORDS : ESC_POOL_WORDS_HALF; }
constexpr int WIN_PIXELS     = 4096;             // nominal window (a batch with 7104 escape words); the 10-10-10 kernel's fixed one
constexpr int WIN_PIXEL_BYTES     = 8;           // basic / HQS depth: the u64 framebuffer word
constexpr int WIN_PIXEL_BYTES_HQS = 20;          // HQS colour: {RG u64, BA u64, depth u32}
constexpr int WIN_PIXELS_MAX = DYN_LDS_BYTES / WIN_PIXEL_BYTES;         // of the small configuration
// How many of a batch's escape words k_render keeps in LDS: all of them plus ESC_SLACK words that follow them in memory
// (the reference's tail over-reads), or as many as the pool holds. A batch with more than that is flagged
// (BF_GENERIC_SLOW_PATH) and its chains check every escape index: the first ESC_POOL_WORDS still come from LDS, only
// the rest from global memory. (Pooling none of an oversized batch's escapes made that batch twice as slow, and a few
// such batches made the whole launch 14 % longer: they finish last on their CUs.)
// (`esc_count`: the escape words of the workgroup's own chains -- the whole batch's, or one part's)
__device__ __forceinline__ uint32_t esc_pool_words(uint32_t esc_count, int parts = 1)
{
    return min(esc_count + (uint32_t)ESC_SLACK, (uint32_t)esc_pool_max(parts));
}
__device__ __forceinline__ uint32_t esc_pool_bytes(uint32_t pool_words) { return (pool_words * 4u + 15u) & ~15u; }
// pixels the batch's LDS window can hold next to its escape pool
__device__ __forceinline__ int window_capacity(uint32_t esc_count, int pixel_bytes, uint32_t dyn_lds_bytes, int parts = 1)
{
    // (one pixel less than fits: the slot behind the last window pixel is k_render's dummy slot, see there)
    return (int)((dyn_lds_bytes - esc_pool_bytes(esc_pool_words(esc_count, parts))) / (uint32_t)pixel_bytes) - 1;
}

// Lane-major copy of the word stream (k_transcode): row r holds the r-th word each of the batch's 1024 chains consumes.
// A chain consumes 2 words up front and one per 32 decoded bits (<= 192 x 12 / 32 = 72), and k_render reads four ahead.
constexpr int LW_ROWS        = 80;
constexpr uint32_t LW_ROW_BYTES = PCR_WORKGROUP_SIZE * 4;
// PCR_LAYOUT_WORDS keeps that copy compact (k_pack_words): per wave (64 chains) only the rows its longest chain consumed, 256
// bytes each, one block per batch. The benchmark stream's chains take 43.6 words on average, a wave's longest 46.7, against
// the 80 rows of the transcode's scratch form: 2.9 B per point resident instead of 5.
constexpr uint32_t LWC_ROW_BYTES = 64 * 4;
// Rows allocated behind the last wave's rows of a segment: a chain requests the two words behind its view for every point, up to four rows
// past the last row its wave's longest chain consumed (2 preloaded + every retired word + 1) -- words nobody looks at (a code's table
// entry does not depend on the bits behind the code), read from the next wave's rows or from this pad instead of being clamped
constexpr uint32_t LWC_PAD_ROWS = 8;
constexpr int LWC_WAVES = PCR_WORKGROUP_SIZE / 64;          // 16 blocks of rows per batch + one entry for the end
// Point windows (k_transcode, layout PCR_LAYOUT_POINT_WINDOWS): for point i of every chain the 40 bits of the chain's own
// bit stream that start at the point's first bit -- its three symbols (<= 36 bits) lie inside. Stored as two planes per
// batch: PW_ROWS rows of 1024 x u32 (bits 0..31 of the window) followed by PW_ROWS rows of 1024 x u8 (bits 32..39). Row i is
// read at point i by every lane of k_render (256 + 64 contiguous bytes per wave), with no word queue to maintain: 5 bytes per
// point of HBM instead of the ~3 the packed words take -- bytes traded for instructions. The key of a point's FIRST symbol is
// the top 12 bits of its own window, which do not depend on the point before it: k_render requests that table entry a whole
// point ahead, and only two of a point's three table reads are left on the dependent chain. (Round 1 stored 64 bits per
// point, round 2's first form 48: 36 + the look-ahead of the next point's first symbol, which the next window holds anyway.)
// PW_GUARD_BYTES: k_render requests two rows ahead, the last batch's low plane is followed by this much slack.
constexpr int PW_ROWS        = PCR_POINTS_PER_THREAD;
constexpr uint32_t PW_HI_ROW_BYTES = PCR_WORKGROUP_SIZE * 4, PW_LO_ROW_BYTES = PCR_WORKGROUP_SIZE * 1;
constexpr uint32_t PW_HI_BYTES = PW_ROWS * PW_HI_ROW_BYTES;                 // offset of the low plane inside a batch's block
constexpr uint32_t PW_BATCH_BYTES = PW_ROWS * (PW_HI_ROW_BYTES + PW_LO_ROW_BYTES);   // 320 KiB
constexpr uint32_t PW_GUARD_BYTES = 2 * PW_HI_ROW_BYTES;
enum { LAYOUT_WORDS = 0, LAYOUT_POINT_WINDOWS = 1 };
// batch_flags: set when some chain of the batch can meet an in-table value outside the packed entry's range or read an
// escape word outside k_render's LDS pool (k_transcode walks all 192 symbols of every chain, garbage tails included)
constexpr uint32_t BF_GENERIC_SLOW_PATH = 1u;      // ... of a whole-batch workgroup (PARTS == 1)
constexpr uint32_t BF_GENERIC_SLOW_PATH_HALF = 2u; // ... of a half-batch workgroup (PARTS == 2: its pool holds its own part's escapes only)
__host__ __device__ constexpr uint32_t bf_generic(int parts) { return parts == 1 ? BF_GENERIC_SLOW_PATH : BF_GENERIC_SLOW_PATH_HALF; }
// packed table entry: byte 0 = len, bits 31:10 = the value as a signed 22-bit number -- (int32)entry >> 10 is the delta, no
// bias to remove -- except that the most negative one, TE_SLOW_VALUE, is reserved: "the value is not in the entry" (an escape,
// bit 8, or an in-table value outside (-2^21, 2^21), bit 9). Cost model behind the layout (tools/exp/instr_rate2.hip, gfx950):
// v_ashrrev_i32 / v_add_u32 / v_sub_u32 / v_and_b32 on VGPRs and constants issue in ~2.3 cycles per wave64, anything VOP3,
// SDWA, 64-bit, packed, compare, convert or with an SGPR operand in ~4.2.
constexpr uint32_t TE_LEN = 0xFFu, TE_ESCAPE = 0x100u, TE_WIDE = 0x200u;
constexpr int TE_VALUE_SHIFT = 10;
constexpr int32_t TE_SLOW_VALUE = -(1 << 21);

// Device-side view of the loaded stream (own layout; the reference keeps nine flat CuBuffers,
// HuffmanLasLoader.h:39-47). Tables are stored as int32 values + int8 lengths (the reference narrows the
// length to `char` in-kernel, render.cu:393).
struct StreamView {
    const pcr_gpu_batch *batches;
    const int32_t  *start_values;     // [nB*1024*3]
    const uint32_t *encoded;          // [encoded_words]   (includes zero pad)
    const int32_t  *separate;         // [separate_words]  (includes zero pad)
    const int32_t  *separate_sizes;   // [nB*1024]
    const int32_t  *table_values;     // [nB*4096]
    const int8_t   *table_lens;       // [nB*4096]
    const int32_t  *cluster_sizes;    // [nB*32]
    const uint8_t  *colors;           // [nB*32768] BC1 (or [nB*65536] BC7 mode 6) as the file has them: block 4 t + s = points 16 s .. 16 s + 15 of chain t
    const uint8_t  *colors_t;         // the same blocks, k_transcode's order: [batch][segment s][chain t] -- the 1024 lanes of
                                      // k_render read the blocks of one segment side by side (as the file has them a wave's 64
                                      // blocks lie 32 B apart and every 128-byte line was fetched once per segment: 4x)
    uint32_t color_block_bytes;       // 8 (BC1) / 16 (BC7)
    const uint32_t *lane_words;       // k_transcode's scratch: [chunk][LW_ROWS][1024] lane-major stream
    const uint32_t *const *lw_block;  // [nB] PCR_LAYOUT_WORDS: the batch's compact copy (k_pack_words), or NULL (layout)
    const uint32_t *lw_wave_row;      // [nB * (LWC_WAVES + 1)] first row of each wave's rows inside the batch's block, and the end
    const uint32_t *batch_flags;      // [nB] BF_* bits written by k_transcode
    const uint32_t *packed_table;     // [nB*4096] k_render's table entries, packed by k_transcode
    const uint8_t  *point_windows;    // [nB * PW_BATCH_BYTES + PW_GUARD_BYTES] or NULL (layout), written by k_transcode
    const uint32_t *batch_runs;       // [nB * RUN_RECORDS * RUN_WORDS] k_bounds: where a batch's (a half-batch's) chains fall apart into spatial clusters
    int64_t encoded_words;
    int64_t separate_words;
    int64_t num_batches;
    int64_t batch_index_base;
};

// (esc_mid: the escape words of chains 0..511 -- where a half-batch workgroup's share of the batch's escapes begins / ends)
struct DrawRec { uint32_t b, lod, esc_total, esc_mid; int64_t sep_off; int64_t reserved2; };     // 32 bytes

struct FrameView {
    uint64_t *fb;
    uint64_t *rg;
    uint64_t *ba;
    uint32_t  fb_elems;
    // Dirty tiles: one byte per TILE_W x TILE_H pixels of the framebuffer, set for every tile a framebuffer word may be written
    // in: the prepass marks the tiles under the screen rectangle of every drawn batch's bounding box (every point of the batch
    // lands inside, in a window or not; a batch without such a rectangle -- its box reaches behind the camera -- sets the word
    // "everything" instead); a point that lands OUTSIDE that rectangle all the same -- the garbage tail of a chain, SURVEY B.4: a
    // thousandth of the points, anywhere on the screen -- has its tile marked by k_render, in its off-window branch.
    // The end of the frame (k_frame_turn_tiles) resolves and clears the marked tiles only -- at 4096x4096 the fused resolve + clear
    // is 57 us of a 420 us frame streaming 335 MB whether a pixel was touched or not. NULL: nobody keeps track (external buffers,
    // the 10-10-10 method), the whole frame is resolved and cleared. (Marking EVERY off-window point's tile in k_render's
    // point loop, with a load in front of the store, cost the loop 6 % for code it practically never runs; profiles/r03_experiments.md.)
    uint8_t  *tiles;                  // [ntiles]
    uint32_t *tiles_all;              // "everything": holds tiles_epoch if so (an epoch per frame: the word never has to be zeroed)
    uint32_t  tiles_x, tiles_epoch, tiles_total;
};
constexpr uint32_t TILE_W_SHIFT = 6, TILE_H_SHIFT = 4;     // 64 x 16 pixels: a tile row is 512 contiguous bytes of the framebuffer

// A batch's points are 65 536 consecutive points of the Morton order, chain t = points 64 t .. 64 t + 63. Where the curve
// jumps, the batch is two (or more) compact clusters far apart: at 1080p 25 of the benchmark's 1526 batches have bounding
// rectangles of up to 600 x 80 pixels for a few hundred touched ones, which no LDS window holds. Those batches took ~4x as
// long as the others -- their points went through global pre-reads and atomics -- and, being stragglers, cost the launch 9 %
// (4096x4096: a fifth of the batches, 38 %). So, once per loaded batch (k_bounds, camera independent): the three largest gaps
// between consecutive chains cut the batch into RUNS runs of chains, each with its own bounding box.
constexpr int RUNS = 4;
// One record per workgroup shape and part -- record 0: the whole batch, records 1 and 2: its halves (chains 0..511, 512..1023), each
// cut at ITS three largest jumps:
constexpr int RUN_WORDS = 4 + RUNS * 6 + 6; // {first chain of run 1, 2, 3, 0}, per run {min xyz, max xyz} as floats, then the box of ALL the part's chains
constexpr int RUN_RECORDS = 1 + MAX_PARTS;  // per batch
__host__ __device__ constexpr int run_record(int parts, int part) { return parts == 1 ? 0 : 1 + part; }
// Per frame (prepass, a placement hint: any plan gives the same frame): one LDS framebuffer window per run, or one for the
// whole batch when the runs' rectangles overlap anyway. first[r] = first chain of run r + 1 (1024: no such run); window r
// sits behind windows 0 .. r-1 in the array of window pixels.
struct WinPlan {
    uint32_t xy[RUNS], wh[RUNS];      // {x0 | y0<<16}, {w | h<<16}; w == 0: none
    uint32_t first[RUNS - 1];
    uint32_t mostly_outside;          // 1: the windows hold less than half of the runs' rectangles -- most points of the batch will land outside
                                      // them -- AND the same goes for most of the 32 batches around it (the prepass workgroup's vote): k_render
                                      // then pre-reads the framebuffer word of such a point, see project_request
    uint32_t whole_xy, whole_wh;      // the screen rectangle of the batch's own bounding box, in which the prepass marked the dirty tiles
                                      // (FrameView::tiles); w == 0: it has none, or no tiles are kept
};

struct RenderArgs {
    pcr_render_params p;
    StreamView s;
    FrameView f;
    uint32_t *lod;            // [nB]
    WinPlan *win;             // [nB * parts] LDS framebuffer windows per workgroup, for pixels of win_pixel_bytes
    WinPlan *win_hqs;         // prepass only: if not NULL, a second plan for the 20-byte pixels of the HQS colour pass is written here
                              // (the depth pass's prepass serves the colour pass of the same frame: one prepass per frame)
    pcr_render_stats *stats;  // device: one partial record per prepass workgroup
    // Dense lists of the batches k_render has to draw (frustum-culled batches and batches whose level of detail is zero
    // points are left out), compacted by the prepass in two levels that keep the file's (Morton) order: every prepass
    // workgroup compacts its PREPASS_BATCHES batches with a wave ballot + prefix count into order[kind][wg * PREPASS_BATCHES ..]
    // and writes how many there are to chunk_count[kind][wg]; k_render's workgroup x finds its batch with a wave-wide
    // prefix sum over the chunk counts. kind 0: ordinary batches, kind 1: those flagged BF_GENERIC_SLOW_PATH (drawn by the
    // checked variant of the kernel). No atomics, nothing to zero between frames. (A first version appended the chunks
    // with one atomic per workgroup: the chunks then land in arrival order, and the close-up frame, whose heavy batches
    // lead the file, ran 25 % longer with them shuffled.)
    struct DrawRec *order;    // [2][order_stride]
    uint32_t *chunk_count;    // [WORK_CLASSES + 1][PCR_MAX_PREPASS_WORKGROUPS]: the ordinary list's classes, then the checked list
    uint32_t order_stride;    // prepass workgroups * PREPASS_BATCHES
    // Longest first: with a level of detail the batches of a frame decode 1..64 points per chain, a workgroup lives 20..55 us, and in
    // the file's order the launch ends with a long tail of half-empty CUs (LOD 10 %: 153 us where the sum of the lifetimes over 512
    // slots is 106). The ordinary list is therefore drawn class by class -- WORK_CLASSES classes of points per chain, heaviest
    // first, the file's order inside a class: a chunk's records are stored class after class, chunk_count[class][chunk] counts them,
    // and k_render's scan walks the classes in turn. A frame at LOD 100 % has one class and scans as before. work_classes: 1 for
    // streams of more than 8192 batches (a scan round per 2048 batches AND class would cost their light workgroups more than it saves).
    uint32_t work_classes;
    // A list entry is a record, not just the batch's index: what k_render's workgroup needs to know before it can request the
    // batch's data (level-of-detail word, escape count, escape offset). Its set-up is then three dependent memory levels (chunk
    // counts -> record -> data) instead of four (... -> list entry -> lod / batch header / escape count -> data): 7 us of a
    // workgroup's 61 us life, spent with its sixteen wave slots idle (profiles/r03_experiments.md). (A fourth step -- the prepass
    // workgroup that finishes last moves the records into one dense array, so that workgroup x starts from ONE load -- was built
    // and measured: k_render -2 %, but 9 us more on the critical path of the launch that carries the prepass. Dropped.)
    int variant_hqs;          // LOD expression variant
    int parts;                // workgroups per batch of the following k_render launches (1: 1024 threads, 2: half-batches of 512): `win` /
                              // `win_hqs` then hold `parts` plans per batch ([b * parts + part]), the lists' "checked" class follows bf_generic(parts)
    int win_pixel_bytes;      // what a window pixel of the following k_render<MODE> takes in LDS (WIN_PIXEL_BYTES*)
    uint32_t dyn_lds_bytes;   // dynamic LDS of the following k_render launch: DYN_LDS_BYTES or DYN_LDS_BYTES_BIG
};

#ifdef PCR_EXP_TIMELINE   /* experiment: wall-clock stamps (100 MHz) of the prepass block's phases, rows 7000 + block of g_timeline */
extern __device__ unsigned long long g_timeline[8192 * 8];
#define PCR_PTL(slot) do { if (threadIdx.x == 0) g_timeline[(size_t)(7000 + block * 2 + (LISTS ? 0 : 1)) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define PCR_PTL(slot) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// wave64 inclusive prefix sum in six DPP adds (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then row_bcast 15 / 31 across
// them): no LDS, no wait. (__shfl_up compiles to ds_bpermute_b32: the scans at the top of k_render -- one per class of the list,
// every wave for itself -- were seven dependent LDS round trips each, ~1 us of a workgroup's set-up per class.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1 (lanes without a source add 0)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);      // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}
// value of lane `lane` (uniform), as a scalar
__device__ __forceinline__ uint32_t wave_read_lane(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }

// ------------------------------------------------------------------------------------------------
// strict-float helpers (helper_math.h semantics, Appendix C.1)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot4(const float *r, float x, float y, float z, float w)
{
    return __fmaf_rn(r[3], w, __fmaf_rn(r[2], z, __fmaf_rn(r[1], y, r[0] * x)));
}

__device__ __forceinline__ bool plane_accepts(float x, float y, float z, float w, const float *bmin, const float *bmax)
{
    float nl = sqrtf(__fmaf_rn(z, z, __fmaf_rn(y, y, x * x)));      // createPlane, render.cu:239-246
    float nx = x / nl, ny = y / nl, nz = z / nl, c = w / nl;
    float vx = nx > 0.0f ? bmax[0] : bmin[0];                       // :261-264
    float vy = ny > 0.0f ? bmax[1] : bmin[1];
    float vz = nz > 0.0f ? bmax[2] : bmin[2];
    float d = __fmaf_rn(nz, vz, __fmaf_rn(ny, vy, nx * vx)) + c;    // :235-237
    return !(d < 0.0f);
}

// Frame statistics: every prepass workgroup sums its batches in LDS and writes ONE partial record (no global atomics,
// nothing to zero beforehand); pcr_get_stats adds the partials of the last launch. PCR_STATS_PARTIALS bounds the grid.
#ifndef PCR_VOTE_NUM
#define PCR_VOTE_NUM 9u      // the prepass workgroup's vote (WinPlan::mostly_outside): at least NUM / DEN of its drawn batches. (With 1 / 2 the
#define PCR_VOTE_DEN 10u     // batches in front of a close-up camera, neighbours in the file, voted themselves in: +6 %, +20 % with culling.)
#endif
constexpr int PREPASS_THREADS = 256;
constexpr int PCR_MAX_PREPASS_WORKGROUPS = 2048;        // ceil(65535 batches / 32 batches per prepass workgroup)
constexpr int WORK_CLASSES = 4;                         // of the ordinary list, by points per chain: 49..64, 33..48, 17..32, 1..16 (RenderArgs::work_classes;
                                                        // 2, 3 and 4 classes measure alike, 8 cost the light workgroups' scans more than they gain)
__device__ __forceinline__ void commit_stats(const pcr_render_stats &mine, pcr_render_stats *partials, uint32_t index)
{
    __shared__ unsigned long long s_sum[4];
    if (threadIdx.x < 4) s_sum[threadIdx.x] = 0;
    __syncthreads();
    if (mine.batches_total)   atomicAdd(&s_sum[0], (unsigned long long)mine.batches_total);
    if (mine.batches_culled)  atomicAdd(&s_sum[1], (unsigned long long)mine.batches_culled);
    if (mine.points_iterated) atomicAdd(&s_sum[2], (unsigned long long)mine.points_iterated);
    if (mine.batches_double)  atomicAdd(&s_sum[3], (unsigned long long)mine.batches_double);
    __syncthreads();
    if (threadIdx.x == 0) {
        pcr_render_stats r;
        r.batches_total = (int64_t)s_sum[0]; r.batches_culled = (int64_t)s_sum[1];
        r.points_iterated = (int64_t)s_sum[2]; r.batches_double = (int64_t)s_sum[3];
        partials[index] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// prepass: cull + LOD per batch
// ------------------------------------------------------------------------------------------------
// Eight lanes per batch: lane k tests frustum plane k (k < 6) and projects bounding-box corner k for the window
// rectangle; the LOD arithmetic is uniform. One lane doing all of it serially made this launch 9 us of every frame.
constexpr int PREPASS_LANES = 8;
constexpr int PREPASS_BATCHES = PREPASS_THREADS / PREPASS_LANES;     // batches per workgroup
struct PlanIn;
__device__ __forceinline__ uint32_t lod_prepass_batch(const RenderArgs &a, int64_t b, int lane, pcr_render_stats &st, bool publish);
__device__ __forceinline__ bool plan_windows(const RenderArgs &a, int64_t b, int part, int j, const PlanIn &in);

// What plan_windows reads from memory, requested at the top of the prepass block together with everything else the block
// reads: the block sits on the critical path of every frame (it shares a launch with the resolve + clear of the frame before:
// 7 us of work at 1080p) and was a chain of five dependent round trips (batch header -> the lod word read back -> list record ->
// runs of chains -> the window plan read back): 12-14 us. Now: one round of loads, values handed on through LDS and registers.
struct PlanIn {
    uint32_t esc_count;             // escape words of the (batch, part)'s chains
    uint32_t cut;                   // first chain of run r + 1 (runs[r]; unused for the last run)
    float box[6], all[6];           // the run's box, the box of all the part's chains (k_bounds)
    float gmin[3], gmax[3];         // the batch's own bounding box, relative to las_min (GPUBatch)
};
__device__ __forceinline__ PlanIn plan_preload(const RenderArgs &a, int64_t b, int part, int r)
{
    PlanIn in;
    const int32_t *ssz = a.s.separate_sizes + (size_t)b * 1024;
    const uint32_t mid = (uint32_t)ssz[511], total = (uint32_t)ssz[1023];
    in.esc_count = a.parts == 1 ? total : part == 0 ? mid : total - mid;
    const uint32_t *runs = a.s.batch_runs + ((size_t)b * RUN_RECORDS + run_record(a.parts, part)) * RUN_WORDS;
    in.cut = runs[min(r, RUNS - 2)];
    const float *box = reinterpret_cast<const float *>(runs + 4 + r * 6), *ab = reinterpret_cast<const float *>(runs + 4 + RUNS * 6);
#pragma unroll
    for (int k = 0; k < 6; ++k) { in.box[k] = box[k]; in.all[k] = ab[k]; }
    const pcr_gpu_batch *g = a.s.batches + b;
    const float lm[3] = { (float)g->las_min_x, (float)g->las_min_y, (float)g->las_min_z };
    in.gmin[0] = g->min_x - lm[0]; in.gmin[1] = g->min_y - lm[1]; in.gmin[2] = g->min_z - lm[2];
    in.gmax[0] = g->max_x - lm[0]; in.gmax[1] = g->max_y - lm[1]; in.gmax[2] = g->max_z - lm[2];
    return in;
}

// The prepass of a chunk of PREPASS_BATCHES batches is done by TWO workgroups of the launch, side by side (round 4): one writes what
// k_render's scan reads (lod words, statistics, the compacted lists), the other the window plans, the dirty tiles and the vote. Both
// work out the chunk's cull / LOD decisions for themselves (a microsecond of arithmetic on CUs that have nothing else to do): the
// block is a chain of cold starts -- kernel arguments, instruction cache, one round of loads -- and sits on every frame's critical
// path; as one workgroup it took 8.6-10.1 us from its start (phase stamps: tools/exp/prepass_timeline.py), of which the plans 2.8-4.5.
template <bool LISTS>
__device__ __forceinline__ void lod_prepass_chunk(const RenderArgs &a, uint32_t block)
{
    const int64_t b = ((int64_t)block * PREPASS_THREADS + threadIdx.x) / PREPASS_LANES;
    const int lane = (int)(threadIdx.x % PREPASS_LANES);
    pcr_render_stats st = {0, 0, 0, 0};
    PCR_PTL(0);
    // the group's lod word (LOD_*), handed to the compaction / to the window plan through LDS (LOD_CULLED: no such batch)
    __shared__ uint32_t s_lod[PREPASS_BATCHES];
    // ---- every load of the block that does not depend on a result of the block, requested before anything is computed ----
    // the record lanes (first wave, one per batch of the block): what a list record holds besides the lod word
    const uint32_t bb = block * PREPASS_BATCHES + threadIdx.x;
    uint32_t rec_flags = 0, rec_esc_total = 0, rec_esc_mid = 0;
    int64_t rec_sep_off = 0;
    if (LISTS && threadIdx.x < PREPASS_BATCHES && (int64_t)bb < a.s.num_batches) {
        rec_flags = a.s.batch_flags[bb];
        rec_esc_total = (uint32_t)a.s.separate_sizes[(size_t)bb * 1024 + 1023];
        rec_esc_mid = (uint32_t)a.s.separate_sizes[(size_t)bb * 1024 + 511];
        rec_sep_off = a.s.batches[bb].separate_batch_offset;
    }
    // the plan lanes: RUNS lanes per batch and part, one per run of chains
    static_assert(PREPASS_BATCHES * RUNS * MAX_PARTS <= PREPASS_THREADS, "one round");
    const uint32_t parts = (uint32_t)a.parts, plan_lanes = RUNS * parts;
    const uint32_t slot = threadIdx.x / plan_lanes, part = (threadIdx.x / RUNS) % parts;
    const int64_t pb = (int64_t)block * PREPASS_BATCHES + slot;
    const bool plan_lane = !LISTS && threadIdx.x < PREPASS_BATCHES * plan_lanes && pb < a.s.num_batches;
    PlanIn pin = {};
    if (plan_lane) pin = plan_preload(a, pb, (int)part, (int)(threadIdx.x % RUNS));

    PCR_PTL(1);
    if (lane == 0) s_lod[threadIdx.x / PREPASS_LANES] = LOD_CULLED;
    if (b < a.s.num_batches) {
        const uint32_t lod = lod_prepass_batch(a, b, lane, st, LISTS);                          // uniform per 8-lane group
        if (lane == 0) s_lod[threadIdx.x / PREPASS_LANES] = lod;
    }
    PCR_PTL(2);
    if (LISTS) {
        commit_stats(st, a.stats, block);                   // (barriers inside: s_lod is complete afterwards)
        PCR_PTL(3);
        // first level of the compaction: one ballot and one prefix count per list, in the workgroup's first wave
        if (threadIdx.x < 64) {
            const uint32_t lod = threadIdx.x < PREPASS_BATCHES ? s_lod[threadIdx.x] : LOD_CULLED;
            // what k_render has to do for the batch: 0 nothing (culled, or no point to draw), 1 draw, 2 draw with the checked variant
            const uint32_t kind = !(lod & LOD_CULLED) && (lod & LOD_NPR_MASK) ? ((rec_flags & bf_generic(a.parts)) ? 2u : 1u) : 0u;
            const uint64_t below = (1ull << threadIdx.x) - 1ull;
            // (the list entry is the whole record k_render's workgroup starts from, see RenderArgs::order)
            DrawRec r = {0, 0, 0, 0, 0, 0};
            if (kind) { r.b = bb; r.lod = lod; r.esc_total = rec_esc_total; r.esc_mid = rec_esc_mid; r.sep_off = rec_sep_off; }
            // the ordinary list: class after class inside the chunk
            const uint32_t npr = r.lod & LOD_NPR_MASK;
            const uint32_t cls = a.work_classes > 1 && npr ? (uint32_t)(WORK_CLASSES - 1) - min((npr - 1u) / (64u / WORK_CLASSES), (uint32_t)(WORK_CLASSES - 1)) : 0u;
            uint32_t base = 0;
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)WORK_CLASSES; ++k) {
                const uint64_t m = __ballot(kind == 1u && cls == k);
                if (threadIdx.x == 0) a.chunk_count[k * PCR_MAX_PREPASS_WORKGROUPS + block] = (uint32_t)__popcll(m);
                if (kind == 1u && cls == k) a.order[block * PREPASS_BATCHES + base + (uint32_t)__popcll(m & below)] = r;
                base += (uint32_t)__popcll(m);
            }
            {   // the checked list
                const uint64_t m = __ballot(kind == 2u);
                if (threadIdx.x == 0) a.chunk_count[WORK_CLASSES * PCR_MAX_PREPASS_WORKGROUPS + block] = (uint32_t)__popcll(m);
                if (kind == 2u) a.order[(size_t)a.order_stride + block * PREPASS_BATCHES + (uint32_t)__popcll(m & below)] = r;
            }
        }
        PCR_PTL(4);
        return;
    }
    // LDS framebuffer windows of the workgroups that draw
    // ... and a vote: do (nearly) all of the chunk's batches (32 neighbours in the file) lie mostly outside their windows? Only then does
    // k_render pre-read the framebuffer words of such a batch's points (WinPlan::mostly_outside, project_request): a few batches of
    // that kind in a frame are cheaper unfiltered, a frame full of them (an unsorted stream) is not.
    __shared__ uint32_t s_vote[2];                          // plans drawn, of those mostly outside
    if (threadIdx.x < 2) s_vote[threadIdx.x] = 0;
    __syncthreads();                                        // (s_lod is complete)
    PCR_PTL(5);
    bool mine_drawn = false, mine_outside = false;
    if (plan_lane) {
        const uint32_t lod = s_lod[slot];
        if (!(lod & LOD_CULLED) && (lod & LOD_NPR_MASK)) {
            const bool outside = plan_windows(a, pb, (int)part, (int)(threadIdx.x % RUNS), pin);     // (uniform per RUNS lanes)
            if (threadIdx.x % RUNS == 0) {
                mine_drawn = true; mine_outside = outside;
                atomicAdd(&s_vote[0], 1u);
                if (outside) atomicAdd(&s_vote[1], 1u);
            }
        }
    }
    PCR_PTL(6);
    __syncthreads();
    PCR_PTL(7);
    // (the plan was written with the batch's own verdict; the neighbourhood's vote takes it back)
    if (mine_drawn && mine_outside && s_vote[1] * PCR_VOTE_DEN < s_vote[0] * PCR_VOTE_NUM) {
        a.win[pb * parts + part].mostly_outside = 0;
        if (a.win_hqs) a.win_hqs[pb * parts + part].mostly_outside = 0;
    }
}

// Workgroup w of a launch's prepass part: chunk w / 2, the lists (even) or the plans (odd)
constexpr uint32_t PREPASS_WGS_PER_CHUNK = 2;
__device__ __forceinline__ void lod_prepass_block(const RenderArgs &a, uint32_t w)
{
    if (w & 1u) lod_prepass_chunk<false>(a, w >> 1);
    else        lod_prepass_chunk<true>(a, w >> 1);
}

// One lane per run: the screen rectangle of the run's bounding box (k_bounds), then LDS pixels for the RUNS rectangles. Only a
// placement hint -- points that land outside their window take the global path -- so nothing here needs exactness, only
// rectangles inside the image and a pixel count within the LDS.
struct IRect { int x0, y0, x1, y1; };
__device__ __forceinline__ int rect_area(IRect r) { return r.x1 >= r.x0 && r.y1 >= r.y0 ? (r.x1 - r.x0 + 1) * (r.y1 - r.y0 + 1) : 0; }
// the central part of a rectangle, scaled by sc < 1 per side (what is outside goes the global way)
__device__ __forceinline__ IRect rect_shrink(IRect r, float sc)
{
    const int w = r.x1 - r.x0 + 1, h = r.y1 - r.y0 + 1;
    if (w <= 0 || h <= 0) return r;
    const int nw = max(1, (int)floorf((float)w * sc)), nh = max(1, (int)floorf((float)h * sc));
    const int x0 = r.x0 + (w - nw) / 2, y0 = r.y0 + (h - nh) / 2;
    return { x0, y0, x0 + nw - 1, y0 + nh - 1 };
}
__device__ __forceinline__ void rect_pack(IRect r, uint32_t &xy, uint32_t &wh)
{
    if (r.x1 < r.x0 || r.y1 < r.y0 || r.x0 < 0 || r.y0 < 0 || r.x0 >= 65536 || r.y0 >= 65536) { xy = 0; wh = 0; return; }
    xy = (uint32_t)r.x0 | ((uint32_t)r.y0 << 16);
    wh = (uint32_t)(r.x1 - r.x0 + 1) | ((uint32_t)(r.y1 - r.y0 + 1) << 16);
}

// Windows for one pixel size, given the run's rectangle `mine` and the rectangle `single` that holds every chain of the workgroup
// (uniform work per RUNS lanes).
// Returns the plan's own verdict "most points will land outside" (the same in every lane of the group).
__device__ __forceinline__ bool assign_windows(int cap, IRect mine, IRect single, uint32_t cut, int r, WinPlan *out)
{
    auto group_sum = [](int v) { v += __shfl_xor(v, 1, RUNS); v += __shfl_xor(v, 2, RUNS); return v; };
    auto group_max = [](int v) { v = max(v, __shfl_xor(v, 1, RUNS)); v = max(v, __shfl_xor(v, 2, RUNS)); return v; };
    const IRect none = { 0x3FFFFFFF, 0x3FFFFFFF, -1, -1 };
    // The rectangle of all the workgroup's chains (the batch's own bounding box, GPUBatch, cut down to what k_bounds saw of the
    // part; the runs' boxes leave the straddling chains out): if the LDS holds it, it is the one window and no point lands outside.
    if (rect_area(single) > 0 && rect_area(single) <= cap) {
        uint32_t xy = 0, wh = 0;
        if (r == 0) rect_pack(single, xy, wh);
        out->xy[r] = xy; out->wh[r] = wh;                                           // (runs 1..3: no window of their own)
        if (r < RUNS - 1) out->first[r] = PCR_WORKGROUP_SIZE;                       // every chain belongs to run 0
        if (r == 0) out->mostly_outside = 0;
        return false;
    }
    // one window per run; while they do not fit together, the largest gives way (a run with a jump of its own inside)
    int sum = group_sum(rect_area(mine));
    const int wanted = sum;
#pragma unroll 1
    for (int round = 0; round < 6 && sum > cap; ++round) {
        const int area = rect_area(mine), largest = group_max(area);
        // (ties: every holder of the largest area shrinks -- the loop ends all the same)
        if (area == largest) {
            const int target = max(largest - (sum - cap), largest / 4);
            mine = target >= 1 ? rect_shrink(mine, sqrtf((float)target / (float)largest)) : none;
        }
        sum = group_sum(rect_area(mine));
    }
    if (sum > cap) mine = none;                                                     // hopeless: this run goes the global way
    rect_pack(mine, out->xy[r], out->wh[r]);
    if (r < RUNS - 1) out->first[r] = min(cut, (uint32_t)PCR_WORKGROUP_SIZE);
    const int kept = group_sum(rect_area(mine));                                    // (every lane of the group: a shuffle)
    const bool outside = kept * 2 < wanted;
    if (r == 0) out->mostly_outside = outside ? 1u : 0u;
    return outside;
}

// The screen rectangle of a box, worked out by a group of RUNS lanes (corners 2 r and 2 r + 1 per lane, then the union over the
// group); no rectangle for a box that reaches behind the camera, holds a NaN, or lies off screen.
__device__ __forceinline__ IRect group_box_rect(const pcr_render_params &p, const float *bmin, const float *bmax, int r)
{
    const IRect none = { 0x3FFFFFFF, 0x3FFFFFFF, -1, -1 };
    const float fw = (float)p.width, fh = (float)p.height;
    auto group_sum = [](int v) { v += __shfl_xor(v, 1, RUNS); v += __shfl_xor(v, 2, RUNS); return v; };
    bool front = true;
    float minx = 3.0e38f, maxx = -3.0e38f, miny = 3.0e38f, maxy = -3.0e38f;
#pragma unroll
    for (int c2 = 0; c2 < 8 / RUNS; ++c2) {
        const int c = r * (8 / RUNS) + c2;
        const float x = (c & 1) ? bmax[0] : bmin[0], y = (c & 2) ? bmax[1] : bmin[1], z = (c & 4) ? bmax[2] : bmin[2];
        const float w = dot4(p.transform + 12, x, y, z, 1.0f);
        front = front && w > 1.0e-6f;
        const float rw = __builtin_amdgcn_rcpf(w);              // (a placement hint: v_rcp_f32's 1 ulp is plenty, an IEEE division is ten instructions)
        const float sx = (dot4(p.transform + 0, x, y, z, 1.0f) * rw * 0.5f + 0.5f) * fw;
        const float sy = (dot4(p.transform + 4, x, y, z, 1.0f) * rw * 0.5f + 0.5f) * fh;
        minx = fminf(minx, sx); maxx = fmaxf(maxx, sx); miny = fminf(miny, sy); maxy = fmaxf(maxy, sy);
    }
#pragma unroll
    for (int m = 1; m < RUNS; m <<= 1) {
        minx = fminf(minx, __shfl_xor(minx, m, RUNS)); maxx = fmaxf(maxx, __shfl_xor(maxx, m, RUNS));
        miny = fminf(miny, __shfl_xor(miny, m, RUNS)); maxy = fmaxf(maxy, __shfl_xor(maxy, m, RUNS));
    }
    const bool all_front = (group_sum(front ? 1 : 0) == RUNS);
    IRect rc = none;
    if (all_front && maxx >= -1.0f && maxy >= -1.0f && minx <= fw + 1.0f && miny <= fh + 1.0f) {
        rc.x0 = max(0, (int)floorf(fmaxf(minx, -2.0f)) - 1); rc.x1 = min(p.width - 1, (int)floorf(fminf(maxx, fw + 2.0f)) + 1);
        rc.y0 = max(0, (int)floorf(fmaxf(miny, -2.0f)) - 1); rc.y1 = min(p.height - 1, (int)floorf(fminf(maxy, fh + 2.0f)) + 1);
        if (rc.x1 < rc.x0 || rc.y1 < rc.y0) rc = none;
    }
    return rc;
}
__device__ __forceinline__ IRect rect_intersect(IRect a, IRect b)
{
    const IRect none = { 0x3FFFFFFF, 0x3FFFFFFF, -1, -1 };
    const IRect c = { max(a.x0, b.x0), max(a.y0, b.y0), min(a.x1, b.x1), min(a.y1, b.y1) };
    return c.x1 >= c.x0 && c.y1 >= c.y0 ? c : none;
}

__device__ __forceinline__ bool plan_windows(const RenderArgs &a, int64_t b, int part, int r, const PlanIn &in)
{
    const pcr_render_params &p = a.p;
    const float fw = (float)p.width, fh = (float)p.height;
    const float bmin[3] = { in.box[0], in.box[1], in.box[2] }, bmax[3] = { in.box[3], in.box[4], in.box[5] };
    const IRect none = { 0x3FFFFFFF, 0x3FFFFFFF, -1, -1 };      // identity of the union
    // the batch's own bounding box (GPUBatch: it holds every point but the garbage tails of SURVEY B.4): the dirty tiles are marked under it
    const IRect whole = group_box_rect(p, in.gmin, in.gmax, r);
    // ... and what k_bounds saw of the workgroup's own chains, garbage tails included (a hint: float dequantisation): the one
    // window of the workgroup if the LDS holds it
    IRect single = group_box_rect(p, in.all, in.all + 3, r);
    if (rect_area(whole) > 0) single = rect_area(single) > 0 ? rect_intersect(single, whole) : whole;
    const int cap = window_capacity(in.esc_count, a.win_pixel_bytes, a.dyn_lds_bytes, a.parts);
    const int cap_hqs = a.win_hqs ? window_capacity(in.esc_count, WIN_PIXEL_BYTES_HQS, a.dyn_lds_bytes, a.parts) : cap;
    // The rectangle of MY run of chains -- eight corners per lane, two thirds of this function's arithmetic, and the prepass block sits
    // on every frame's critical path (it was 6 of the 12 us of the frame turn) -- is worked out only if some plan of the group needs
    // windows per run: 96 % of the half-batches of the benchmark frame get the one window.
    IRect mine = none;
    const bool single_fits = rect_area(single) > 0 && rect_area(single) <= min(cap, cap_hqs);
    if (!single_fits) {
        bool front = true;
        float minx = 3.0e38f, maxx = -3.0e38f, miny = 3.0e38f, maxy = -3.0e38f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float x = (c & 1) ? bmax[0] : bmin[0], y = (c & 2) ? bmax[1] : bmin[1], z = (c & 4) ? bmax[2] : bmin[2];
            const float w = dot4(p.transform + 12, x, y, z, 1.0f);
            front = front && w > 1.0e-6f;
            const float rw = __builtin_amdgcn_rcpf(w);
            const float sx = (dot4(p.transform + 0, x, y, z, 1.0f) * rw * 0.5f + 0.5f) * fw;
            const float sy = (dot4(p.transform + 4, x, y, z, 1.0f) * rw * 0.5f + 0.5f) * fh;
            minx = fminf(minx, sx); maxx = fmaxf(maxx, sx); miny = fminf(miny, sy); maxy = fmaxf(maxy, sy);
        }
        // (a box that reaches behind the camera, holds a NaN, or lies off screen gets no rectangle: its points take the global path)
        if (front && maxx >= -1.0f && maxy >= -1.0f && minx <= fw + 1.0f && miny <= fh + 1.0f) {
            mine.x0 = max(0, (int)floorf(fmaxf(minx, -2.0f)) - 1); mine.x1 = min(p.width - 1, (int)floorf(fminf(maxx, fw + 2.0f)) + 1);
            mine.y0 = max(0, (int)floorf(fmaxf(miny, -2.0f)) - 1); mine.y1 = min(p.height - 1, (int)floorf(fminf(maxy, fh + 2.0f)) + 1);
            if (mine.x1 < mine.x0 || mine.y1 < mine.y0) mine = none;
        }
        // No window reaches outside the rectangle the dirty tiles are marked under (FrameView::tiles): a point INSIDE its window is
        // never tested against that rectangle, so a window sticking out of it -- a run's box holds its chains' garbage tails -- would
        // let the merge write framebuffer words in a tile nobody marked. What lies outside goes the off-window way,
        // which marks. (`single` is cut to it above.)
        if (rect_area(whole) > 0) mine = rect_intersect(mine, whole);
    }
    const bool outside = assign_windows(cap, mine, single, in.cut, r, a.win + b * a.parts + part);
    if (a.win_hqs)                                              // (uniform) the colour pass of the same frame: 20-byte pixels
        assign_windows(cap_hqs, mine, single, in.cut, r, a.win_hqs + b * a.parts + part);
    // dirty tiles (FrameView::tiles): everything under the batch's rectangle; a batch without one can write anywhere
    uint32_t wxy = 0, wwh = 0;
    if (a.f.tiles) {
        if (rect_area(whole) > 0) {
            rect_pack(whole, wxy, wwh);
            const uint32_t tx0 = (uint32_t)whole.x0 >> TILE_W_SHIFT, ty0 = (uint32_t)whole.y0 >> TILE_H_SHIFT;
            const uint32_t ntx = ((uint32_t)whole.x1 >> TILE_W_SHIFT) - tx0 + 1u, nty = ((uint32_t)whole.y1 >> TILE_H_SHIFT) - ty0 + 1u;
            if (ntx * nty * 4u > a.f.tiles_total) {
                // a batch that close to the camera covers a quarter of the screen or more: "everything", and the turn walks the frame linearly
                if (r == 0 && part == 0) *a.f.tiles_all = a.f.tiles_epoch;
            } else if (part == 0) {                                                     // (the batch's first group of lanes marks)
                for (uint32_t ty = (uint32_t)r; ty < nty; ty += RUNS) {                  // a row of tiles per lane and turn
                    uint8_t *row = a.f.tiles + (ty0 + ty) * a.f.tiles_x + tx0, *end = row + ntx;
                    while (row < end && ((uintptr_t)row & 3u)) *row++ = 1;
                    for (; row + 4 <= end; row += 4) *(uint32_t *)row = 0x01010101u;
                    while (row < end) *row++ = 1;
                }
            }
        } else if (r == 0 && part == 0) {
            *a.f.tiles_all = a.f.tiles_epoch;
        }
    }
    if (r == 0) {
        WinPlan *w = a.win + b * a.parts + part;
        w->whole_xy = wxy; w->whole_wh = wwh;
        if (a.win_hqs) { w = a.win_hqs + b * a.parts + part; w->whole_xy = wxy; w->whole_wh = wwh; }
    }
    return outside;
}

__global__ void __launch_bounds__(PREPASS_THREADS) k_lod_prepass(RenderArgs a) { lod_prepass_block(a, blockIdx.x); }

// publish: write lod[b] and count the batch in the statistics (the lists' workgroup; the plans' workgroup only wants the decision)
__device__ __forceinline__ uint32_t lod_prepass_batch(const RenderArgs &a, int64_t b, int lane, pcr_render_stats &st, bool publish)
{
    const uint32_t group_shift = (threadIdx.x & 63u) & ~(uint32_t)(PREPASS_LANES - 1);   // my group's bits in a wave ballot
    const pcr_gpu_batch g = a.s.batches[b];
    const pcr_render_params &p = a.p;
    const float lm[3] = { (float)g.las_min_x, (float)g.las_min_y, (float)g.las_min_z };      // :336
    const float bmin[3] = { g.min_x - lm[0], g.min_y - lm[1], g.min_z - lm[2] };             // :340
    const float bmax[3] = { g.max_x - lm[0], g.max_y - lm[1], g.max_z - lm[2] };             // :341

    if (lane == 0 && publish) st.batches_total = 1;
    if (p.enable_frustum_culling) {                                                          // :342-344
        // planes (3-0), (3+0), (3+1), (3-1), (3-2), (3+2) of the transposed matrix (three.js convention, :246-259);
        // x + s*y with s = +-1 is the same rounding as x +- y
        const float *M = p.transform;
        const int r = (lane >> 1) & 3;
        const float sgn = (lane == 0 || lane == 3 || lane == 4) ? -1.0f : 1.0f;
        bool accept = true;
        if (lane < 6)
            accept = plane_accepts(M[12] + sgn * M[4 * r + 0], M[13] + sgn * M[4 * r + 1], M[14] + sgn * M[4 * r + 2],
                                   M[15] + sgn * M[4 * r + 3], bmin, bmax);
        const uint32_t votes = (uint32_t)(__ballot(accept) >> group_shift) & 0xFFu;
        if (votes != 0xFFu) {
            if (lane == 0 && publish) { a.lod[b] = LOD_CULLED; st.batches_culled = 1; }      // (plain stores to distinct fields: with `+=` hipcc
                                                                                      // merged them into one store at a computed offset -- scratch)
            return LOD_CULLED;
        }
    }
    // :349-375
    const float cx = 0.5f * (bmin[0] + bmax[0]), cy = 0.5f * (bmin[1] + bmax[1]), cz = 0.5f * (bmin[2] + bmax[2]);
    const float dx = bmin[0] - bmax[0], dy = bmin[1] - bmax[1], dz = bmin[2] - bmax[2];
    const float rad = sqrtf(__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx)));
    float vc[4], ve[4], pc[4], pe[4];
    for (int r = 0; r < 4; ++r) vc[r] = dot4(p.world_view + 4 * r, cx, cy, cz, 1.0f);
    ve[0] = vc[0] + rad; ve[1] = vc[1] + 0.0f; ve[2] = vc[2] + 0.0f; ve[3] = vc[3] + 0.0f;
    for (int r = 0; r < 4; ++r) {
        pc[r] = dot4(p.proj + 4 * r, vc[0], vc[1], vc[2], vc[3]);
        pe[r] = dot4(p.proj + 4 * r, ve[0], ve[1], ve[2], ve[3]);
    }
    const float fw = (float)p.width, fh = (float)p.height;
    const float scx = fw * (0.5f * (pc[0] / pc[3] + 1.0f)), scy = fh * (0.5f * (pc[1] / pc[3] + 1.0f));
    const float sex = fw * (0.5f * (pe[0] / pe[3] + 1.0f)), sey = fh * (0.5f * (pe[1] / pe[3] + 1.0f));
    const float ddx = sex - scx, ddy = sey - scy;
    float px = sqrtf(__fmaf_rn(ddy, ddy, ddx * ddx));
    const bool use_double = px >= 100.0f;                                    // :370
    if (a.variant_hqs) px = (float)((double)px / 100.0);                     // hqs depth.cu:223 / render.cu:388
    else               px = px / 100.0f;                                     // mem_iter render.cu:372
    float pct = (float)((double)(1.8f * px) - 0.3);                          // :373
    pct = fmaxf((float)p.lod_percent / 100.0f, fminf(pct, 1.0f));            // :374
    int npr = (int)(pct * (float)p.points_per_thread);                       // :375
    npr = min(npr, p.points_per_thread);
    npr = max(npr, 0);
    const uint32_t lod = (uint32_t)npr | (use_double ? LOD_DOUBLE : 0u);
    if (lane == 0 && publish) {
        a.lod[b] = lod;
        st.points_iterated = (int64_t)npr * PCR_WORKGROUP_SIZE;
        st.batches_double = use_double ? 1 : 0;
    }
    return lod;
}

// ------------------------------------------------------------------------------------------------
// decode + rasterize: one workgroup per batch, one chain per lane (lanes are independent after k_transcode)
//
// Memory plan per workgroup (PARTS = 1: LDS 76 KiB -> two workgroups per CU; PARTS = 2: 40 KiB -> four; 8 waves per SIMD either way):
//   s_table  16 KiB  decoder table packed to one dword per key (TE_* above): value << 10 | wide << 9 | escape << 8 | len
//                    (an in-table value outside +-2^21 is flagged `wide` and re-read from global memory)
//   s_dyn    60 KiB  (or 140 KiB) shared per batch (see DYN_LDS_BYTES above) between
//   s_esc            the escape ("separate") words of the batch, bulk-loaded coalesced up front (batches with
//                    more than ESC_POOL_WORDS escapes read the ones past the pool from global memory), and
//   s_win            the framebuffer words of the batch's screen rectangle -- or of one rectangle per run of chains
//                    (WinPlan, k_lod_prepass): they start empty, the depth pre-read and the atomicMin of every point that
//                    lands inside run on LDS (ds_read_b64 / ds_min_u64); at the end the windows are merged into the global
//                    framebuffer with one row-coalesced atomicMin per pixel a point reached and improved. min is
//                    associative, so the result is the same u64 per pixel; what changes is the number of global atomics:
//                    one per touched pixel and batch instead of one per new per-pixel minimum.
//   registers        the point's 40-bit window + the next one + the one in flight (LAYOUT_POINT_WINDOWS), or three words of
//                    the chain's own sequence + two requested a point ahead (LAYOUT_WORDS; see the word window below)
// Global loads left in the loop are consumed at least one iteration after they are issued.
// ------------------------------------------------------------------------------------------------
// BC1 block -> its four palette colours (render.cu:31-62, always 4-colour mode), one register per channel: byte k of
// r / g / b is that channel of palette entry k. A point's colour 0x00BBGGRR is then two v_perm_b32 with selectors built from
// its 2-bit index -- five instructions per point, no compare/select chain.
struct Bc1Palette { uint32_t r, g, b, selectors; };

__device__ __forceinline__ Bc1Palette bc1_palette(uint2 blk)
{
    const uint32_t l = blk.x & 0xFFFFu, h = blk.x >> 16;
    const uint32_t cr0 = (l >> 11) & 31, cg0 = (l >> 5) & 63, cb0 = l & 31;
    const uint32_t r0 = (cr0 << 3) | (cr0 >> 2), g0 = (cg0 << 2) | (cg0 >> 4), b0 = (cb0 << 3) | (cb0 >> 2);
    const uint32_t cr1 = (h >> 11) & 31, cg1 = (h >> 5) & 63, cb1 = h & 31;
    const uint32_t r1 = (cr1 << 3) | (cr1 >> 2), g1 = (cg1 << 2) | (cg1 >> 4), b1 = (cb1 << 3) | (cb1 >> 2);
    auto four = [](uint32_t c0, uint32_t c1) {              // entries 2 and 3: (2 c0 + c1) / 3 and (c0 + 2 c1) / 3, integer division
        return c0 | (c1 << 8) | (((c0 * 2 + c1) / 3) << 16) | (((c0 + c1 * 2) / 3) << 24);
    };
    Bc1Palette p;
    p.r = four(r0, r1); p.g = four(g0, g1); p.b = four(b0, b1);
    p.selectors = blk.y;                                    // byte 4 + local/4, bits 2*(local%4): render.cu:48
    return p;
}

__device__ __forceinline__ uint32_t bc1_color(const Bc1Palette &p, uint32_t local)
{
    const uint32_t sel = (p.selectors >> (2 * local)) & 3u;
    // v_perm_b32(s0, s1, m): result byte i = byte m.byte[i] of {s0 (4..7), s1 (0..3)}, 0x0c = zero
    const uint32_t rg = __builtin_amdgcn_perm(p.g, p.r, sel * 0x0101u + 0x0c0c0400u);          // g[sel] << 8 | r[sel]
    return __builtin_amdgcn_perm(p.b, rg, (sel << 16) + 0x0c040100u);                          // b[sel] << 16 | rg
}

// The same for the HQS colour pass, which wants r << 16 | g and b << 16 | 1 (the contribution of one point to the two packed sums): one
// selector, two v_perm_b32 -- the colour itself is never put together.
__device__ __forceinline__ void bc1_contribution(const Bc1Palette &p, uint32_t local, uint32_t &rg16, uint32_t &bc16)
{
    const uint32_t sel = (p.selectors >> (2 * local)) & 3u;
    const uint32_t m = sel * 0x00010001u + 0x0c040c00u;     // byte 2 <- s0[sel], byte 0 <- s1[sel], zero elsewhere
    rg16 = __builtin_amdgcn_perm(p.r, p.g, m);              // r[sel] << 16 | g[sel]
    bc16 = __builtin_amdgcn_perm(p.b, 0x01010101u, m);      // b[sel] << 16 | 1
}

// BC7 mode-6 block as the reference's kernels decode it (huffman_hqs/render.cu:240-273, struct bc7_mode_6 render.cu:66-110):
// endpoints = 7 bits << 1 | p-bit, the 4-bit field at 4 * local of the high quadword as the index of EVERY pixel (so pixel 0
// gets index << 1 | p1: reproduced), weight = round(idx * 64 / 15) = (idx * 64 + 7) / 15, channel = (e0 (64 - w) + e1 w + 32) >> 6.
// Kept per 16-point block as: r | b << 16 of endpoint 0, of endpoint 1, g0 | g1 << 16, and the two index words.
struct Bc7Block { uint32_t rb0, rb1, g01, idx_lo, idx_hi; };
__device__ __forceinline__ Bc7Block bc7_block(uint4 blk)
{
    const uint64_t lo = ((uint64_t)blk.y << 32) | blk.x;
    const uint32_t p0 = blk.y >> 31, p1 = blk.z & 1u;
    auto end = [&](int shift, uint32_t pbit) { return ((((uint32_t)(lo >> shift)) & 127u) << 1) | pbit; };
    Bc7Block k;
    k.rb0 = end(7, p0) | (end(35, p0) << 16);
    k.rb1 = end(14, p1) | (end(42, p1) << 16);
    k.g01 = end(21, p0) | (end(28, p1) << 16);
    k.idx_lo = blk.z; k.idx_hi = blk.w;
    return k;
}
__device__ __forceinline__ uint32_t bc7_color(const Bc7Block &k, uint32_t local)        // 0x00BBGGRR (the pass has no use for alpha)
{
    const uint32_t idx = ((local < 8u ? k.idx_lo : k.idx_hi) >> (4u * (local & 7u))) & 15u;
    const uint32_t w = ((idx * 64u + 7u) * 4370u) >> 16, iw = 64u - w;                  // (n / 15 for the sixteen n = 64 idx + 7)
    const uint32_t rb = ((k.rb0 * iw + k.rb1 * w + 0x00200020u) >> 6) & 0x00FF00FFu;
    const uint32_t g = (((k.g01 & 0xFFFFu) * iw + (k.g01 >> 16) * w + 32u) >> 6) & 0xFFu;
    return rb | (g << 8);
}

// Row/column of linear index i in a window of width ww (i < 2^23): one float multiply and a correction of at most one
// row instead of a 32-bit integer division (which costs ~25 VALU on this hardware and sat in every set-up and merge).
__device__ __forceinline__ void window_row_col(uint32_t i, uint32_t ww, float inv_ww, uint32_t &y, uint32_t &x)
{
    y = (uint32_t)((float)i * inv_ww);
    int32_t r = (int32_t)(i - y * ww);
    if (r < 0) { --y; r += (int32_t)ww; }
    else if (r >= (int32_t)ww) { ++y; r -= (int32_t)ww; }
    x = (uint32_t)r;
}

// One packed dword per table key (layout above): value and length of render.cu:435-439 in a single LDS read.
__device__ __forceinline__ bool table_value_fits(int32_t value) { return value > TE_SLOW_VALUE && value < -TE_SLOW_VALUE; }
__device__ __forceinline__ uint32_t pack_table_entry(int32_t value, uint32_t lbyte)
{
    const int len = (int)(int8_t)lbyte;                             // render.cu:393 narrows to char
    const uint32_t slow = (uint32_t)TE_SLOW_VALUE << TE_VALUE_SHIFT;
    if (len <= 0) return slow | TE_ESCAPE | (uint32_t)abs(len);     // escapes never use the table value
    return table_value_fits(value) ? (((uint32_t)value << TE_VALUE_SHIFT) | (uint32_t)len) : (slow | TE_WIDE | (uint32_t)len);
}

// ------------------------------------------------------------------------------------------------
// k_transcode: the reference's lockstep decode (render.cu:404-451), run ONCE per loaded batch instead of every frame.
//
// The stream format interleaves the words of 32 chains in the order a 32-lane warp requests them, so finding "my next
// word" costs a ballot, a prefix count and a shared read per symbol — and the reference's tail quirk makes
// the word a lane receives depend on what the other 31 lanes did. This kernel performs exactly that walk (lengths only:
// no values, no escapes, no points) and writes down, per chain, the sequence of words it received: row r of
// lane_words holds the r-th word of each of the 1024 chains. k_render then decodes every chain from its own word
// sequence — same bits, same symbols, garbage tails included — without any cross-lane step. A decode truncated by the
// level of detail consumes a prefix of the same sequence (the walk is causal), so one transcode serves every frame.
// The walk also counts the escapes every chain reads, which tells whether k_render may take them from its LDS pool
// without checking (batch_flags).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(PCR_WORKGROUP_SIZE) k_transcode(StreamView s, uint32_t *lane_words, uint32_t *batch_flags,
                                                                  uint32_t *packed_table, uint8_t *point_windows, uint8_t *colors_t,
                                                                  uint32_t *any_generic, int first_batch, int lane_words_first, uint32_t *wave_rows)
{
    const uint32_t b = (uint32_t)first_batch + blockIdx.x;
    const uint32_t tid = threadIdx.x;
    // my chain's four colour blocks -> segment-major order (StreamView::colors_t)
    if (s.color_block_bytes == 16) {
        const uint4 *src = reinterpret_cast<const uint4 *>(s.colors) + ((size_t)b * 4096 + tid * 4);
        uint4 *dst = reinterpret_cast<uint4 *>(colors_t) + (size_t)b * 4096 + tid;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k * PCR_WORKGROUP_SIZE] = src[k];
    } else {
        const uint2 *src = reinterpret_cast<const uint2 *>(s.colors) + ((size_t)b * 4096 + tid * 4);
        uint2 *dst = reinterpret_cast<uint2 *>(colors_t) + (size_t)b * 4096 + tid;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k * PCR_WORKGROUP_SIZE] = src[k];
    }
    __shared__ __align__(16) uint8_t s_len[PCR_HUFFMAN_TABLE_SIZE];
    __shared__ __align__(16) uint32_t s_ring[PCR_CLUSTERS_PER_BATCH * RING_WORDS];
    bool generic = false;
    {
        const uint32_t l4 = reinterpret_cast<const uint32_t *>(s.table_lens + (size_t)b * PCR_HUFFMAN_TABLE_SIZE)[tid];
        const int4 v = reinterpret_cast<const int4 *>(s.table_values + (size_t)b * PCR_HUFFMAN_TABLE_SIZE)[tid];
        // byte: |len| (render.cu:393, :439), bit 7: escape (len <= 0, as k_render packs it)
        auto entry = [&](uint32_t lbyte, int32_t value) -> uint32_t {
            const int len = (int)(int8_t)lbyte;
            if (len > 0 && !table_value_fits(value)) generic = true;                       // "wide" entry of k_render's table
            return (uint32_t)abs(len) | (len <= 0 ? 0x80u : 0u);
        };
        reinterpret_cast<uint32_t *>(s_len)[tid] = entry(l4 & 0xFF, v.x) | (entry((l4 >> 8) & 0xFF, v.y) << 8) |
                                                   (entry((l4 >> 16) & 0xFF, v.z) << 16) | (entry(l4 >> 24, v.w) << 24);
        // k_render's table, packed once here instead of by every frame's prologue
        uint4 e;
        e.x = pack_table_entry(v.x, l4 & 0xFF); e.y = pack_table_entry(v.y, (l4 >> 8) & 0xFF);
        e.z = pack_table_entry(v.z, (l4 >> 16) & 0xFF); e.w = pack_table_entry(v.w, l4 >> 24);
        reinterpret_cast<uint4 *>(packed_table + (size_t)b * PCR_HUFFMAN_TABLE_SIZE)[tid] = e;
    }
    const pcr_gpu_batch *gb = s.batches + b;
    const int64_t enc_off = gb->encoding_batch_offset;      // :404
    const uint32_t *enc = s.encoded + enc_off;
    const uint32_t enc_last = (uint32_t)min((int64_t)0x7FFFFFF0, s.encoded_words + (PCR_GUARD_WORDS - 2) - enc_off);
    auto enc_load = [&](uint32_t i) -> uint32_t { return enc[min(i, enc_last)]; };
    auto enc_load2 = [&](uint32_t i) -> uint2 { uint2 v; __builtin_memcpy(&v, enc + min(i, enc_last), 8); return v; };
    const uint32_t cluster = tid >> 5, lane32 = tid & 31u, half_shift = tid & 32u, lanes_below = (1u << lane32) - 1u;
    const uint32_t cbase = cluster ? (uint32_t)s.cluster_sizes[(size_t)b * 32 + cluster - 1] : 0u;   // :407-410
    uint32_t *ring = s_ring + cluster * RING_WORDS;
    uint32_t *out = lane_words + (size_t)(b - (uint32_t)lane_words_first) * LW_ROWS * PCR_WORKGROUP_SIZE + tid;   // resident, or the launch's scratch
    uint32_t row = 2;
    uint64_t bits = ((uint64_t)enc_load(cbase + lane32) << 32) | enc_load(cbase + 32 + lane32);   // :416-417
    out[0] = (uint32_t)(bits >> 32);
    out[PCR_WORKGROUP_SIZE] = (uint32_t)bits;
    reinterpret_cast<uint2 *>(ring + CHUNK_WORDS)[lane32] = enc_load2(cbase + CHUNK_WORDS + lane32 * 2);
    reinterpret_cast<uint2 *>(ring)[lane32] = enc_load2(cbase + 2 * CHUNK_WORDS + lane32 * 2);
    uint2 stage = enc_load2(cbase + 3 * CHUNK_WORDS + lane32 * 2);
    uint32_t ep = 64, next_cross = 2 * CHUNK_WORDS;         // already_read (:418)
    uint32_t sft = 32 + 20;                                 // cur_bits (:419) + 20: (bits >> sft) & 0xFFF is the key of :431-433
    uint32_t nesc = 0;
    // my column of the batch's two window planes (high: u32 per point, low: u8 per point)
    uint32_t *pw = point_windows ? reinterpret_cast<uint32_t *>(point_windows + (size_t)b * PW_BATCH_BYTES) + tid : nullptr;
    uint8_t *pwl = point_windows ? point_windows + (size_t)b * PW_BATCH_BYTES + PW_HI_BYTES + tid : nullptr;
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < PCR_POINTS_PER_THREAD * 3; ++k) {   // :428-430
        // first bit of point k/3 in the chain's own word sequence: `bits` holds rows row-2 and row-1, of which 52 - sft
        // bits are consumed (parked in the window slot until the words behind it are known, see below)
        if (pw && k % 3 == 0) pw[(size_t)(k / 3) * PCR_WORKGROUP_SIZE] = 32u * (row - 2u) + (52u - sft);
        const uint32_t l = s_len[(uint32_t)(bits >> sft) & 0xFFFu];
        sft -= l & 0x7Fu;                                   // :435-439
        nesc += l >> 7;                                     // :438
        const bool need = sft <= 20u;                       // :442 (cur_bits <= 0)
        const uint64_t m = __ballot(need);                  // :443
        const uint32_t mh = (uint32_t)(m >> half_shift);
        if (need) {                                         // :444-449
            const uint32_t w = ring[(ep + __popc(mh & lanes_below)) & (RING_WORDS - 1)];
            out[(size_t)row * PCR_WORKGROUP_SIZE] = w;
            ++row;
            bits = (bits << 32) | w;
            sft += 32;
        }
        ep += __popc(mh);                                   // :450
        if (ep >= next_cross) {
            // My half of the wave has consumed the ring's older chunk (a step takes at most 32 words, so every refill
            // stays inside the two resident chunks): overwrite it with the staged chunk and fetch the chunk after that.
            // DS operations of a wave execute in order; the fences below only stop the compiler from reordering.
            reinterpret_cast<uint2 *>(ring + ((next_cross + CHUNK_WORDS) & (RING_WORDS - 1)))[lane32] = stage;
            stage = enc_load2(cbase + next_cross + 2 * CHUNK_WORDS + lane32 * 2);
            next_cross += CHUNK_WORDS;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // rows of the lane-major copy my wave's longest chain wrote (PCR_LAYOUT_WORDS keeps exactly those, k_pack_words)
    if (wave_rows) {
        uint32_t rows = row;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) rows = max(rows, (uint32_t)__shfl_xor((int)rows, m));
        if ((tid & 63u) == 0) wave_rows[(size_t)(b - (uint32_t)lane_words_first) * LWC_WAVES + (tid >> 6)] = rows;
    }
    // may k_render read this batch's escapes from its LDS pool unchecked? (same pool rules as there: the whole batch's escapes in
    // the pool of a 1024-thread workgroup, a half's in the pool of a half-batch workgroup)
    const int32_t *ssz = s.separate_sizes + (size_t)b * 1024;
    const uint32_t esc_total = (uint32_t)ssz[1023], esc_mid = (uint32_t)ssz[511];
    const uint32_t sp0 = tid ? (uint32_t)ssz[tid - 1] : 0u;
    const bool over1 = nesc && sp0 + nesc > esc_pool_words(esc_total, 1);
    const uint32_t e0 = tid < 512u ? 0u : esc_mid, e1 = tid < 512u ? esc_mid : esc_total;
    const bool over2 = nesc && (sp0 - e0) + nesc > esc_pool_words(e1 - e0, 2);
    const int any1 = __syncthreads_or(generic || over1 ? 1 : 0), any2 = __syncthreads_or(generic || over2 ? 1 : 0);
    if (tid == 0) {
        const uint32_t flags = (any1 ? BF_GENERIC_SLOW_PATH : 0u) | (any2 ? BF_GENERIC_SLOW_PATH_HALF : 0u);
        batch_flags[b] = flags;
        if (flags) atomicOr(any_generic, flags);            // sticky, per stream: the host launches the checked kernel only if set
    }

    // Point windows: 40 bits of my word sequence from each point's first bit. A point's window can reach two words past
    // the last word the walk had fetched when the point began, so they are cut once the whole sequence is written. The two
    // rows behind the chain's last word are zeroed first: no symbol can reach those bits, but the buffer may be the
    // transcode scratch of an earlier launch and the windows should not depend on what it held.
    if (pw) {
        out[(size_t)row * PCR_WORKGROUP_SIZE] = 0;
        out[(size_t)(row + 1) * PCR_WORKGROUP_SIZE] = 0;
        __threadfence();            // my own stores above (lane_words column, parked positions) before I read them back
#pragma unroll 4
        for (int i = 0; i < PW_ROWS; ++i) {
            const uint32_t pos = pw[(size_t)i * PCR_WORKGROUP_SIZE];
            const uint32_t r = pos >> 5, o = pos & 31u;
            const uint32_t a0 = out[(size_t)r * PCR_WORKGROUP_SIZE], a1 = out[(size_t)(r + 1) * PCR_WORKGROUP_SIZE],
                           a2 = out[(size_t)(r + 2) * PCR_WORKGROUP_SIZE];
            // alignbit(hi, lo, s) = low 32 bits of (hi:lo) >> (s & 31): s = 32 - o cuts 32 bits starting o bits into hi
            const uint32_t hi = o ? __builtin_amdgcn_alignbit(a0, a1, 32u - o) : a0;
            const uint32_t lo = o ? __builtin_amdgcn_alignbit(a1, a2, 32u - o) : a1;
            pw[(size_t)i * PCR_WORKGROUP_SIZE] = hi;                     // bits 0..31 of the window
            pwl[(size_t)i * PCR_WORKGROUP_SIZE] = (uint8_t)(lo >> 24);   // bits 32..39
        }
    }
}

// k_pack_words: the compact copy of the lane-major words (PCR_LAYOUT_WORDS), once per loaded batch behind k_transcode: wave w of
// batch b copies rows [0, rows_w) of its 64 columns from the scratch form into the batch's block, 256 bytes per row.
__global__ void __launch_bounds__(PCR_WORKGROUP_SIZE) k_pack_words(const uint32_t *lane_words, uint32_t *const *lw_block,
                                                                   const uint32_t *lw_wave_row, int first_batch, int lane_words_first)
{
    const uint32_t b = (uint32_t)first_batch + blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t *src = lane_words + (size_t)(b - (uint32_t)lane_words_first) * LW_ROWS * PCR_WORKGROUP_SIZE + tid;
    const uint32_t r0 = lw_wave_row[(size_t)b * (LWC_WAVES + 1) + wave], r1 = lw_wave_row[(size_t)b * (LWC_WAVES + 1) + wave + 1];
    uint32_t *dst = lw_block[b] + (size_t)r0 * 64 + lane;
    for (uint32_t r = 0; r < r1 - r0; ++r) dst[(size_t)r * 64] = src[(size_t)r * PCR_WORKGROUP_SIZE];
}

// The provisional last batch of a stream that is still loading (enqueue_transcode): its block is a fixed scratch block, every wave's
// rows uncompacted (LW_ROWS each) -- nothing to read back, nothing to allocate.
__global__ void k_provisional_block(const uint32_t **lw_block, uint32_t *lw_wave_row, const uint32_t *scratch, int b)
{
    if (threadIdx.x == 0) lw_block[b] = scratch;
    if (threadIdx.x <= (uint32_t)LWC_WAVES) lw_wave_row[(size_t)b * (LWC_WAVES + 1) + threadIdx.x] = threadIdx.x * (uint32_t)LW_ROWS;
}

// ------------------------------------------------------------------------------------------------
// k_bounds: once per loaded batch, behind k_transcode (same launch geometry, same chunk of batches): where the batch's chains
// fall apart into spatial clusters (BatchRuns above). Every lane decodes its chain (the plain checked form of the decode, from
// the chain's own word sequence) and takes the bounding box of its 64 points; the three largest jumps between the box centres
// of consecutive chains cut the batch into four runs; each run gets the box of its chains. A hint and nothing more: it is
// computed with the float form of the dequantisation only (the double form of :459-461 differs by an ulp) and includes
// whatever the chains' garbage tails decode to -- and a frame is the same whatever the windows are.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t float_order(float f) { const uint32_t u = __float_as_uint(f); return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__device__ __forceinline__ float float_unorder(uint32_t u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

__global__ void __launch_bounds__(PCR_WORKGROUP_SIZE) k_bounds(StreamView s, const uint32_t *lane_words, uint32_t *batch_runs,
                                                               int first_batch, int lane_words_first)
{
    const uint32_t b = (uint32_t)first_batch + blockIdx.x;
    const uint32_t tid = threadIdx.x;
    __shared__ __align__(16) uint32_t s_table[PCR_HUFFMAN_TABLE_SIZE];
    __shared__ float s_centre[PCR_WORKGROUP_SIZE][3];
    __shared__ uint8_t s_outlier[PCR_WORKGROUP_SIZE];
    __shared__ float s_mean[PCR_WORKGROUP_SIZE / 64];
    __shared__ unsigned long long s_best[PCR_WORKGROUP_SIZE / 64];
    __shared__ uint32_t s_cut[MAX_PARTS][RUNS - 1];
    __shared__ uint32_t s_box[MAX_PARTS][RUNS][6];
    __shared__ uint32_t s_all[MAX_PARTS][6];
    reinterpret_cast<uint4 *>(s_table)[tid] = reinterpret_cast<const uint4 *>(s.packed_table + (size_t)b * PCR_HUFFMAN_TABLE_SIZE)[tid];
    const pcr_gpu_batch *gb = s.batches + b;
    const int32_t *sep = s.separate + gb->separate_batch_offset;
    const uint32_t sep_last = (uint32_t)min((int64_t)0x7FFFFFF0, s.separate_words + (PCR_GUARD_WORDS - 2) - gb->separate_batch_offset);
    const int32_t *tvalues = s.table_values + (size_t)b * PCR_HUFFMAN_TABLE_SIZE;
    uint32_t esc = tid ? (uint32_t)s.separate_sizes[(size_t)b * 1024 + tid - 1] : 0u;
    const uint32_t *col = lane_words + (size_t)(b - (uint32_t)lane_words_first) * LW_ROWS * PCR_WORKGROUP_SIZE + tid;
    const int32_t *sv = s.start_values + ((size_t)b * 1024 + tid) * 3;
    int32_t pos[3] = { sv[0], sv[1], sv[2] };
    const float fs[3] = { (float)gb->scale_x, (float)gb->scale_y, (float)gb->scale_z };
    const float fo[3] = { (float)(gb->offset_x - gb->las_min_x), (float)(gb->offset_y - gb->las_min_y), (float)(gb->offset_z - gb->las_min_z) };
    float mn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, mx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    __syncthreads();
    // the chain's bit stream = its words in order; `buf` holds the next `avail` (> 32) unconsumed bits at its top
    uint64_t buf = ((uint64_t)col[0] << 32) | col[PCR_WORKGROUP_SIZE];
    uint32_t row = 2;
    int avail = 64;
#pragma unroll 1
    for (int i = 0; i < PCR_POINTS_PER_THREAD; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t key = (uint32_t)(buf >> 52);
            const uint32_t e = s_table[key];
            int32_t val = (int32_t)e >> TE_VALUE_SHIFT;
            if (val == TE_SLOW_VALUE) {
                if (e & TE_ESCAPE) val = sep[min(esc++, sep_last)];
                else               val = tvalues[key];
            }
            pos[k] = (int32_t)((uint32_t)pos[k] + (uint32_t)val);
            const int len = (int)(e & 63u);                 // (a malformed table's longer "lengths" only make the hint useless)
            buf <<= len; avail -= len;
            if (avail <= 32) {
                const uint32_t w = row < (uint32_t)LW_ROWS ? col[(size_t)row * PCR_WORKGROUP_SIZE] : 0u;
                ++row;
                buf |= (uint64_t)w << ((32 - avail) & 63);
                avail += 32;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float f = __fmaf_rn((float)pos[k], fs[k], fo[k]);
            mn[k] = fminf(mn[k], f); mx[k] = fmaxf(mx[k], f);
        }
    }
    // A chain whose own 64 points straddle a jump of the curve (or whose garbage tail flew off) has a box far larger than its
    // neighbours': it would give a jump on either side and then blow up the box of the run it ends up in. Such chains -- extent
    // over four times the batch's mean -- are looked through when jumps are measured and left out of the runs' boxes (their few
    // points go the global way).
    float extent = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) extent = fmaxf(extent, mx[k] - mn[k]);
    if (!(extent >= 0.0f)) extent = 3.0e38f;                // (NaN)
    {
        float sum = fminf(extent, 1.0e30f);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) sum += __shfl_xor(sum, m);
        if ((tid & 63u) == 0) s_mean[tid >> 6] = sum;
    }
    __syncthreads();
    float mean = 0.0f;
#pragma unroll
    for (int w = 0; w < PCR_WORKGROUP_SIZE / 64; ++w) mean += s_mean[w];
    mean *= 1.0f / (float)PCR_WORKGROUP_SIZE;
    const bool outlier = extent > 4.0f * mean;
    // jump between my chain's box centre and the one of the nearest ordinary chain before it (at most four back)
#pragma unroll
    for (int k = 0; k < 3; ++k) s_centre[tid][k] = 0.5f * mn[k] + 0.5f * mx[k];
    s_outlier[tid] = outlier ? 1 : 0;
    __syncthreads();
    float gap0 = -1.0f;
    uint32_t prev = 0;
    if (tid && !outlier) {
        prev = tid - 1;
#pragma unroll
        for (int back = 0; back < 3; ++back) if (prev && s_outlier[prev]) --prev;
        gap0 = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) gap0 = fmaxf(gap0, fabsf(s_centre[tid][k] - s_centre[prev][k]));
        if (!(gap0 >= 0.0f) || s_outlier[prev]) gap0 = 0.0f;
    }
    // One record per workgroup shape and part (RUN_RECORDS): the whole batch, then its two halves side by side (every chain
    // belongs to one of them), each cut at the RUNS - 1 largest jumps INSIDE it.
#pragma unroll 1
    for (int cfg = 0; cfg < 2; ++cfg) {
        const uint32_t part_shift = 10u - (uint32_t)cfg, part = tid >> part_shift, part_first = part << part_shift;
        const uint32_t part_waves = 16u >> cfg, wave0 = part * part_waves;
        if (tid < 2 * RUNS * 6) s_box[tid / (RUNS * 6)][(tid / 6) % RUNS][tid % 6] = (tid % 6) < 3 ? 0xFFFFFFFFu : 0u;
        if (tid < 2 * 6) s_all[tid / 6][tid % 6] = (tid % 6) < 3 ? 0xFFFFFFFFu : 0u;
        // (no jump into a part's first chain, none measured against a chain of the other part)
        float gap = (tid == part_first || prev < part_first) ? -1.0f : gap0;
        // the RUNS - 1 largest jumps, one after the other: wave maximum of (gap, chain) by shuffles, then over the part's waves
        for (int round = 0; round < RUNS - 1; ++round) {
            unsigned long long key = gap >= 0.0f ? ((unsigned long long)__float_as_uint(gap) << 32) | tid : 0ull;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
                const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(key >> 32), m) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)key, m);
                key = key > o ? key : o;
            }
            if ((tid & 63u) == 0) s_best[tid >> 6] = key;
            __syncthreads();
            unsigned long long best = s_best[wave0];
            for (uint32_t w = 1; w < part_waves; ++w) best = best > s_best[wave0 + w] ? best : s_best[wave0 + w];
            const uint32_t cut = (uint32_t)best;                // (0 if there is no chain left to cut at: a run of no chains)
            if (tid == cut) gap = -1.0f;
            if (tid == part_first) s_cut[part][round] = cut ? cut : (uint32_t)PCR_WORKGROUP_SIZE;
            __syncthreads();
        }
        uint32_t c0 = s_cut[part][0], c1 = s_cut[part][1], c2 = s_cut[part][2];
        static_assert(RUNS == 4, "three cuts, sorted by hand");
        if (c0 > c1) { const uint32_t t = c0; c0 = c1; c1 = t; }
        if (c1 > c2) { const uint32_t t = c1; c1 = c2; c2 = t; }
        if (c0 > c1) { const uint32_t t = c0; c0 = c1; c1 = t; }
        const uint32_t run = (tid >= c0) + (tid >= c1) + (tid >= c2);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!outlier) {
                atomicMin(&s_box[part][run][k], float_order(mn[k]));
                atomicMax(&s_box[part][run][3 + k], float_order(mx[k]));
            }
            // (every chain of the part, straddling ones and garbage tails included: the one window a workgroup would need)
            atomicMin(&s_all[part][k], float_order(mn[k]));
            atomicMax(&s_all[part][3 + k], float_order(mx[k]));
        }
        __syncthreads();
        // the records' words, one thread each: the parts of this shape lie back to back
        const uint32_t nparts = 1u << cfg;
        if (tid < nparts * RUN_WORDS) {
            const uint32_t pi = tid / RUN_WORDS, w = tid % RUN_WORDS;
            uint32_t *out = batch_runs + ((size_t)b * RUN_RECORDS + run_record((int)nparts, (int)pi)) * RUN_WORDS;
            uint32_t v;
            if (w < 4) {
                uint32_t d0 = s_cut[pi][0], d1 = s_cut[pi][1], d2 = s_cut[pi][2];
                if (d0 > d1) { const uint32_t t = d0; d0 = d1; d1 = t; }
                if (d1 > d2) { const uint32_t t = d1; d1 = d2; d2 = t; }
                if (d0 > d1) { const uint32_t t = d0; d0 = d1; d1 = t; }
                v = w == 0 ? d0 : w == 1 ? d1 : w == 2 ? d2 : 0u;
            } else if (w < 4 + RUNS * 6) {
                v = __float_as_uint(float_unorder(s_box[pi][(w - 4) / 6][(w - 4) % 6]));
            } else {
                v = __float_as_uint(float_unorder(s_all[pi][w - 4 - RUNS * 6]));
            }
            out[w] = v;
        }
        __syncthreads();
    }
}

#ifdef PCR_EXP_TIMELINE   /* experiment: per-workgroup time stamps of k_render's phases (100 MHz wall clock) + the hardware slot it ran on */
__device__ unsigned long long g_timeline[8192 * 8];         // (rows 7000..: the prepass blocks' stamps, PCR_PTL)
__device__ unsigned long long g_wave_end[8192 * 16];        // per wave: wall clock at the end of its point loop
#define PCR_TL(slot) do { if (threadIdx.x == 0) g_timeline[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define PCR_TL(slot) do { } while (0)
#endif

#ifdef PCR_EXP_FAR_STATS   /* experiment: what happens to the points outside their windows (counters per launch sequence) */
__device__ unsigned long long g_far[8];   // wave-iterations with such lanes, lanes, pre-read iterations, -, -, waves
#endif
constexpr uint32_t NO_PIXEL = 0xFFFFFFFFu;
typedef float v2f __attribute__((ext_vector_type(2)));

// GENERIC: the checked variant of the decode step, for the batches the prepass put on the second list (flagged
// BF_GENERIC_SLOW_PATH by k_transcode): escape indices are tested against the LDS pool, `wide` table values come from
// global memory. A kernel of its own, so that the ordinary batches' loop carries neither its tests nor its code; the host
// launches it only for streams that have such batches.
// PARTS: workgroups per batch. 1: 1024 threads, the batch's 1024 chains (the reference's shape, render.cu:328). 2: 512 threads,
// chains [512 part, 512 part + 512) -- four workgroups per CU (see DYN_LDS_BYTES_HALF). Workgroup x of the grid draws part
// (x >> 3) & 1 of list entry (x >> 4) * 8 + (x & 7): the hardware deals workgroups round-robin over the eight XCDs, so the two
// halves of a batch run on the same XCD and the second one's table comes out of that XCD's L2.
template <int MODE, int LAYOUT, bool GENERIC, int PARTS = 1>
__global__ void __launch_bounds__(PCR_WORKGROUP_SIZE / PARTS, 8) k_render(RenderArgs a)   // 8 waves/SIMD = two (four) workgroups per CU -> <= 64 VGPRs
{
    constexpr bool COLOR_PASS = MODE == MODE_HQS_COLOR || MODE == MODE_HQS_COLOR_BC7, BC7 = MODE == MODE_HQS_COLOR_BC7;
    constexpr uint32_t THREADS = PCR_WORKGROUP_SIZE / PARTS;
    static_assert(PARTS == 1 || PARTS == 2, "whole batches or halves");
    const uint32_t part = PARTS == 1 ? 0u : (blockIdx.x >> 3) & 1u;
    PCR_TL(0);
#ifdef PCR_EXP_TIMELINE
    if (threadIdx.x == 0) {
        uint32_t hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_timeline[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    // second level of the compaction: which batch is the blockIdx.x-th of my list? Every wave works it out for itself (a
    // 64-lane inclusive prefix sum over the chunk counts, 64 chunks = 2048 batches per round): no barrier, no LDS.
    DrawRec rec;
    {
        const uint32_t lane = threadIdx.x & 63u, chunks = a.order_stride / PREPASS_BATCHES;
        const uint32_t classes = GENERIC ? 1u : a.work_classes;
        uint32_t x = PARTS == 1 ? blockIdx.x : (blockIdx.x >> 4) * 8u + (blockIdx.x & 7u), found = 0xFFFFFFFFu;
        for (uint32_t cls = 0; cls < classes && found == 0xFFFFFFFFu; ++cls) {                  // (uniform; one class unless the frame has a level of detail)
            const uint32_t *cc = a.chunk_count + (GENERIC ? WORK_CLASSES : cls) * PCR_MAX_PREPASS_WORKGROUPS;
            uint32_t before = 0;
            for (uint32_t c0 = 0; c0 < chunks; c0 += 64) {                  // (uniform)
                const uint32_t cnt = c0 + lane < chunks ? cc[c0 + lane] : 0u;
                const uint32_t incl = wave_inclusive_sum(cnt);
                const uint32_t total = wave_read_lane(incl, 63);
                if (x < before + total) {
                    const uint64_t m = __ballot(before + incl > x);        // first chunk whose inclusive count passes x
                    const uint32_t first = (uint32_t)__ffsll((unsigned long long)m) - 1u;
                    const uint32_t excl = wave_read_lane(incl - cnt, first);
                    uint32_t lighter = 0;                                   // the chunk's records of the classes in front of this one
                    for (uint32_t k = 0; k < cls; ++k) lighter += a.chunk_count[k * PCR_MAX_PREPASS_WORKGROUPS + c0 + first];
                    found = (c0 + first) * PREPASS_BATCHES + lighter + (x - before - excl);
                    break;
                }
                before += total;
            }
            x -= before;                                                    // (not found: `before` is the class's total)
        }
        found = __builtin_amdgcn_readfirstlane(found);                      // (the same in every lane: keep it, and the record, scalar)
        if (found == 0xFFFFFFFFu) return;                                   // the grid is sized for "every batch visible"
        rec = a.order[(GENERIC ? (size_t)a.order_stride : 0) + found];
    }
    const uint32_t b = rec.b;
    const uint32_t lod = rec.lod;
    const int npr = (int)(lod & LOD_NPR_MASK);
    const bool use_double = (lod & LOD_DOUBLE) != 0;
    const uint32_t tid = threadIdx.x;
    const uint32_t chain = part * THREADS + tid;            // my chain of the batch's 1024
    // (the lambdas below capture these scalars, not the argument block: with `a` captured by reference and the loop body
    // instantiated five times, hipcc once kept the whole block in scratch memory)
    uint64_t *const g_fb = a.f.fb, *const g_rg = a.f.rg, *const g_ba = a.f.ba;
    uint8_t *const g_tiles = a.f.tiles;
    const uint32_t tiles_x = a.f.tiles_x;
    const int img_w = a.p.width;
    const uint32_t fb_elems = a.f.fb_elems;
    const float m00 = a.p.transform[0], m01 = a.p.transform[1], m02 = a.p.transform[2], m03 = a.p.transform[3];
    const float m10 = a.p.transform[4], m11 = a.p.transform[5], m12 = a.p.transform[6], m13 = a.p.transform[7];

    __shared__ __align__(16) uint32_t s_table[PCR_HUFFMAN_TABLE_SIZE];
    extern __shared__ __align__(16) unsigned char s_dyn[];                  // a.dyn_lds_bytes: escape pool, then the framebuffer window
    int32_t *const s_esc = reinterpret_cast<int32_t *>(s_dyn);

    // decoder table -> LDS (render.cu:383-395), four entries per thread, already packed by k_transcode
    const int32_t *tvalues = a.s.table_values + (size_t)b * PCR_HUFFMAN_TABLE_SIZE;   // only for `wide` entries
#pragma unroll
    for (int k = 0; k < PARTS; ++k)
        reinterpret_cast<uint4 *>(s_table)[tid + k * THREADS] = reinterpret_cast<const uint4 *>(a.s.packed_table + (size_t)b * PCR_HUFFMAN_TABLE_SIZE)[tid + k * THREADS];

    const pcr_gpu_batch *gb = a.s.batches + b;
    const int64_t sep_off = rec.sep_off;                    // :405
    const int32_t *sep = a.s.separate + sep_off;            // batch-relative base (uniform)
    // Reads past the logical end of the escape stream (zero pad included) are defined as 0. The allocation carries
    // PCR_GUARD_WORDS extra zero words that nothing ever writes, so clamping the index to the guard is enough:
    // branch-free loads keep every global load of the loop on one control-flow path (no conservative waits).
    const uint32_t sep_last = (uint32_t)min((int64_t)0x7FFFFFF0, a.s.separate_words + (PCR_GUARD_WORDS - 2) - sep_off);
    auto sep_load = [&](uint32_t i) -> int32_t { return sep[min(i, sep_last)]; };

    // ---- escape words of my chains -> LDS (all of them, or none) ---------------------------------------------
    const int32_t *ssz = a.s.separate_sizes + (size_t)b * 1024;
    // [esc_first, esc_first + esc_count): the escape words of the workgroup's chains inside the batch's
    const uint32_t esc_first = PARTS == 1 || part == 0 ? 0u : rec.esc_mid;
    const uint32_t esc_count = PARTS == 1 ? rec.esc_total : part == 0 ? rec.esc_mid : rec.esc_total - rec.esc_mid;
    // The pool also takes ESC_SLACK words that FOLLOW those in memory, so that the reference's tail over-reads
    // find in LDS what they would find in global memory; reads beyond even that, and workgroups whose escapes do not fit, go
    // to global memory (slow variant of the decode step).
    const uint32_t esc_lds = esc_pool_words(esc_count, PARTS);
    unsigned long long *const s_win = reinterpret_cast<unsigned long long *>(s_dyn + esc_pool_bytes(esc_lds));
    const uint32_t win_cap = (uint32_t)window_capacity(esc_count, COLOR_PASS ? WIN_PIXEL_BYTES_HQS : WIN_PIXEL_BYTES, a.dyn_lds_bytes, PARTS);
    {   // all loads of a thread in flight together (7-8 per thread: the pool of an ordinary batch)
        constexpr int EAGER = PARTS == 1 ? 7 : 8;
        int32_t v[EAGER];
#pragma unroll
        for (int k = 0; k < EAGER; ++k) v[k] = sep_load(esc_first + tid + k * THREADS);
#pragma unroll
        for (int k = 0; k < EAGER; ++k)
            if (tid + k * THREADS < esc_lds) s_esc[tid + k * THREADS] = v[k];
        constexpr int REST = (esc_pool_max(PARTS) - EAGER * (int)THREADS) / (int)THREADS;
        static_assert(REST * (int)THREADS + EAGER * (int)THREADS == esc_pool_max(PARTS), "the pool in whole rounds of the workgroup");
        if (esc_lds > (uint32_t)EAGER * THREADS) {          // (uniform) an escape-heavy batch: the rest of its pool
            int32_t u[REST];
#pragma unroll
            for (int k = 0; k < REST; ++k) u[k] = sep_load(esc_first + (EAGER + k) * THREADS + tid);
#pragma unroll
            for (int k = 0; k < REST; ++k) {
                const uint32_t i = (EAGER + k) * THREADS + tid;
                if (i < esc_lds) s_esc[i] = u[k];
            }
        }
    }
    // :411-413 (batch-relative): my chain's next escape word, as a pointer into the pool (the LDS address travels in one
    // register: nothing to add per read)
    const int32_t *esc_next = s_esc + ((chain ? (uint32_t)ssz[chain - 1] : 0u) - esc_first);

    // ---- framebuffer window of the batch's rectangle -> LDS --------------------------------------------------
    // Up to RUNS rectangles per batch (WinPlan): my chain scatters into the window of its run; all of them live in one array of
    // window pixels, in run order. The window is a per-lane matter (runs do not end at wave boundaries): its origin, size and
    // first pixel sit in vector registers.
    const WinPlan *const plan_p = a.win + ((size_t)b * PARTS + part);   // (read field by field: a local copy indexed in a loop lands in scratch)
    uint32_t wpix = 0;                                      // pixels of all windows together; 0: no window for this batch
    uint32_t wx0 = 0, wy0 = 0, ww = 0, wh = 0, wbase = 0;
    {
        const uint32_t run = (chain >= plan_p->first[0]) + (chain >= plan_p->first[1]) + (chain >= plan_p->first[2]);
        static_assert(RUNS == 4, "three run boundaries");
#pragma unroll
        for (int r = 0; r < RUNS; ++r) {
            const uint32_t xy = plan_p->xy[r], wh2 = plan_p->wh[r];
            if (run == (uint32_t)r) { wx0 = xy & 0xFFFFu; wy0 = xy >> 16; ww = wh2 & 0xFFFFu; wh = wh2 >> 16; wbase = wpix; }
            wpix += (wh2 & 0xFFFFu) * (wh2 >> 16);
        }
    }
    const uint32_t W = (uint32_t)a.p.width;
    // colour pass layout of the same bytes: sums in the framebuffer's own packed format + the depth to test against
    // (RG and BA of a pixel side by side: one address per flush of a run, the second add at offset 8; record `wpix`, behind the
    // last window pixel, is a dummy that the first flush of a chain -- which has no run yet -- adds zeros to)
    unsigned long long *const s_acc = s_win;
    uint32_t *const s_depth = reinterpret_cast<uint32_t *>(s_win + 2 * (win_cap + 1));
    // every pixel of every window in turn: fn(index in the window arrays, index in the framebuffer)
    // (row and column of a thread's first pixel by one multiply-and-correct, then stepped: THREADS pixels on are `qy` rows and `qx`
    // columns, one more row where the column wraps -- the division per pixel was 15 of the merge loop's 25 vector instructions per
    // pixel, 4 % of a 4096x4096 launch whose windows have 6000 pixels)
    auto for_window_pixels = [&](auto fn) __attribute__((always_inline)) {
        uint32_t base = 0;
#pragma unroll 1
        for (int r = 0; r < RUNS; ++r) {                                    // (uniform)
            const uint32_t xy = plan_p->xy[r], wh2 = plan_p->wh[r];
            const uint32_t x0 = xy & 0xFFFFu, y0 = xy >> 16, w2 = wh2 & 0xFFFFu, n = w2 * (wh2 >> 16);
            if (tid < n) {                                                   // (n == 0: no such window)
                const float inv_w2 = 1.0f / (float)max(w2, 1u);
                uint32_t y, x, qy, qx;
                window_row_col(tid, w2, inv_w2, y, x);
                window_row_col(THREADS, w2, inv_w2, qy, qx);                 // (uniform)
                uint32_t gp = (y0 + y) * W + x0 + x;
                const uint32_t step = qy * W + qx, wrap = W - w2;
                for (uint32_t i = tid; i < n; i += THREADS) {
                    fn(base + i, (size_t)gp);
                    x += qx; gp += step;
                    if (x >= w2) { x -= w2; gp += wrap; }
                }
            }
            base += n;
        }
    };
    if (!COLOR_PASS) {
        for (uint32_t i = tid; i < wpix; i += THREADS) s_win[i] = ~0ull;
        // The dummy slot, behind the last window pixel: the window word of every point that has none -- outside the frustum, or
        // inside it but outside its window. It starts as 0, so the depth pre-filter below turns such a lane away without a mask
        // for "the pending point is valid" having to be kept (an off-window point's word is replaced by the global one; what its
        // LDS atomic then leaves in the slot is a depth no smaller than any real one there: harmless, and never merged).
        if (tid == 0) s_win[wpix] = 0ull;
    } else {
        // colour pass: the depths of the rectangles as the depth pass left them, sums zeroed
        for_window_pixels([&](uint32_t i, size_t gp) {
            s_depth[i] = (uint32_t)(a.f.fb[gp] >> 32); s_acc[2 * i] = 0; s_acc[2 * i + 1] = 0;
        });
        if (tid == 0) { s_acc[2 * wpix] = 0; s_acc[2 * wpix + 1] = 0; }
    }

    // ---- my chain's own word sequence (k_transcode) ---------------------------------------------------------
    // The chain's bit stream is the concatenation of its words. w0..w2 are three consecutive words, `spare` = 32 minus
    // the number of bits of w0 already consumed (0..31); far0/far1 are the two words after w2, requested one point
    // before they can be needed. Per point a 64-bit view `bits` of the next unconsumed bits is cut out once, the three
    // symbols (<= 36 bits) and the look-ahead of the next point's first symbol (12 more) index into it, and the words
    // that ran dry are retired afterwards: no refill test, load or wait inside the symbol steps.
    // (Cur/Next of :416-419 are the first two words; the stand-in w0 = 0 is "fully consumed" from the start.)
    // LAYOUT_POINT_WINDOWS: the view of every point was cut by k_transcode; row i of point_windows is read at point i
    // (requested two points earlier) and there is no queue.
    // LAYOUT_WORDS: my wave's rows of the batch's compact block (k_pack_words); a request past the last row the wave's longest
    // chain consumed reads the next wave's rows or the segment's pad (LWC_PAD_ROWS: such a word is requested ahead but never looked at)
    // (a pointer loaded from memory is a generic one to hipcc: flat_load_dword behind a 64-bit vector add per request. Said to be
    // global, the block's base stays in a scalar register pair and a request is global_load_dword v, voffset, s[base:base+1])
    typedef const __attribute__((address_space(1))) char *global_bytes;
    global_bytes lwb = nullptr;
    if (LAYOUT == LAYOUT_WORDS) {
        const uint32_t wave = __builtin_amdgcn_readfirstlane(chain >> 6);
        const uint32_t *wr = a.s.lw_wave_row + (size_t)b * (LWC_WAVES + 1) + wave;
        const uint32_t r0 = wr[0], r1 = wr[1];
        lwb = (global_bytes)(reinterpret_cast<const char *>(a.s.lw_block[b]) + (size_t)r0 * LWC_ROW_BYTES);      // uniform per wave
        (void)r1;
    }
    auto lw_load = [&](uint32_t byte_off) -> uint32_t { return *(const __attribute__((address_space(1))) uint32_t *)(lwb + byte_off); };
    const char *pwb = reinterpret_cast<const char *>(a.s.point_windows) + (size_t)b * PW_BATCH_BYTES;               // uniform
    // the 40-bit window of a point as the top of a 64-bit view: high plane u32, low plane u8 (the 24 bits below are zero).
    // Window rows come through buffer loads: the lane's column offset stays put in a vector register, the row advances in a scalar
    // one (buffer_load_dword v, voffset, s[rsrc], soffset offen) -- no vector add per row and plane (with plain pointers hipcc either
    // advanced a per-lane offset or added the lane's offset to an advancing uniform pointer with a 64-bit vector add). Reads past the
    // batch's block + guard return 0. (HQS -0.7 %, LOD 10 % -1.2 %, LOD 100 % within the noise: the two adds were of the cheap class.)
    const __amdgpu_buffer_rsrc_t pw_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)const_cast<char *>(pwb), 0, (int)(PW_BATCH_BYTES + PW_GUARD_BYTES), 0x00020000);
    uint32_t pw_row_hi = 0, pw_row_lo = PW_HI_BYTES;        // (uniform) byte offsets of the current row in the two planes
    const uint32_t pw_col_hi = chain * 4, pw_col_lo = chain;
    auto pw_load_hi = [&]() -> uint32_t { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(pw_rsrc, (int)pw_col_hi, (int)pw_row_hi, 0); };
#ifdef PCR_EXP_NO_LO      // timing experiment only (wrong frames): what 32-bit windows would save -- the low plane is never read
    auto pw_load_lo = [&]() -> uint32_t { return 0u; };
#else
    auto pw_load_lo = [&]() -> uint32_t { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(pw_rsrc, (int)pw_col_lo, (int)pw_row_lo, 0); };
#endif
#define PCR_PW_NEXT_ROW() do { pw_row_hi += PW_HI_ROW_BYTES; pw_row_lo += PW_LO_ROW_BYTES; } while (0)
    uint32_t lwo = (tid & 63u) * 4;                         // packed words: byte offset of my column in the row of far0
    uint32_t w0 = 0, w1 = 0, w2 = 0, far0 = 0, far1 = 0, spare = 0;
    uint64_t bits;
    uint32_t nwin_hi = 0, nwin_lo = 0;
    if (LAYOUT == LAYOUT_WORDS) {
        w1 = lw_load(lwo); w2 = lw_load(lwo + LWC_ROW_BYTES);
        far0 = lw_load(lwo + 2 * LWC_ROW_BYTES); far1 = lw_load(lwo + 3 * LWC_ROW_BYTES);
        lwo += 2 * LWC_ROW_BYTES;
        bits = ((uint64_t)__builtin_amdgcn_alignbit(w0, w1, spare) << 32) | __builtin_amdgcn_alignbit(w1, w2, spare);
    } else {
        bits = ((uint64_t)pw_load_hi() << 32) | (pw_load_lo() << 24);
        PCR_PW_NEXT_ROW();
        nwin_hi = pw_load_hi(); nwin_lo = pw_load_lo() << 24;
    }
    constexpr uint32_t SFT0 = 50;                           // (bits >> 50) & 0x3FFC = 4 x the top 12 bits of the view
    uint32_t sft = SFT0;

    const int32_t *sv = a.s.start_values + ((size_t)b * 1024 + chain) * 3; // :421-424
    int32_t px = sv[0], py = sv[1], pz = sv[2];

    const double sx = gb->scale_x, sy = gb->scale_y, sz = gb->scale_z;
    const double ox = gb->offset_x - gb->las_min_x, oy = gb->offset_y - gb->las_min_y, oz = gb->offset_z - gb->las_min_z;
    const float fsx = (float)sx, fsy = (float)sy, fsz = (float)sz;          // :469
    const float fox = (float)ox, foy = (float)oy, foz = (float)oz;          // :470

    uint32_t payload = 0;
    if (MODE == MODE_HQS_DEPTH) {
        if (a.p.show_num_points)      payload = (uint32_t)npr;                                   // depth.cu:139-140
        else if (a.p.colorize_chunks) payload = (uint32_t)(a.s.batch_index_base + b);            // :141-142
    }

    // BC1 blocks of my chain (4 blocks of 16 points, 8 bytes each): the block of the current 16-point segment in
    // registers, the next one prefetched a whole segment (16 iterations) before its first use
    const uint2 *cblocks = reinterpret_cast<const uint2 *>(a.s.colors_t) + ((size_t)b * 4096 + chain);        // [segment][chain]
    uint2 cnext = make_uint2(0, 0);
    Bc1Palette pal = {0, 0, 0, 0};
    if (MODE != MODE_HQS_DEPTH && !BC7) cnext = cblocks[0];
    // (BC7 colours, 16 bytes per block: the block of a segment is read at its start -- no register for a prefetched one)
    const uint4 *blocks7 = reinterpret_cast<const uint4 *>(a.s.colors_t) + ((size_t)b * 4096 + chain);
    Bc7Block pal7 = {0, 0, 0, 0, 0};

    // Second half of rasterize() (render.cu:297-301 / depth.cu:148-151 / hqs render.cu:292-313) for the point whose
    // framebuffer word `old` was fetched one iteration earlier, from the LDS window (widx) or from global memory.
    // A stale `old` only makes the filter less selective: framebuffer words never increase during a pass.
    // colour pass: run of contributions to one pixel held in registers
    // (two 16-bit sums per register: a chain adds at most 64 * 255 per channel)
    uint32_t run_pix = NO_PIXEL, run_widx = wpix;           // (no run yet: zeros for the dummy record)
    uint32_t run_rg16 = 0, run_bc16 = 0;                    // r << 16 | g,  b << 16 | count
    auto flush_run = [&]() __attribute__((always_inline)) {
        const unsigned long long run_rg = ((unsigned long long)(run_rg16 >> 16) << 32) | (run_rg16 & 0xFFFFu);
        const unsigned long long run_ba = ((unsigned long long)(run_bc16 >> 16) << 32) | (run_bc16 & 0xFFFFu);
        if (run_widx != NO_PIXEL) {
            // per-batch partial sums in LDS (a batch adds at most 65 536 * 255 < 2^32 per 32-bit half)
            __hip_atomic_fetch_add(&s_acc[2 * run_widx], run_rg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&s_acc[2 * run_widx + 1], run_ba, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            atomicAdd((unsigned long long *)&g_rg[run_pix], run_rg);        // :309-310
            atomicAdd((unsigned long long *)&g_ba[run_pix], run_ba);        // :311-312
        }
    };
    // `valid`: the pending point is inside the frustum; `off`: ... but outside the batch's LDS window (then `pix` is its pixel
    // and `old` came from global memory). Both are lane masks the compiler keeps in scalar registers, so choosing between the
    // LDS and the global path costs no vector instruction. The window word of the pixel: `wp` (basic / depth pass) or the
    // index `w` (colour pass: three planes).
    // colour pass: the contribution of a point that passed the 1 % test (hqs render.cu:297-313)
    auto accumulate = [&](bool off, uint32_t pix, uint32_t w, int point) __attribute__((always_inline)) {
        uint32_t vrg, vbc;                                                     // r << 16 | g,  b << 16 | 1
        if (BC7) {
            const uint32_t rgba = bc7_color(pal7, (uint32_t)point & 15u);
            vrg = __builtin_amdgcn_perm(0u, rgba, 0x0C000C01u);                 // (rgba = 0x00BBGGRR)
            vbc = (rgba & 0x00FF0000u) | 1u;
        } else {
            bc1_contribution(pal, (uint32_t)point & 15u, vrg, vbc);
        }
        // Consecutive points of a chain are Morton neighbours and mostly land in the same pixel: their
        // contributions are summed in registers and written once per run (sums commute, so the totals
        // are unchanged; a chain adds at most 64 * 255 per 32-bit half).
        if (pix == run_pix) {
            run_rg16 += vrg; run_bc16 += vbc;
        } else {
            flush_run();
            run_pix = pix; run_widx = off ? NO_PIXEL : w; run_rg16 = vrg; run_bc16 = vbc;
        }
    };
    auto scatter = [&](bool valid, bool off, uint32_t pix, uint32_t w, unsigned long long *wp, uint32_t depth, uint64_t old, int point) __attribute__((always_inline)) {
        if (COLOR_PASS) {
            const float pw = __uint_as_float(depth);
            const float old_depth = __uint_as_float((uint32_t)(old >> 32));
            if (valid && (double)pw <= (double)old_depth * 1.01) {          // hqs render.cu:296
                accumulate(off, pix, w, point);
            }
            return;
        }
        // pre-read filter (:297-298) on the depth half only: the result is min(depth<<32|payload) over all inside points
        // whatever passes it (min is idempotent), so ties go to the atomic instead of a 64-bit compare here
        if (!valid || depth > (uint32_t)(old >> 32)) return;
#ifdef PCR_EXP_KEY_FILTER
        if (!((((unsigned long long)depth << 32) | (MODE == MODE_BASIC ? bc1_color(pal, (uint32_t)point & 15u) : payload)) < old)) return;
#endif
        const unsigned long long key = ((unsigned long long)depth << 32) | (MODE == MODE_BASIC ? bc1_color(pal, (uint32_t)point & 15u) : payload);   // :299 / depth.cu:139-145
        if (!off) __hip_atomic_fetch_min(wp, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else      atomicMin((unsigned long long *)&g_fb[pix], key);         // :300
    };

    // Basic / depth pass: the same for a pending point described without per-lane flags. Its window word `wp` is the dummy slot
    // unless the point lies in its window; `old` is that word, or the global framebuffer word for the lanes of `off_mask`
    // (inside the frustum, outside the window). A lane with nothing pending holds a dummy word of depth 0 and fails the filter.
    // Scalar instructions are not free here (+16 of them per point: +6 % kernel time, profiles/r03_experiments.md): lane masks
    // carried as 64-bit values and tested as such cost a compare and a branch, bools carried across the loop cost three
    // mask merges each.
    auto scatter_min = [&](__attribute__((address_space(3))) unsigned long long *wp, uint32_t depth, uint32_t old_hi, uint64_t off_mask, uint32_t pix, int point) __attribute__((always_inline)) {
        // pre-read filter (:297-298) on the depth half only: the result is min(depth<<32|payload) over all inside points
        // whatever passes it (min is idempotent), so ties go to the atomic instead of a 64-bit compare here
        if (depth > old_hi) return;
        const unsigned long long key = ((unsigned long long)depth << 32) | (MODE == MODE_BASIC ? bc1_color(pal, (uint32_t)point & 15u) : payload);   // :299 / depth.cu:139-145
        __hip_atomic_fetch_min(wp, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__builtin_amdgcn_inverse_ballot_w64(off_mask)) atomicMin((unsigned long long *)&g_fb[pix], key);   // :300 (rare)
    };

    // End of a point: SFT0 - sft bits were consumed. Retire the 0..2 words that ran dry, pull in far0/far1 (requested a
    // whole point ago), request the next two, cut the next view. u = spare - consumed + 64 lies in [28, 95].
#define PCR_ADVANCE_WORD_WINDOW()                                                          \
    do {                                                                                   \
        if (LAYOUT == LAYOUT_POINT_WINDOWS) {                                              \
            bits = ((uint64_t)nwin_hi << 32) | nwin_lo;                                    \
            sft = SFT0;                                                                    \
            break;                                                                         \
        }                                                                                  \
        const uint32_t u_ = spare + (sft & 63u) + (64u - SFT0);                            \
        const uint32_t k_ = u_ >> 5;               /* 2: no word retired, 1: one, 0: two */ \
        spare = u_ & 31u;                                                                  \
        const uint32_t n0_ = k_ == 2u ? w0 : k_ == 1u ? w1 : w2;                           \
        const uint32_t n1_ = k_ == 2u ? w1 : k_ == 1u ? w2 : far0;                         \
        const uint32_t n2_ = k_ == 2u ? w2 : k_ == 1u ? far0 : far1;                       \
        w0 = n0_; w1 = n1_; w2 = n2_;                                                      \
        lwo += (2u - k_) * LWC_ROW_BYTES;                                                  \
        /* the two words behind w2, requested afresh for every point (round 3 fetched only the words that moved up, 0.7 loads per \
           point, under two branches: loads that may or may not have been issued make every later vmcnt wait a wait for all -- in the \
           depth pass hipcc ended up waiting for this iteration's requests in the middle of the iteration) */ \
        far0 = lw_load(lwo);                                                               \
        far1 = lw_load(lwo + LWC_ROW_BYTES);                                               \
        bits = ((uint64_t)__builtin_amdgcn_alignbit(w0, w1, spare) << 32) | __builtin_amdgcn_alignbit(w1, w2, spare); \
        sft = SFT0;                                                                        \
    } while (0)

    uint64_t pend_valid_mask = 0;                           // colour pass: lanes whose pending point is inside the frustum
    uint64_t pend_off_mask = 0;                             // lanes whose pending point is inside the frustum but outside its window
    const uint32_t whole_x0 = plan_p->whole_xy & 0xFFFFu, whole_y0 = plan_p->whole_xy >> 16;
    const uint32_t whole_w = plan_p->whole_wh & 0xFFFFu, whole_h = plan_p->whole_wh >> 16;
    const bool mostly_outside = plan_p->mostly_outside != 0;    // (uniform) the batch and most of its neighbours lie mostly outside their windows
    uint32_t pend_pix = NO_PIXEL, pend_w = 0, pend_depth = 0;
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    lds_u64 *const s_win_mine = (lds_u64 *)(s_win + wbase); // my run's window
    lds_u64 *const s_dummy = (lds_u64 *)(s_win + wpix);     // (see the window's set-up)
    const uint32_t dummy_idx = wpix - wbase;                // ... as an index into my run's window
    lds_u64 *pend_p = s_dummy;                              // basic / depth pass: the pending point's word in the LDS window
    uint64_t pend_old = 0;                                  // (colour pass)
    uint32_t pend_old_hi = 0;                               // basic / depth pass: the depth half of the pending point's framebuffer word

    const float *M = a.p.transform;
    const float fw = (float)a.p.width, fh = (float)a.p.height;
    // Loop constants kept in vector registers on purpose: a v_fma_f32 / v_sub_u32 whose operands are all VGPRs or inline
    // constants issues in ~2.3 cycles per wave64, the same instruction with an SGPR operand in ~4.2
    // (tools/exp/instr_rate2.hip). The kernel has the registers to spare (<= 64 for eight waves per SIMD).
    // (only where there are registers to spare: not in the colour pass, which keeps its run of sums in registers, not with
    // the packed-words variant's five-word queue, not in the checked variant)
#ifdef PCR_EXP_WORDS_VGPR_CONSTANTS
    constexpr bool VGPR_CONSTANTS = !COLOR_PASS && !GENERIC;
#else
    constexpr bool VGPR_CONSTANTS = !COLOR_PASS && LAYOUT == LAYOUT_POINT_WINDOWS && !GENERIC;
#endif
    auto in_vgpr_f = [](float v) { if (!VGPR_CONSTANTS) return v; float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(v)); return r; };
    auto in_vgpr_u = [](uint32_t v) { if (!VGPR_CONSTANTS) return v; uint32_t r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(v)); return r; };
    const float m30 = in_vgpr_f(M[12]), m31 = in_vgpr_f(M[13]), m32 = in_vgpr_f(M[14]), m33 = in_vgpr_f(M[15]);   // the w row
    const uint32_t v_wx0 = wx0, v_wy0 = wy0;                // (per-lane values: in vector registers anyway)

    PCR_TL(1);
    __syncthreads();        // table, escapes and window are visible
    PCR_TL(2);

#ifdef PCR_EXP_PROLOGUE_ONLY   /* experiment only: cost of the per-batch set-up and the window merge (results are wrong) */
    const int npr_run = a.p.reserved == 12345 ? npr : 0;
#else
    const int npr_run = npr;
#endif
    uint32_t toff_ahead = (uint32_t)(bits >> (sft & 63u)) & 0x3FFCu;
    uint32_t e_ahead = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_table) + toff_ahead);
    // (point windows: the second symbol's entry of the point ahead as well, see the loop)
    uint32_t sft_ahead = SFT0 - e_ahead;                    // :439 (the whole entry: byte 0 is the length)
    uint32_t toff1_ahead = ((uint32_t)(bits >> 32) >> (sft_ahead & 31u)) & 0x3FFCu;
    uint32_t e1_ahead = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_table) + toff1_ahead);
    // DECODE_AHEAD (point windows, basic / depth pass): all three entries of a point are requested during the iteration before
    // it, spread over that iteration (see the loop); these are the ones of point 0
    // Packed words (round 4): the same order. A point's view is cut from the chain's word queue once the lengths of the point before
    // it are known -- at the top of the iteration, where all three of its entries have arrived -- so the queue advances THERE,
    // the words it needs were requested a whole iteration earlier, and the three entries of point i+1 are requested from the new
    // view during iteration i exactly as with the point windows (round 3's form, kept behind PCR_EXP_WORDS_SERIAL for A/B: every
    // table read of a point hung on the one before it, two round trips per point exposed).
#ifdef PCR_EXP_WORDS_SERIAL
    constexpr bool DECODE_AHEAD = LAYOUT == LAYOUT_POINT_WINDOWS;
#else
    constexpr bool DECODE_AHEAD = true;
#endif
    uint32_t toff2_ahead = 0, e2_ahead = 0;
    if (DECODE_AHEAD) {
        toff2_ahead = (uint32_t)(bits >> ((sft_ahead - e1_ahead) & 63u)) & 0x3FFCu;
        e2_ahead = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_table) + toff2_ahead);
    }

    // ---- the point loop ---------------------------------------------------------------------------------------------------------
    // A wave issues in order and the decode is a chain of LDS round trips (table entry -> length -> next key -> next
    // entry). What can run inside those waits does: the table entry of a symbol is requested one step ahead, the scatter of
    // the previous point runs under the read for this point's second symbol (-2.5 % kernel time for moving one call), the
    // framebuffer word of a point is requested a whole point before it is used. A deeper software pipeline (decode k+2 |
    // project k+1 | scatter k, the projection split over the second and third table read) was built and measured: +4 %,
    // hipcc spends the 64 registers on copies between the stages (profiles/r02_experiments.md).

    // One symbol step (:430-451) of the point being decoded. `e` is the table entry of this symbol, fetched one step ahead;
    // (bits >> sft) & 0x3FFC is 4 x the 12-bit window of :431-433 (== ((L|R) & mask) >> 20) at the current position, i.e.
    // the byte offset of an entry. Returns the decoded delta.
    // esc_first (the third symbol's form): the escape word, if any, is requested BEFORE the next table entry. LDS results
    // return in issue order, so its value can be waited for (lgkmcnt(1)) while the look-ahead entry of the next point's first
    // symbol -- needed only at the top of the next iteration -- stays in flight.
    // The delta a table entry stands for (:435-438): the entry's own value, or -- escape -- the chain's next escape word.
    auto entry_value = [&](uint32_t e, uint32_t toff) __attribute__((always_inline)) -> uint32_t {
        int32_t val = (int32_t)e >> TE_VALUE_SHIFT;                         // the delta itself (v_ashrrev_i32)
        if (val == TE_SLOW_VALUE) {                                         // escape or wide
            if (!GENERIC) {                                                 // every such entry is an escape whose word is in the pool
                val = *esc_next++;
            } else if (e & TE_ESCAPE) {                                     // :438
                if (esc_next < s_esc + esc_lds) {
                    val = *esc_next;
                } else {
                    // outside the pool: the load is consumed inside this branch so no pending VMEM result leaves it
                    val = sep_load(esc_first + (uint32_t)(esc_next - s_esc));
                    asm volatile("; escape word from global memory %0" : "+v"(val));
                }
                ++esc_next;
            } else {
                val = tvalues[toff >> 2];
                asm volatile("; wide table value from global memory %0" : "+v"(val));
            }
        }
        return (uint32_t)val;
    };
    auto table_entry = [&](uint32_t toff) __attribute__((always_inline)) -> uint32_t {
        return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_table) + toff);
    };
    // DECODE_AHEAD, top of iteration i (the entries e0..e2 of point i have arrived): request the first entry of point i+1. Point
    // windows: its key is the top of the window that has been in registers for an iteration. Packed words: the queue advances by
    // what point i consumed (the three lengths: byte 0 of the entries, subtracted whole -- only the low six bits of `sft` count),
    // the view of point i+1 is cut, its top twelve bits are the key. (A macro, not a lambda: a closure that captures the word queue
    // by reference put the kernel's whole parameter block into scratch memory.)
#define PCR_NEXT_FIRST_ENTRY(e0_, e1_, e2_)                                                 \
    do {                                                                                    \
        if (LAYOUT == LAYOUT_POINT_WINDOWS) {                                               \
            toff_ahead = (nwin_hi >> (SFT0 & 31u)) & 0x3FFCu;                               \
        } else {                                                                            \
            sft = SFT0 - (e0_) - (e1_) - (e2_);                                             \
            PCR_ADVANCE_WORD_WINDOW();                                                      \
            toff_ahead = ((uint32_t)(bits >> 32) >> (SFT0 & 31u)) & 0x3FFCu;                \
        }                                                                                   \
        e_ahead = table_entry(toff_ahead);                                                  \
    } while (0)
    // LAYOUT_WORDS: the symbols of a point AND the first symbol of the next one are cut from one 64-bit view, so every
    // table read hangs on the one before it. esc_first (the third symbol's form): the escape word, if any, is requested
    // BEFORE the next table entry. LDS results return in issue order, so its value can be waited for (lgkmcnt(1)) while the
    // look-ahead entry of the next point's first symbol -- needed only at the top of the next iteration -- stays in flight.
    auto symbol_step = [&](auto esc_first) __attribute__((always_inline)) -> uint32_t {
        constexpr bool ESC_FIRST = decltype(esc_first)::value;
        const uint32_t e = e_ahead, toff = toff_ahead;                      // :435-436
        // :439. The whole entry is subtracted: byte 0 is the length, and only the low six bits of `sft` are ever used (the
        // 64-bit shift below takes its count modulo 64; 14 <= true sft <= 50) -- a plain v_sub_u32
        sft -= e;
        toff_ahead = (uint32_t)(bits >> (sft & 63u)) & 0x3FFCu;
        if (!ESC_FIRST) e_ahead = table_entry(toff_ahead);
        const uint32_t val = entry_value(e, toff);
        if (ESC_FIRST) e_ahead = table_entry(toff_ahead);
#ifdef PCR_EXP_PAD_VALU   /* experiment: PCR_EXP_PAD_VALU extra independent VALU instructions per symbol step */
        {
            uint32_t pad = tid;
#pragma unroll
            for (int k = 0; k < PCR_EXP_PAD_VALU; ++k) asm volatile("v_add_u32 %0, %0, %0" : "+v"(pad));
        }
#endif
        return val;
    };

#if defined(PCR_EXP_PRIO_LAST)   /* experiment: the last workgroups to start share their CU with an older one that would starve them: raise them */
    if (blockIdx.x + PCR_EXP_PRIO_LAST >= gridDim.x) __builtin_amdgcn_s_setprio(3);
#endif
#if defined(PCR_EXP_PRIO_WAVE)   /* experiment: the hardware issues oldest-first; give a workgroup's later waves the higher priority */
    switch ((tid >> 8) & 3u) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
#endif
    for (int seg = 0; seg < npr_run; seg += 16) {
#if defined(PCR_EXP_PRIO_SEG)    /* experiment: a wave that is behind (an earlier segment) goes first */
      switch (seg >> 4) {
          case 0: __builtin_amdgcn_s_setprio(3); break;
          case 1: __builtin_amdgcn_s_setprio(2); break;
          case 2: __builtin_amdgcn_s_setprio(1); break;
          default: __builtin_amdgcn_s_setprio(0); break;
      }
#endif
      // Segment boundary: the point still pending belongs to the previous BC1 block, so it is scattered before the
      // block registers rotate (its framebuffer word has been in flight for the whole decode of the last point).
      if (COLOR_PASS) { scatter(__builtin_amdgcn_inverse_ballot_w64(pend_valid_mask), __builtin_amdgcn_inverse_ballot_w64(pend_off_mask), pend_pix, pend_w, nullptr, pend_depth, pend_old, seg - 1); pend_valid_mask = 0; pend_off_mask = 0; }
      else { scatter_min(pend_p, pend_depth, pend_old_hi, pend_off_mask, pend_pix, seg - 1); pend_p = s_dummy; pend_off_mask = 0;
             // (the dummy slot's zero, READ rather than set: with an LDS read behind the last table request on this way into the loop
             // as well as on the way round it, hipcc's wait for that entry at the top of an iteration leaves one result in flight)
             pend_old_hi = reinterpret_cast<__attribute__((address_space(3))) const volatile uint32_t *>(s_dummy)[1]; }
      if (MODE != MODE_HQS_DEPTH) {
          if (BC7) pal7 = bc7_block(blocks7[(seg >> 4) * PCR_WORKGROUP_SIZE]);
          else {
              pal = bc1_palette(cnext);                 // once per 16 points instead of once per surviving point
                                                        // (palettes expanded at load time, 16 bytes per block: measured twice, rounds 2 and 4: nothing)
              cnext = cblocks[min((seg >> 4) + 1, 3) * PCR_WORKGROUP_SIZE];
          }
      }
      const int seg_end = min(seg + 16, npr_run);
#pragma unroll 1
      for (int i = seg; i < seg_end; ++i) {                                 // :428
        uint32_t fetched_hi = 0, fetched_lo = 0;
        if (LAYOUT == LAYOUT_POINT_WINDOWS) {
            // row i+2, requested at the top of point i and taken over at its very end: a whole point of latency cover
            // (past row 63 of a plane lies the batch's other plane, the next batch, or the guard)
            PCR_PW_NEXT_ROW();
            fetched_hi = pw_load_hi(); fetched_lo = pw_load_lo();
        }
#if defined(PCR_EXP_PAD_FAST) || defined(PCR_EXP_PAD_SLOW) || defined(PCR_EXP_PAD_SALU)   /* experiment: what one more instruction per point costs */
        {
            uint32_t pad = tid;
            (void)pad;
#ifdef PCR_EXP_PAD_FAST
#pragma unroll
            for (int k = 0; k < PCR_EXP_PAD_FAST; ++k) asm volatile("v_add_u32 %0, %0, %0" : "+v"(pad));
#endif
#ifdef PCR_EXP_PAD_SLOW
#pragma unroll
            for (int k = 0; k < PCR_EXP_PAD_SLOW; ++k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(pad));
#endif
#ifdef PCR_EXP_PAD_SALU
            uint32_t spad = (uint32_t)i;
#pragma unroll
            for (int k = 0; k < PCR_EXP_PAD_SALU; ++k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(spad));
#endif
        }
#endif
        uint32_t d0, d1, d2;
        uint32_t sft1_next = 0;
        if (DECODE_AHEAD) {
            // All three table entries of point i were requested while point i-1 was projected: nothing of the decode's dependent
            // chain (entry -> length -> key -> entry) is waited for here. The escape words of the three symbols, if any, are
            // requested back to back; the scatter of point i-1 runs under them. The entries of point i+1 (its window has been in
            // registers since the top of the last iteration) are requested in three places of this iteration, each a table
            // round trip after the one before: here, in front of the dot products, behind the division.
            const uint32_t e0 = e_ahead, toff0 = toff_ahead, e1 = e1_ahead, toff1 = toff1_ahead, e2 = e2_ahead, toff2 = toff2_ahead;
            if (!GENERIC) {
                // Everything that hangs on a result of the last iteration -- the three entries, the framebuffer word of point i-1 --
                // is looked at BEFORE this iteration issues its first LDS instruction: results return in issue order and hipcc
                // counts a read under a branch (an escape word's) as maybe not issued, so a wait placed behind such reads waits for
                // them as well. What is tested becomes a lane mask in a scalar register pair.
                int32_t v0 = (int32_t)e0 >> TE_VALUE_SHIFT, v1 = (int32_t)e1 >> TE_VALUE_SHIFT, v2 = (int32_t)e2 >> TE_VALUE_SHIFT;
                const uint64_t esc0 = __builtin_amdgcn_ballot_w64(v0 == TE_SLOW_VALUE), esc1 = __builtin_amdgcn_ballot_w64(v1 == TE_SLOW_VALUE),
                               esc2 = __builtin_amdgcn_ballot_w64(v2 == TE_SLOW_VALUE);
                // the pre-read filter of scatter_min / the 1 % test of the colour pass (hqs render.cu:296)
                const uint64_t draw = COLOR_PASS
                    ? pend_valid_mask & __builtin_amdgcn_ballot_w64((double)__uint_as_float(pend_depth) <= (double)__uint_as_float((uint32_t)(pend_old >> 32)) * 1.01)
                    : __builtin_amdgcn_ballot_w64(pend_depth <= pend_old_hi);
                if (__builtin_expect(__builtin_amdgcn_inverse_ballot_w64(esc0), 1)) v0 = *esc_next++;               // :438 (every such entry is an escape whose word is in the pool)
                if (__builtin_expect(__builtin_amdgcn_inverse_ballot_w64(esc1), 1)) v1 = *esc_next++;
                if (__builtin_expect(__builtin_amdgcn_inverse_ballot_w64(esc2), 1)) v2 = *esc_next++;
                PCR_NEXT_FIRST_ENTRY(e0, e1, e2);
                if (COLOR_PASS) {
                    if (__builtin_amdgcn_inverse_ballot_w64(draw)) accumulate(__builtin_amdgcn_inverse_ballot_w64(pend_off_mask), pend_pix, pend_w, i - 1);
                } else if (__builtin_amdgcn_inverse_ballot_w64(draw)) {                         // second half of rasterize() for point i-1
                    const unsigned long long key = ((unsigned long long)pend_depth << 32) | (MODE == MODE_BASIC ? bc1_color(pal, (uint32_t)(i - 1) & 15u) : payload);
                    __hip_atomic_fetch_min(pend_p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (__builtin_amdgcn_inverse_ballot_w64(pend_off_mask)) atomicMin((unsigned long long *)&g_fb[pend_pix], key);   // :300 (rare)
                }
                d0 = (uint32_t)v0; d1 = (uint32_t)v1; d2 = (uint32_t)v2;
            } else {
                d0 = entry_value(e0, toff0);                                    // :430
                d1 = entry_value(e1, toff1);
                d2 = entry_value(e2, toff2);
                PCR_NEXT_FIRST_ENTRY(e0, e1, e2);
                if (COLOR_PASS) scatter(__builtin_amdgcn_inverse_ballot_w64(pend_valid_mask), __builtin_amdgcn_inverse_ballot_w64(pend_off_mask), pend_pix, pend_w, nullptr, pend_depth, pend_old, i - 1);
                else scatter_min(pend_p, pend_depth, pend_old_hi, pend_off_mask, pend_pix, i - 1);
            }
        } else if (LAYOUT == LAYOUT_POINT_WINDOWS) {
            // The first symbol's key is the top of the point's own window: its entry was requested a whole point ago (the one of
            // point i+1, whose window is already in registers, is requested now). The second symbol's entry was requested before
            // the point began as well: at the end of the iteration before, as soon as the first entry gave the second key (the
            // second symbol starts and ends inside the window's first 32 bits: a 32-bit shift, the hardware takes the count
            // modulo 32 and SFT0 - 32 = 18). Left on the dependent chain of a point: entry 1 -> length -> key 2 -> entry 2, one LDS
            // round trip (the third symbol may reach into the low plane's byte).
            const uint32_t e0 = e_ahead, toff0 = toff_ahead, e1 = e1_ahead, toff1 = toff1_ahead;
            sft = sft_ahead;
            toff_ahead = (nwin_hi >> (SFT0 & 31u)) & 0x3FFCu;
            e_ahead = table_entry(toff_ahead);
            d0 = entry_value(e0, toff0);                                    // :430
            // second half of rasterize() for point i-1, under the table read of this point's second symbol: its framebuffer
            // word has been in flight since the end of the last iteration
            // (the colour pass, which carries a run of sums and has no register to spare, scatters after the third symbol)
            if (!COLOR_PASS) scatter_min(pend_p, pend_depth, pend_old_hi, pend_off_mask, pend_pix, i - 1);
            sft -= e1;
            const uint32_t toff2 = (uint32_t)(bits >> (sft & 63u)) & 0x3FFCu;
            const uint32_t e2 = table_entry(toff2);
            d1 = entry_value(e1, toff1);
            d2 = entry_value(e2, toff2);
        } else {
            constexpr std::integral_constant<bool, false> table_first{};
            constexpr std::integral_constant<bool, true> escape_first{};
            d0 = symbol_step(table_first);                                  // :430
            if (!COLOR_PASS) scatter_min(pend_p, pend_depth, pend_old_hi, pend_off_mask, pend_pix, i - 1);
            d1 = symbol_step(table_first);
            d2 = symbol_step(escape_first);
        }
        if (COLOR_PASS && !DECODE_AHEAD) scatter(__builtin_amdgcn_inverse_ballot_w64(pend_valid_mask), __builtin_amdgcn_inverse_ballot_w64(pend_off_mask), pend_pix, pend_w, nullptr, pend_depth, pend_old, i - 1);
        px = (int32_t)((uint32_t)px + d0);                                  // :454-456, :463
        py = (int32_t)((uint32_t)py + d1);
        pz = (int32_t)((uint32_t)pz + d2);
        if (!(DECODE_AHEAD && LAYOUT == LAYOUT_WORDS)) PCR_ADVANCE_WORD_WINDOW();     // (packed words, decode ahead: the queue advanced at the top)
#ifdef PCR_EXP_NO_RASTER   /* experiment only: decode cost alone (results are wrong) */
        if ((px ^ py ^ pz) == 0x7fffffff && i == 63) g_fb[tid] = 0;
        if (LAYOUT == LAYOUT_POINT_WINDOWS) { nwin_hi = fetched_hi; nwin_lo = fetched_lo << 24; }
        continue;
#endif
        // (per-point temporaries and the three parts of the projection live inside the loop body: nothing of a point's
        // projection is carried into the next iteration -- with ix / iy declared outside, every iteration copied them)
        float fx, fy, fz;
        float qx, qy, qw;
        bool inside;
        uint64_t cand_mask;                 // lanes whose point is inside the frustum, as a 64-bit lane mask (a scalar register pair)
        int ix, iy;
        // first half of rasterize() (:278-287), part 1: the three dot products and the inside test without dividing. For finite
        // w > 0 the correctly rounded quotient RN(x/w) lies in [-1,1] exactly when |x| <= w (if x > w then x/w >= 1 + ulp(w)/w >
        // 1 + 2^-24, which rounds above 1); the compare is false for a NaN and for w < 0.
        auto project_dots = [&]() __attribute__((always_inline)) {
            qx = __fmaf_rn(m03, 1.0f, __fmaf_rn(m02, fz, __fmaf_rn(m01, fy, m00 * fx)));       // dot4(M + 0, ...)
            qy = __fmaf_rn(m13, 1.0f, __fmaf_rn(m12, fz, __fmaf_rn(m11, fy, m10 * fx)));       // dot4(M + 4, ...)
            qw = __fmaf_rn(m33, 1.0f, __fmaf_rn(m32, fz, __fmaf_rn(m31, fy, m30 * fx)));       // dot4(M + 12, ...), operands in VGPRs
            // (lane masks as 64-bit scalars: a ballot of a compare IS the compare's result register, mask logic is one scalar
            // instruction each, and nothing is merged where control flow joins)
            cand_mask = __builtin_amdgcn_ballot_w64(fabsf(qx) <= qw) & __builtin_amdgcn_ballot_w64(fabsf(qy) <= qw);
        };
        // part 2: the division and the pixel (:279-285). The division is the IEEE sequence hipcc emits for `/` with its range
        // scaling removed, shared reciprocal, x and y packed; it is bit-identical to `/` for w in [2^-64, 2^64) (no intermediate
        // leaves the normal range unless |x/w| < 2^-36, where the pixel is the screen centre whatever the last bits are). A
        // wave in which some candidate's w lies outside that range (or is 0) takes `/` for all lanes.
        auto project_divide = [&]() __attribute__((always_inline)) {
            // (both forms run for every lane, candidate or not -- a lane that is not one produces a pixel nobody looks at: the
            // instructions issue for the wave anyway, and with ix / iy written on every path hipcc neither masks the block nor
            // carries last iteration's pixel along for the lanes that skipped it)
            {
                const float r0 = __builtin_amdgcn_rcpf(qw);
                const float r1 = __fmaf_rn(__fmaf_rn(-qw, r0, 1.0f), r0, r0);
                const v2f xy = {qx, qy}, rr = {r1, r1}, nw = {-qw, -qw};
                const v2f q0 = xy * rr;
                const v2f q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(nw, q0, xy), rr, q0);
                const v2f q2 = __builtin_elementwise_fma(__builtin_elementwise_fma(nw, q1, xy), rr, q1);
                const v2f half = {0.5f, 0.5f}, size = {fw, fh};
                const v2f img = __builtin_elementwise_fma(q2, half, half) * size;       // :283
                ix = (int)img.x; iy = (int)img.y;                                       // :284 (a candidate's pixel index is always < fb_elems here)
            }
            // (the plain `/` as an override behind the fast form, not as its `else`: an if / else of a uniform condition costs the
            // common path two branches and three scalar instructions here, an `if` alone one of each)
            const uint64_t w_ok_mask = __builtin_amdgcn_ballot_w64((__float_as_uint(qw) - 0x1F800000u) < 0x40000000u);   // 2^-64 <= w < 2^64
            if (__builtin_expect((cand_mask & ~w_ok_mask) != 0, 0)) {
                const float nx = qx / qw, ny = qy / qw;
                ix = (int)(__fmaf_rn(nx, 0.5f, 0.5f) * fw);                 // :283-284
                iy = (int)(__fmaf_rn(ny, 0.5f, 0.5f) * fh);
                // (inside <=> candidate here as well: for w > 0 the correctly rounded quotient lies in [-1,1] exactly when |x| <= w,
                // a NaN or w < 0 fails both forms, and a pixel of a point inside is below fb_elems -- Appendix C.2. w == 0 with
                // x == y == 0 -- a point exactly in the eye, or a degenerate matrix -- passes |x| <= w and is no candidate: :296 `w <= 0`)
                cand_mask &= __builtin_amdgcn_ballot_w64(qw > 0.0f);
            }
            inside = __builtin_amdgcn_inverse_ballot_w64(cand_mask);
        };
        // part 3: the point becomes the pending one; its framebuffer word is requested now and consumed an iteration later:
        // from the LDS window if the pixel lies in the batch's rectangle (nearly always), from global memory otherwise (:297)
        auto project_request = [&]() __attribute__((always_inline)) {
            if (!COLOR_PASS) {
                // No flags per lane, no block under `if (inside)`: every lane works out a window position (garbage for a lane that is
                // not inside), the masks decide. The window word is read by EVERY lane (the dummy slot's if it has none): with the
                // read issued on every path hipcc knows how many LDS results are outstanding at the top of the next iteration and
                // waits for the look-ahead table entry alone (lgkmcnt(1)) instead of for this read as well.
                const uint32_t rx = (uint32_t)ix - v_wx0, ry = (uint32_t)iy - v_wy0;
                // (bitwise: `&&` would put the two compares under a branch of their own)
                const uint64_t in_mask = cand_mask & __builtin_amdgcn_ballot_w64(rx < ww) & __builtin_amdgcn_ballot_w64(ry < wh);
#ifdef PCR_EXP_NO_OFFWIN   /* timing experiment only (wrong frames): what the points outside their LDS window cost */
                pend_off_mask = 0;
#else
                pend_off_mask = cand_mask & ~in_mask;
#endif
                pend_depth = __float_as_uint(qw);                                       // :287
                pend_p = s_win_mine + (__builtin_amdgcn_inverse_ballot_w64(in_mask) ? (uint32_t)__umul24(ry, ww) + rx : dummy_idx);
                // (the depth half alone: the filter looks at nothing else, and a 64-bit read whose low half nobody wants had hipcc reuse
                // that register while the read was in flight -- and wait for it)
                pend_old_hi = reinterpret_cast<__attribute__((address_space(3))) const uint32_t *>(pend_p)[1];
                if (__builtin_expect(pend_off_mask != 0, 0)) {              // (uniform, rare: the window plan keeps nearly every point inside)
                    // Is one of them outside the rectangle the prepass marked the dirty tiles under as well (FrameView::tiles)? The
                    // garbage tails of chains are: a thousandth of the points, anywhere on the screen. Whole-wave
                    // compares under the scalar branch: a mask updated inside the divergent branch below is no longer uniform -- and
                    // the packed-words variant then drew wrong frames (round 3; convergent operations in divergent control flow).
                    const uint64_t stray_now = g_tiles ? pend_off_mask & (__builtin_amdgcn_ballot_w64((uint32_t)ix - whole_x0 >= whole_w) |
                                                                          __builtin_amdgcn_ballot_w64((uint32_t)iy - whole_y0 >= whole_h)) : 0;
                    const bool sample = mostly_outside;                             // (uniform) see below
#ifdef PCR_EXP_FAR_STATS
                    if ((threadIdx.x & 63u) == 0) { atomicAdd(&g_far[0], 1ull); atomicAdd(&g_far[1], (unsigned long long)__builtin_popcountll(pend_off_mask)); if (sample) atomicAdd(&g_far[2], 1ull); }
#endif
                    if (__builtin_amdgcn_inverse_ballot_w64(pend_off_mask)) {
                        pend_pix = (uint32_t)(ix + iy * img_w);                         // :285
                        if (__builtin_amdgcn_inverse_ballot_w64(stray_now)) {
                            // its own tile (ndc == 1.0 maps to column W, Appendix C.2: in the linear framebuffer that is column 0 of the next row)
                            const uint32_t mx = ix >= img_w ? 0u : (uint32_t)ix, my = ix >= img_w ? (uint32_t)iy + 1u : (uint32_t)iy;
                            g_tiles[(my >> TILE_H_SHIFT) * tiles_x + (mx >> TILE_W_SHIFT)] = 1;
                        }
                        // The pre-read of the global framebuffer word (:297) is a filter in front of the global atomic -- min is
                        // idempotent, the frame is the same without it -- that costs the wave a memory round trip per point. It stops
                        // 70-85 % of the points it sees, whatever the frame (tools/exp/far_stats.py); what differs is how many there
                        // are. In a Morton-sorted stream 0.2 % (1080p) to 5 % (close-up) of a frame's wave-iterations have such lanes,
                        // 9-31 of them, concentrated in a few waves that the round trips turn into the launch's stragglers: without
                        // the load 4096x4096 is 8 % faster, a close-up 7 %, the benchmark frame 2.6 %. In an unsorted stream (a batch
                        // is a strip across the scene, ten points per pixel) it is every iteration and 57 lanes, and unfiltered the
                        // frame takes 1.7 x the time in atomics. A few batches of that kind among the others are cheaper unfiltered (the
                        // atomic units have room), a frame full of them is not: the prepass votes per 32 neighbouring batches
                        // (WinPlan::mostly_outside).
                        pend_old_hi = 0xFFFFFFFFu;
                        if (sample) {
                            // (consumed inside the branch: no load is pending where the paths join, so hipcc's waits for the window
                            // rows, requested two points ahead, stay exact)
                            uint32_t seen = __hip_atomic_load(reinterpret_cast<const uint32_t *>(&g_fb[pend_pix]) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            asm volatile("; framebuffer word of a point outside its window %0" : "+v"(seen));
                            pend_old_hi = seen;
                        }
                    }
                }
                return;
            }
            // colour pass, the same without flags per lane: every lane works out a pixel and a window position (garbage for a lane
            // that is not inside), the masks decide; a lane with no window word reads the window's first one (any valid address
            // will do: its result is not looked at, or replaced below)
            const uint32_t rx = (uint32_t)ix - v_wx0, ry = (uint32_t)iy - v_wy0;
            const uint64_t in_mask = cand_mask & __builtin_amdgcn_ballot_w64(rx < ww) & __builtin_amdgcn_ballot_w64(ry < wh);
            pend_valid_mask = cand_mask;
            pend_off_mask = cand_mask & ~in_mask;
            pend_depth = __float_as_uint(qw);                                           // :287
            pend_pix = (uint32_t)(ix + iy * img_w);                         // the colour pass names its runs by pixel
            pend_w = wbase + (__builtin_amdgcn_inverse_ballot_w64(in_mask) ? ry * ww + rx : 0u);
            pend_old = (uint64_t)s_depth[pend_w] << 32;
            if (__builtin_expect(pend_off_mask != 0, 0)) {
                if (__builtin_amdgcn_inverse_ballot_w64(pend_off_mask)) pend_old = __hip_atomic_load(&g_fb[pend_pix], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        };

        // (the float form always, the double form as an override for the batches that want it: see project_divide)
        fx = __fmaf_rn((float)px, fsx, fox);                                // :529-531
        fy = __fmaf_rn((float)py, fsy, foy);
        fz = __fmaf_rn((float)pz, fsz, foz);
        if (use_double) {                                                   // :459-461
            fx = (float)__fma_rn((double)px, sx, ox);
            fy = (float)__fma_rn((double)py, sy, oy);
            fz = (float)__fma_rn((double)pz, sz, oz);
        }
        if (DECODE_AHEAD) {                         // second entry of point i+1 (`bits` is its window by now; the first has arrived)
            sft1_next = SFT0 - e_ahead;
            toff1_ahead = ((uint32_t)(bits >> 32) >> (sft1_next & 31u)) & 0x3FFCu;
            e1_ahead = table_entry(toff1_ahead);
        }
        project_dots();                                                     // first half of rasterize() (:278-287) for point i
        project_divide();
        if (DECODE_AHEAD) {                         // third entry of point i+1
            toff2_ahead = (uint32_t)(bits >> ((sft1_next - e1_ahead) & 63u)) & 0x3FFCu;
            e2_ahead = table_entry(toff2_ahead);
        }
#ifdef PCR_EXP_NO_FBLOAD   /* experiment only: decode + projection, no framebuffer traffic (results are wrong) */
        if (inside && ix == 0x12345678) g_fb[tid] = __float_as_uint(qw);
        pend_valid_mask = 0; pend_p = s_dummy; pend_old_hi = 0; pend_off_mask = 0;
#else
        project_request();
#endif
        if (LAYOUT == LAYOUT_POINT_WINDOWS && !DECODE_AHEAD) {       // `bits` is the next point's window by now, e_ahead its first entry (requested at the top)
            sft_ahead = SFT0 - e_ahead;
            toff1_ahead = ((uint32_t)(bits >> 32) >> (sft_ahead & 31u)) & 0x3FFCu;
            e1_ahead = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(s_table) + toff1_ahead);
        }
        if (LAYOUT == LAYOUT_POINT_WINDOWS) { nwin_hi = fetched_hi; nwin_lo = fetched_lo << 24; }
      }
    }
    if (COLOR_PASS) { scatter(__builtin_amdgcn_inverse_ballot_w64(pend_valid_mask), __builtin_amdgcn_inverse_ballot_w64(pend_off_mask), pend_pix, pend_w, nullptr, pend_depth, pend_old, npr_run - 1); flush_run(); }
    else scatter_min(pend_p, pend_depth, pend_old_hi, pend_off_mask, pend_pix, npr_run - 1);

    // merge the window into the global framebuffer: rows of the rectangle are contiguous, so the 64 lanes of a wave
    // hit a handful of cache lines; only pixels this batch improved issue an atomic
#if defined(PCR_EXP_PRIO_WAVE) || defined(PCR_EXP_PRIO_SEG) || defined(PCR_EXP_PRIO_LAST)
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef PCR_EXP_FAR_STATS
    if ((threadIdx.x & 63u) == 0) atomicAdd(&g_far[5], 1ull);
#endif
#ifdef PCR_EXP_TIMELINE
    if ((threadIdx.x & 63u) == 0) g_wave_end[(size_t)blockIdx.x * 16 + (threadIdx.x >> 6)] = wall_clock64();
#endif
    PCR_TL(3);
    if (wpix) {
        __syncthreads();
        PCR_TL(4);
        for_window_pixels([&](uint32_t i, size_t gp) {
            if (COLOR_PASS) {
                const unsigned long long vba = s_acc[2 * i + 1];
                if (vba) {
                    atomicAdd((unsigned long long *)&a.f.rg[gp], s_acc[2 * i]);
                    atomicAdd((unsigned long long *)&a.f.ba[gp], vba);
                }
            } else {
                const unsigned long long v = s_win[i];
                unsigned long long *g = (unsigned long long *)&a.f.fb[gp];
                // (no read of the global word in front of the atomic: the compare only saved atomics -- min is idempotent -- at the
                // price of a memory round trip per row of the window while the workgroup holds its LDS; 4096x4096 -4 %, 1080p -0.6 %)
                if (v != ~0ull) atomicMin(g, v);                            // (a pixel no point of the batch reached issues nothing)
            }
        });
    }
    PCR_TL(5);
}

// ------------------------------------------------------------------------------------------------
// 10-10-10 path ("loop_las_cuda", modules/compute_loop_las_cuda/render.cu:130-442)
//
// Streaming kernel: one workgroup per batch of 65 536 points, thread t takes the point quads t, t+1024, ... so every
// level buffer is read with 16-byte-per-lane loads (the reference's USE_PREFETCH uint4 idea, render.cu:204-327; the
// point -> thread mapping is free because the result is a min over all points). Level 2..4 batches read 4 B per
// point, level 1 reads 8 B, level 0 reads 12 B. Scatter uses the same LDS framebuffer window as the Huffman kernel.
// ------------------------------------------------------------------------------------------------
struct LasView {
    const pcr_xyz_batch *batches;
    const uint32_t *xyz12, *xyz8, *xyz4;
    int64_t num_batches;
};

struct LasArgs {
    pcr_render_params p;
    LasView s;
    FrameView f;
    int32_t *level;           // [nB] -1 culled, 0 / 1 / 2..4
    uint2 *win;               // [nB]
    pcr_render_stats *stats;
    int win_capacity;
    // dense list of the batches k_las_render draws (not culled, not the last one), compacted by the prepass per workgroup of
    // LAS_PREPASS_BATCHES batches in the file's order, as the Huffman methods' lists are (RenderArgs::order): order[wg * 256 ..],
    // chunk_count[wg]; k_las_render's workgroup x finds the x-th entry with a wave-wide prefix sum over the chunk counts
    uint32_t *order;          // [chunks * LAS_PREPASS_BATCHES]
    uint32_t *chunk_count;    // [LAS_CLASSES][chunks]
    uint32_t chunks;
};
constexpr int LAS_PREPASS_BATCHES = PREPASS_THREADS;     // one lane per batch
// Heaviest first, as the Huffman lists' classes (RenderArgs::work_classes): class 0 = batches whose screen rectangle the LDS window does
// not hold (a batch of a tile-ordered cloud that straddles the end of one row of tiles and the start of the next: every point then
// goes through a global pre-read and a global atomic, 4-6 x the time of a batch with a window -- a few dozen of them in a frame, and
// drawn where the file has them they end the launch alone on their CUs: 0.23 ms where the wave lifetimes sum to 0.11), class 1 = the rest.
// A chunk's records are stored class after class, chunk_count[class][chunk] counts them.
constexpr int LAS_CLASSES = 2;

// (*partial: the rectangle had to be cut down to the capacity, or the batch gets no window at all although it is on screen -- its
// points, or many of them, take the global path)
__device__ __forceinline__ uint2 window_rect(const pcr_render_params &p, const float *bmin, const float *bmax, int capacity, bool *partial = nullptr)
{
    if (partial) *partial = true;
    float minx = 3.0e38f, maxx = -3.0e38f, miny = 3.0e38f, maxy = -3.0e38f;
    const float fw = (float)p.width, fh = (float)p.height;
    for (int c = 0; c < 8; ++c) {
        const float x = (c & 1) ? bmax[0] : bmin[0], y = (c & 2) ? bmax[1] : bmin[1], z = (c & 4) ? bmax[2] : bmin[2];
        const float w = dot4(p.transform + 12, x, y, z, 1.0f);
        if (!(w > 1.0e-6f)) return make_uint2(0, 0);
        const float sx = (dot4(p.transform + 0, x, y, z, 1.0f) / w * 0.5f + 0.5f) * fw;
        const float sy = (dot4(p.transform + 4, x, y, z, 1.0f) / w * 0.5f + 0.5f) * fh;
        minx = fminf(minx, sx); maxx = fmaxf(maxx, sx); miny = fminf(miny, sy); maxy = fmaxf(maxy, sy);
    }
    if (!(maxx >= -1.0f && maxy >= -1.0f && minx <= fw + 1.0f && miny <= fh + 1.0f)) return make_uint2(0, 0);
    int x0 = max(0, (int)floorf(fmaxf(minx, -2.0f)) - 1), x1 = min(p.width - 1, (int)floorf(fminf(maxx, fw + 2.0f)) + 1);
    int y0 = max(0, (int)floorf(fmaxf(miny, -2.0f)) - 1), y1 = min(p.height - 1, (int)floorf(fminf(maxy, fh + 2.0f)) + 1);
    int ww = x1 - x0 + 1, wh = y1 - y0 + 1;
    if (partial && ww > 0 && wh > 0 && (int64_t)ww * wh <= capacity) *partial = false;      // the whole rectangle fits
    if (ww > 0 && wh > 0 && (int64_t)ww * wh > capacity && (int64_t)ww * wh <= 16 * (int64_t)capacity) {
        const float sc = sqrtf((float)capacity / ((float)ww * (float)wh));
        const int nw = max(1, (int)floorf((float)ww * sc)), nh = max(1, (int)floorf((float)wh * sc));
        x0 += (ww - nw) / 2; y0 += (wh - nh) / 2; ww = nw; wh = nh;
    }
    if (ww > 0 && wh > 0 && ww * wh <= capacity && x0 < 65536 && y0 < 65536)
        return make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)ww | ((uint32_t)wh << 16));
    return make_uint2(0, 0);
}

__device__ __forceinline__ void las_prepass_batch(const LasArgs &a, int64_t b, pcr_render_stats &st, bool *heavy);

__global__ void __launch_bounds__(PREPASS_THREADS) k_las_prepass(LasArgs a)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    pcr_render_stats st = {0, 0, 0, 0};
    uint32_t cls = LAS_CLASSES;                                             // not drawn
    if (b < a.s.num_batches) {
        bool heavy = false;
        las_prepass_batch(a, b, st, &heavy);
        if (a.level[b] >= 0 && b != a.s.num_batches - 1) cls = heavy ? 0u : 1u;      // render.cu:153-155, :201-202
    }
    commit_stats(st, a.stats, blockIdx.x);
    // compaction of the batches to draw, order preserving inside a class: a ballot per wave and class, the waves' counts through LDS
    __shared__ uint32_t s_wave_count[LAS_CLASSES][PREPASS_THREADS / 64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint64_t mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)LAS_CLASSES; ++k) {
        const uint64_t m = __ballot(cls == k);
        if (lane == 0) s_wave_count[k][wave] = (uint32_t)__popcll(m);
        if (cls == k) mine = m;
    }
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (uint32_t k = 0; k < (uint32_t)LAS_CLASSES; ++k) {
        uint32_t total = 0;
#pragma unroll
        for (uint32_t w = 0; w < PREPASS_THREADS / 64; ++w) { if (k == cls && w < wave) base += s_wave_count[k][w]; total += s_wave_count[k][w]; }
        if (k < cls) base += total;                                         // my chunk's records of the classes in front of mine
        if (threadIdx.x == 0) a.chunk_count[k * a.chunks + blockIdx.x] = total;
    }
    if (cls < (uint32_t)LAS_CLASSES) a.order[(size_t)blockIdx.x * LAS_PREPASS_BATCHES + base + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull))] = (uint32_t)b;
}

__device__ __forceinline__ void las_prepass_batch(const LasArgs &a, int64_t b, pcr_render_stats &st, bool *heavy)
{
    const pcr_xyz_batch g = a.s.batches[b];
    const pcr_render_params &p = a.p;
    const float bmin[3] = { g.min_x, g.min_y, g.min_z }, bmax[3] = { g.max_x, g.max_y, g.max_z };
    st.batches_total += 1;
    if (p.enable_frustum_culling) {                                          // render.cu:153-155
        const float *M = p.transform;
#define T(i) M[((i) % 4) * 4 + ((i) / 4)]
        bool in = plane_accepts(T(3) - T(0), T(7) - T(4), T(11) - T(8),  T(15) - T(12), bmin, bmax)
               && plane_accepts(T(3) + T(0), T(7) + T(4), T(11) + T(8),  T(15) + T(12), bmin, bmax)
               && plane_accepts(T(3) + T(1), T(7) + T(5), T(11) + T(9),  T(15) + T(13), bmin, bmax)
               && plane_accepts(T(3) - T(1), T(7) - T(5), T(11) - T(9),  T(15) - T(13), bmin, bmax)
               && plane_accepts(T(3) - T(2), T(7) - T(6), T(11) - T(10), T(15) - T(14), bmin, bmax)
               && plane_accepts(T(3) + T(2), T(7) + T(6), T(11) + T(10), T(15) + T(14), bmin, bmax);
#undef T
        if (!in) {
            a.level[b] = -1;
            st.batches_culled += 1;
            return;
        }
    }
    // :157-197
    const float cx = 0.5f * (bmin[0] + bmax[0]), cy = 0.5f * (bmin[1] + bmax[1]), cz = 0.5f * (bmin[2] + bmax[2]);
    const float dx = bmin[0] - bmax[0], dy = bmin[1] - bmax[1], dz = bmin[2] - bmax[2];
    const float rad = sqrtf(__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx)));
    float vc[4], ve[4], pc[4], pe[4];
    for (int r = 0; r < 4; ++r) vc[r] = dot4(p.world_view + 4 * r, cx, cy, cz, 1.0f);
    ve[0] = vc[0] + rad; ve[1] = vc[1] + 0.0f; ve[2] = vc[2] + 0.0f; ve[3] = vc[3] + 0.0f;
    for (int r = 0; r < 4; ++r) {
        pc[r] = dot4(p.proj + 4 * r, vc[0], vc[1], vc[2], vc[3]);
        pe[r] = dot4(p.proj + 4 * r, ve[0], ve[1], ve[2], ve[3]);
    }
    const float fw = (float)p.width, fh = (float)p.height;
    const float scx = fw * (0.5f * (pc[0] / pc[3] + 1.0f)), scy = fh * (0.5f * (pc[1] / pc[3] + 1.0f));
    const float sex = fw * (0.5f * (pe[0] / pe[3] + 1.0f)), sey = fh * (0.5f * (pe[1] / pe[3] + 1.0f));
    const float ddx = sex - scx, ddy = sey - scy;
    const float px = sqrtf(__fmaf_rn(ddy, ddy, ddx * ddx));
    a.level[b] = px < 100.0f ? 4 : px < 200.0f ? 3 : px < 500.0f ? 2 : px < 10000.0f ? 1 : 0;
    if (b != a.s.num_batches - 1)                                            // the last workgroup returns early (:201-202)
        st.points_iterated += PCR_POINTS_PER_BATCH;
    a.win[b] = window_rect(p, bmin, bmax, a.win_capacity, heavy);
}

// Round 4: brought to k_render's standard. The window starts EMPTY (round 1 copied the framebuffer's words into it: a global read
// of every window pixel in front of the barrier); a point's depth is tested against the depth half of its window word, read a
// whole point before it is used, and only then is the 64-bit key put together for the ds_min_u64 (round 1: a 64-bit LDS read and
// compare in front of the atomic, the round trip exposed four times per quad); "inside the frustum" / "inside the window" are
// lane masks in scalar registers, a lane without a window word reads a dummy slot whose depth 0 turns it away; the merge issues
// its atomic without reading the pixel first; batches are drawn through the prepass's compacted list. Points outside the window
// keep their pre-read of the global word (strip-ordered clouds put most points there: unfiltered 8 x the time in atomics), now
// requested a point ahead like the window word. Reference: modules/compute_loop_las_cuda/render.cu:204-327 (USE_PREFETCH loop),
// :108-128 (rasterize).
__global__ void __launch_bounds__(PCR_WORKGROUP_SIZE, 8) k_las_render(LasArgs a)
{
    // which batch is the blockIdx.x-th of the list? (every wave for itself: a 64-lane inclusive prefix sum over the chunk counts)
    uint32_t b;
    {
        const uint32_t lane = threadIdx.x & 63u;
        // Heaviest class first only while it is a minority (a few stragglers among batches with windows: started first, they run beside
        // everything else). A frame made of such batches (a close-up: every batch larger on screen than its window) is bound by its
        // global atomics, and drawn class by class it was 18 % slower than in the file's order -- then the chunks are walked in order.
        uint32_t heavy_total = 0, all_total = 0;
        for (uint32_t c0 = 0; c0 < a.chunks; c0 += 64) {                    // (uniform)
            uint32_t h = c0 + lane < a.chunks ? a.chunk_count[c0 + lane] : 0u, l = c0 + lane < a.chunks ? a.chunk_count[a.chunks + c0 + lane] : 0u;
            h = wave_read_lane(wave_inclusive_sum(h), 63); l = wave_read_lane(wave_inclusive_sum(l), 63);
            heavy_total += h; all_total += h + l;
        }
        const bool by_class = heavy_total * 4u < all_total;
        static_assert(LAS_CLASSES == 2, "heavy / light");
        uint32_t x = blockIdx.x, found = 0xFFFFFFFFu;
        for (uint32_t cls = 0; cls < (by_class ? 2u : 1u) && found == 0xFFFFFFFFu; ++cls) {       // (uniform)
            uint32_t before = 0;
            for (uint32_t c0 = 0; c0 < a.chunks; c0 += 64) {                // (uniform)
                const uint32_t c = c0 + lane;
                // by class: the chunk's records of this class; in the file's order: all of the chunk's records (heavy ones first inside it)
                const uint32_t cnt = c < a.chunks ? (by_class ? a.chunk_count[cls * a.chunks + c] : a.chunk_count[c] + a.chunk_count[a.chunks + c]) : 0u;
                const uint32_t incl = wave_inclusive_sum(cnt);
                const uint32_t total = wave_read_lane(incl, 63);
                if (x < before + total) {
                    const uint64_t m = __ballot(before + incl > x);
                    const uint32_t first = (uint32_t)__ffsll((unsigned long long)m) - 1u;
                    const uint32_t excl = wave_read_lane(incl - cnt, first);
                    const uint32_t lighter = by_class && cls == 1u ? a.chunk_count[c0 + first] : 0u;      // the chunk's heavy records lie in front
                    found = (c0 + first) * LAS_PREPASS_BATCHES + lighter + (x - before - excl);
                    break;
                }
                before += total;
            }
            x -= before;                                                    // (not found: `before` is the class's total)
        }
        found = __builtin_amdgcn_readfirstlane(found);
        if (found == 0xFFFFFFFFu) return;                                   // the grid is sized for "every batch drawn"
        b = a.order[found];
    }
    const int level = a.level[b];
    const uint32_t tid = threadIdx.x;
    __shared__ __align__(16) unsigned long long s_win[WIN_PIXELS + 1];

    const uint2 wr = a.win[b];
    const uint32_t wx0 = wr.x & 0xFFFFu, wy0 = wr.x >> 16, ww = wr.y & 0xFFFFu, wh = wr.y >> 16;
    const uint32_t wpix = ww * wh;
    const uint32_t W = (uint32_t)a.p.width;
    const float inv_ww = 1.0f / (float)max(ww, 1u);
    for (uint32_t i = tid; i < wpix; i += PCR_WORKGROUP_SIZE) s_win[i] = ~0ull;
    if (tid == 0) s_win[wpix] = 0ull;                                        // the dummy slot: depth 0, no point passes it
    const pcr_xyz_batch g = a.s.batches[b];
    const float div = level >= 2 ? 1024.0f : 1073741824.0f;                  // STEPS_10BIT / STEPS_30BIT
    const float sx = (g.max_x - g.min_x) / div, sy = (g.max_y - g.min_y) / div, sz = (g.max_z - g.min_z) / div;   // :145, :345
    // (the w row of the matrix in vector registers, as in k_render: a v_fma_f32 whose operands are all VGPRs issues in ~2.3 cycles,
    // with an SGPR operand in ~4.2; the x and y rows run as packed operations with scalar pairs)
    auto in_vgpr = [](float v) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(v)); return r; };
    const float *M = a.p.transform;
    const float m00 = M[0], m01 = M[1], m02 = M[2], m03 = M[3];
    const float m10 = M[4], m11 = M[5], m12 = M[6], m13 = M[7];
    const float m30 = in_vgpr(M[12]), m31 = in_vgpr(M[13]), m32 = in_vgpr(M[14]), m33 = in_vgpr(M[15]);
    const float vsx = sx, vsy = sy, vsz = sz, vox = g.min_x, voy = g.min_y, voz = g.min_z;
    const float fw = (float)a.p.width, fh = (float)a.p.height;
    const int img_w = a.p.width;
    uint64_t *const g_fb = a.f.fb;
    const size_t base = (size_t)b * PCR_POINTS_PER_BATCH;
    const uint4 *q4 = reinterpret_cast<const uint4 *>(a.s.xyz4 + base);
    const uint4 *q8 = reinterpret_cast<const uint4 *>(a.s.xyz8 + base);
    const uint4 *q12 = reinterpret_cast<const uint4 *>(a.s.xyz12 + base);
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    lds_u64 *const s_w = (lds_u64 *)s_win;
    // the pending point: projected, its window word's depth half requested; scattered while the next point is projected
    lds_u64 *pend_p = s_w + wpix;
    uint32_t pend_depth = 0, pend_old_hi = 0, pend_index = 0, pend_pix = 0;
    uint64_t pend_off_mask = 0;                                              // lanes whose pending point is inside the frustum but outside the window
    __syncthreads();

    auto scatter_pending = [&]() __attribute__((always_inline)) {
        // rasterize, second half (:119-126): the pre-read filter on the depth half, then the min (ties go to the atomic: min is idempotent)
        if (pend_depth <= pend_old_hi) {
            const unsigned long long key = ((unsigned long long)pend_depth << 32) | pend_index;             // :119-120
            // (either / or: in a strip-ordered cloud most lanes are outside the window, and 50 of them on the dummy slot's one
            // address would serialise the LDS atomic)
            if (__builtin_amdgcn_inverse_ballot_w64(pend_off_mask)) atomicMin((unsigned long long *)&g_fb[pend_pix], key);   // :123-126
            else __hip_atomic_fetch_min(pend_p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto point = [&](uint32_t X, uint32_t Y, uint32_t Z, uint32_t index) __attribute__((always_inline)) {
        const float x = __fmaf_rn((float)X, vsx, vox), y = __fmaf_rn((float)Y, vsy, voy), z = __fmaf_rn((float)Z, vsz, voz);
        // rasterize, first half (:108-118); projection as in k_render (exact inside test without dividing, shared-reciprocal division)
        const float qx = __fmaf_rn(m03, 1.0f, __fmaf_rn(m02, z, __fmaf_rn(m01, y, m00 * x)));
        const float qy = __fmaf_rn(m13, 1.0f, __fmaf_rn(m12, z, __fmaf_rn(m11, y, m10 * x)));
        const float qw = __fmaf_rn(m33, 1.0f, __fmaf_rn(m32, z, __fmaf_rn(m31, y, m30 * x)));
        uint64_t cand_mask = __builtin_amdgcn_ballot_w64(fabsf(qx) <= qw) & __builtin_amdgcn_ballot_w64(fabsf(qy) <= qw);
        int ix, iy;
        {
            const float r0 = __builtin_amdgcn_rcpf(qw);
            const float r1 = __fmaf_rn(__fmaf_rn(-qw, r0, 1.0f), r0, r0);
            const v2f xy = {qx, qy}, rr = {r1, r1}, nw = {-qw, -qw};
            const v2f q0 = xy * rr;
            const v2f q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(nw, q0, xy), rr, q0);
            const v2f q2 = __builtin_elementwise_fma(__builtin_elementwise_fma(nw, q1, xy), rr, q1);
            const v2f half = {0.5f, 0.5f}, size = {fw, fh};
            const v2f img = __builtin_elementwise_fma(q2, half, half) * size;
            ix = (int)img.x; iy = (int)img.y;
        }
        const uint64_t w_ok_mask = __builtin_amdgcn_ballot_w64((__float_as_uint(qw) - 0x1F800000u) < 0x40000000u);   // 2^-64 <= w < 2^64
        if (__builtin_expect((cand_mask & ~w_ok_mask) != 0, 0)) {           // (uniform, practically never) the plain `/` for all lanes
            const float nx = qx / qw, ny = qy / qw;
            ix = (int)(__fmaf_rn(nx, 0.5f, 0.5f) * fw);
            iy = (int)(__fmaf_rn(ny, 0.5f, 0.5f) * fh);
            cand_mask &= __builtin_amdgcn_ballot_w64(qw > 0.0f);            // (w == 0 with x == y == 0 passes |x| <= w: rasterize rejects w <= 0, :113)
        }
        scatter_pending();                                                   // the point before this one: its window word has arrived by now
        const uint32_t rx = (uint32_t)ix - wx0, ry = (uint32_t)iy - wy0;
        const uint64_t in_mask = cand_mask & __builtin_amdgcn_ballot_w64(rx < ww) & __builtin_amdgcn_ballot_w64(ry < wh);
        pend_off_mask = cand_mask & ~in_mask;
        pend_depth = __float_as_uint(qw);
        pend_index = index;
        pend_p = s_w + (__builtin_amdgcn_inverse_ballot_w64(in_mask) ? (uint32_t)__umul24(ry, ww) + rx : wpix);
        pend_old_hi = reinterpret_cast<__attribute__((address_space(3))) const uint32_t *>(pend_p)[1];
        if (__builtin_expect(pend_off_mask != 0, 0)) {                      // (uniform) some lane's point lies outside the window: the global word's depth (:123)
            if (__builtin_amdgcn_inverse_ballot_w64(pend_off_mask)) {
                pend_pix = (uint32_t)(ix + iy * img_w);
                // (consumed inside the branch: no load is pending where the paths join, so hipcc's waits for the next quad's words,
                // requested a whole quad ahead, stay counted instead of becoming vmcnt(0) in front of every scatter)
                uint32_t seen = __hip_atomic_load(reinterpret_cast<const uint32_t *>(&g_fb[pend_pix]) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                asm volatile("; framebuffer word of a point outside its window %0" : "+v"(seen));
                pend_old_hi = seen;
            }
        }
    };

    // One copy of the loop per number of level arrays read (4 / 8 / 12 bytes per point): the copy that reads one array does not
    // carry the registers of the other two (with one loop for all three, the 64-register budget of eight waves per SIMD spilled).
    auto quads = [&](auto arrays_) __attribute__((always_inline)) {
        constexpr int ARRAYS = decltype(arrays_)::value;
        uint4 n4 = q4[tid], n8 = make_uint4(0, 0, 0, 0), n12 = make_uint4(0, 0, 0, 0);
        if (ARRAYS >= 2) n8 = q8[tid];
        if (ARRAYS >= 3) n12 = q12[tid];
#pragma unroll 1
        for (int i = 0; i < PCR_POINTS_PER_BATCH / 4 / PCR_WORKGROUP_SIZE; ++i) {
            const uint4 c4 = n4, c8 = n8, c12 = n12;
            const uint32_t quad = tid + (uint32_t)i * PCR_WORKGROUP_SIZE;
            {   // next quad in flight while this one is projected (the last iteration re-reads its own: no branch around the loads)
                const uint32_t nq = min(quad + PCR_WORKGROUP_SIZE, (uint32_t)(PCR_POINTS_PER_BATCH / 4 - 1));
                n4 = q4[nq];
                if (ARRAYS >= 2) n8 = q8[nq];
                if (ARRAYS >= 3) n12 = q12[nq];
            }
            const uint32_t w4[4] = { c4.x, c4.y, c4.z, c4.w }, w8[4] = { c8.x, c8.y, c8.z, c8.w }, w12[4] = { c12.x, c12.y, c12.z, c12.w };
            const uint32_t index0 = (uint32_t)base + quad * 4;
            if (ARRAYS == 1) {                                                   // :381-392 (levels 2..4)
#pragma unroll
                for (int j = 0; j < 4; ++j) point(w4[j] & 1023u, (w4[j] >> 10) & 1023u, (w4[j] >> 20) & 1023u, index0 + j);
            } else {                                                             // :333-378 (level 1: the 12-byte array's bits are zero)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    point(((w4[j] & 1023u) << 20) | ((w8[j] & 1023u) << 10) | (w12[j] & 1023u),
                          (((w4[j] >> 10) & 1023u) << 20) | (((w8[j] >> 10) & 1023u) << 10) | ((w12[j] >> 10) & 1023u),
                          (((w4[j] >> 20) & 1023u) << 20) | (((w8[j] >> 20) & 1023u) << 10) | ((w12[j] >> 20) & 1023u), index0 + j);
            }
        }
    };
    if (level >= 2)      quads(std::integral_constant<int, 1>{});
    else if (level == 1) quads(std::integral_constant<int, 2>{});
    else                 quads(std::integral_constant<int, 3>{});
    scatter_pending();
    if (wpix) {
        __syncthreads();
        for (uint32_t i = tid; i < wpix; i += PCR_WORKGROUP_SIZE) {
            uint32_t y, x;
            window_row_col(i, ww, inv_ww, y, x);
            const unsigned long long v = s_win[i];
            if (v != ~0ull) atomicMin((unsigned long long *)&a.f.fb[(size_t)(wy0 + y) * W + wx0 + x], v);      // (a pixel no point reached issues nothing)
        }
    }
}

// modules/compute_loop_las_cuda/resolve.cu (every pixel; the reference's floor(w/16) x floor(h/16) launch skips the rest)
__global__ void __launch_bounds__(256) k_las_resolve(int width, int height, const uint64_t *fb, const uint32_t *rgba_points,
                                                     uint32_t *rgba)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= width || y >= height) return;
    const int pix = x + y * width;
    const uint32_t id = (uint32_t)fb[pix];
    rgba[pix] = id < 0x7FFFFFFFu ? rgba_points[id] : PCR_BACKGROUND_COLOR;
}

// ------------------------------------------------------------------------------------------------
// CLEAR block (huffman_hqs.h:266-270): fb <- all ones, and RG/BA <- 0 when a colour pass has written them; one launch,
// 16-byte stores
// ------------------------------------------------------------------------------------------------
// Tile flags at a CLEAR (FrameView::tiles): the half the coming frame marks starts empty; the other half -- tiles whose image pixels
// may hold something -- takes over what was marked so far (a separate resolve may have drawn it), or everything if nobody kept track.
struct TileFlags { uint8_t *cur, *prev; uint32_t *cur_all, *prev_all; uint32_t ntiles, e_cur, e_prev; int tracked; };
__device__ __forceinline__ void clear_tile_flags(const TileFlags &t)
{
    if (!t.cur) return;
    for (uint32_t i = threadIdx.x; i < t.ntiles; i += blockDim.x) {
        t.prev[i] = t.tracked ? (uint8_t)(t.prev[i] | t.cur[i]) : (uint8_t)1;
        t.cur[i] = 0;
    }
    // ("everything": the host gives the emptied half a fresh epoch, so its word needs no reset)
    if (threadIdx.x == 0 && (!t.tracked || *t.cur_all == t.e_cur)) *t.prev_all = t.e_prev;
}

__device__ __forceinline__ void clear_block(uint64_t *fb, uint64_t *rg, uint64_t *ba, size_t n, uint64_t empty,
                                            uint32_t block, uint32_t blocks)
{
    const size_t pairs = n / 2, stride = (size_t)blocks * blockDim.x;
    const ulonglong2 ones = make_ulonglong2(empty, empty), zero = make_ulonglong2(0ull, 0ull);
    for (size_t i = (size_t)block * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        reinterpret_cast<ulonglong2 *>(fb)[i] = ones;
        if (rg) reinterpret_cast<ulonglong2 *>(rg)[i] = zero;
        if (ba) reinterpret_cast<ulonglong2 *>(ba)[i] = zero;
    }
    if ((n & 1) && block == 0 && threadIdx.x == 0) {
        fb[n - 1] = empty;
        if (rg) rg[n - 1] = 0;
        if (ba) ba[n - 1] = 0;
    }
}

__global__ void __launch_bounds__(256) k_clear(uint64_t *fb, uint64_t *rg, uint64_t *ba, size_t n, uint64_t empty, TileFlags tiles)
{
    clear_block(fb, rg, ba, n, empty, blockIdx.x, gridDim.x);
    if (blockIdx.x == gridDim.x - 1) clear_tile_flags(tiles);
}

// CLEAR and the prepass of the frame's first pass in one launch (pcr_frame_begin): the first `prepass_blocks` workgroups
// do the cull/LOD work (it touches no framebuffer), the others fill. Saves the prepass's 5 us and a launch gap per frame.
static_assert(PREPASS_THREADS == 256, "k_frame_begin runs both bodies with 256 threads");
__global__ void __launch_bounds__(256) k_frame_begin(RenderArgs a, uint32_t prepass_blocks, uint64_t *fb, uint64_t *rg,
                                                     uint64_t *ba, size_t n, uint64_t empty, TileFlags tiles)
{
    if (blockIdx.x < prepass_blocks) { lod_prepass_block(a, blockIdx.x); return; }
    clear_block(fb, rg, ba, n, empty, blockIdx.x - prepass_blocks, gridDim.x - prepass_blocks);
    if (blockIdx.x == gridDim.x - 1) clear_tile_flags(tiles);
}

// ------------------------------------------------------------------------------------------------
// HBM ceiling probes (pcr_measure_hbm): a streaming read (16-byte loads, xor-folded so nothing is optimised away) and a
// streaming copy, over buffers larger than the 256 MiB Infinity Cache. The practical ceiling the roofline is set beside.
// ------------------------------------------------------------------------------------------------
typedef uint32_t hbm_vec4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_hbm_read(const hbm_vec4 *src, size_t n16, uint32_t *sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    hbm_vec4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        acc ^= __builtin_nontemporal_load(src + i);
    const uint32_t f = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (f == 0x9E3779B9u) *sink = f;          // practically never: keeps the loads alive
}

__global__ void __launch_bounds__(256) k_hbm_copy(const hbm_vec4 *src, hbm_vec4 *dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// ------------------------------------------------------------------------------------------------
// resolve (resolve.cu:149-191, huffman_hqs/resolve.cu:2-47)
// ------------------------------------------------------------------------------------------------
// The basic method's resolve depends on the framebuffer word only, not on the pixel's position: a contiguous range of
// pixels (a slice of the frame a rank owns after the all-to-all merge) resolves on its own. Same arithmetic as k_resolve<false>.
__global__ void __launch_bounds__(256) k_resolve_range(int show_num_points, int colorize_chunks, const uint64_t *fb,
                                                       size_t count, uint32_t *rgba)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t id = (uint32_t)fb[i];
        uint32_t color = PCR_BACKGROUND_COLOR;
        if (id < 0xFFFFFFFFu) {
            if (show_num_points) {
                const uint32_t shade = (uint32_t)(((double)(float)(int)id / 64.0) * 255.0);
                color = (shade << 24) | (shade << 16) | (shade << 8) | shade;
            } else if (colorize_chunks) {
                color = id * 1234567u;
            } else {
                color = id;
            }
        }
        rgba[i] = color;
    }
}

// The HQS resolve of a contiguous range of pixels (a slice of the frame a rank owns after the sums were reduce-scattered):
// same arithmetic as k_resolve<true>, which depends on the pixel's three words only.
__global__ void __launch_bounds__(256) k_resolve_range_hqs(int show_num_points, int colorize_chunks, const uint64_t *fb,
                                                           const uint64_t *rg, const uint64_t *ba, size_t count, uint32_t *rgba)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t id = (uint32_t)fb[i];
        uint32_t color = PCR_BACKGROUND_COLOR;
        if (id < 0xFFFFFFFFu) {
            if (show_num_points) {
                const uint32_t shade = (uint32_t)(((double)(float)(int)id / 512.0) * 255.0);
                color = (shade << 24) | (shade << 16) | (shade << 8) | shade;
            } else if (colorize_chunks) {
                color = id * 1234567u;
            } else {
                const uint64_t vrg = rg[i], vba = ba[i];
                const uint32_t cnt = (uint32_t)vba;
                if (cnt == 0) color = 0;
                else color = (((uint32_t)(vba >> 32) / cnt) << 16) | (((uint32_t)vrg / cnt) << 8) | ((uint32_t)(vrg >> 32) / cnt);
            }
        }
        rgba[i] = color;
    }
}

// slices[0 .. S) <- element-wise min over `ns` slices of S words laid out back to back (what a rank holds after the
// all-to-all: everyone's copy of the slice of the frame it owns)
__global__ void __launch_bounds__(256) k_merge_min_slices(uint64_t *slices, int ns, size_t S)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t m = slices[i];
        for (int r = 1; r < ns; ++r) m = min(m, slices[(size_t)r * S + i]);
        slices[i] = m;
    }
}

template <bool HQS>
__global__ void __launch_bounds__(256) k_resolve(int show_num_points, int colorize_chunks, int width, int height,
                                                 const uint64_t *fb, const uint64_t *rg, const uint64_t *ba,
                                                 uint32_t *rgba)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15);     // 16x16 tiles as the reference launches
    const int y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= width || y >= height) return;
    const int pix = x + y * width;
    const uint32_t id = (uint32_t)fb[pix];
    uint32_t color = PCR_BACKGROUND_COLOR;
    if (id < 0xFFFFFFFFu) {
        if (show_num_points) {
            const double div = HQS ? 512.0 : 64.0;
            const uint32_t shade = (uint32_t)(((double)(float)(int)id / div) * 255.0);
            color = (shade << 24) | (shade << 16) | (shade << 8) | shade;
        } else if (colorize_chunks) {
            color = id * 1234567u;
        } else if (HQS) {
            const uint64_t vrg = rg[pix], vba = ba[pix];
            const uint32_t cnt = (uint32_t)vba;
            if (cnt == 0) color = 0;
            else color = (((uint32_t)(vba >> 32) / cnt) << 16) | (((uint32_t)vrg / cnt) << 8) | ((uint32_t)(vrg >> 32) / cnt);
        } else {
            color = id;
        }
    }
    rgba[pix] = color;
}

// RESOLVE of the finished frame, CLEAR for the next one and the next frame's cull/LOD prepass in ONE launch (pcr_frame_turn):
// the reference's frame ends with resolve + clear (huffman_hqs.h:240-270); done separately they are two passes over the
// framebuffer and two launches (21 us of a 269 us frame at 1080p), fused one pass reads every word once, writes the pixel,
// and writes the empty word back. The first `prepass_blocks` workgroups do the prepass instead (it touches no framebuffer).
// Same arithmetic as k_resolve<HQS>; pixels are walked linearly (the resolve is per pixel, the tile shape does not matter).
template <bool HQS>
__global__ void __launch_bounds__(256) k_frame_turn(RenderArgs a, uint32_t prepass_blocks, int show_num_points, int colorize_chunks,
                                                    uint32_t pixels, uint64_t *fb, uint64_t *rg, uint64_t *ba, uint32_t *rgba,
                                                    uint32_t n, uint64_t empty, TileFlags tiles)
{
    if (blockIdx.x < prepass_blocks) { lod_prepass_block(a, blockIdx.x); return; }
    // (the whole frame is resolved and cleared: from here on the tile flags can be kept again -- nothing written yet, any image
    // pixel may hold something)
    if (blockIdx.x == gridDim.x - 1) { TileFlags t = tiles; t.tracked = 0; clear_tile_flags(t); }
    const uint32_t stride = (gridDim.x - prepass_blocks) * 256u;
    for (uint32_t i = (blockIdx.x - prepass_blocks) * 256u + threadIdx.x; i < n; i += stride) {
        if (i < pixels) {
            const uint32_t id = (uint32_t)fb[i];
            uint32_t color = PCR_BACKGROUND_COLOR;
            if (id < 0xFFFFFFFFu) {
                if (show_num_points) {
                    const double div = HQS ? 512.0 : 64.0;
                    const uint32_t shade = (uint32_t)(((double)(float)(int)id / div) * 255.0);
                    color = (shade << 24) | (shade << 16) | (shade << 8) | shade;
                } else if (colorize_chunks) {
                    color = id * 1234567u;
                } else if (HQS) {
                    const uint64_t vrg = rg[i], vba = ba[i];
                    const uint32_t cnt = (uint32_t)vba;
                    if (cnt == 0) color = 0;
                    else color = (((uint32_t)(vba >> 32) / cnt) << 16) | (((uint32_t)vrg / cnt) << 8) | ((uint32_t)(vrg >> 32) / cnt);
                } else {
                    color = id;
                }
            }
            rgba[i] = color;
        }
        fb[i] = empty;
        if (HQS) { rg[i] = 0; ba[i] = 0; }
    }
}

// The same over the dirty tiles only (FrameView::tiles). `cur`: tiles a framebuffer word was written in during the finished frame
// -> resolved and cleared; `prev`: tiles whose image pixels may still hold something of the frame before -> background colour
// again, unless they are dirty now. A workgroup takes whole tiles (256 threads x 4 pixels), reads its tile's two flags, and
// leaves `prev` zeroed. Three arrays rotate: this launch's own prepass blocks mark the tiles of the NEXT frame in a third one
// (all zero since the turn before), `cur` is what the next turn reads as its `prev`, the zeroed `prev` is marked by the frame after.
template <bool HQS>
__global__ void __launch_bounds__(256) k_frame_turn_tiles(RenderArgs a, uint32_t prepass_blocks, int show_num_points, int colorize_chunks,
                                                          uint32_t width, uint32_t pixels, uint64_t *fb, uint64_t *rg, uint64_t *ba, uint32_t *rgba,
                                                          uint32_t n, uint64_t empty, TileFlags tf, uint32_t tiles_x)
{
    if (blockIdx.x < prepass_blocks) { lod_prepass_block(a, blockIdx.x); return; }
    const uint8_t *cur = tf.cur;
    uint8_t *prev = tf.prev;
    const uint32_t ntiles = tf.ntiles;
    const uint32_t row = threadIdx.x >> 4, col = (threadIdx.x & 15u) * 4u;
    const bool all_dirty = *tf.cur_all == tf.e_cur, all_was = *tf.prev_all == tf.e_prev;
    if (all_dirty) {
        // everything: the frame as k_frame_turn walks it (linear, 8-byte loads side by side), and `prev` zeroed on the way
        const uint32_t stride = (gridDim.x - prepass_blocks) * 256u, first = (blockIdx.x - prepass_blocks) * 256u + threadIdx.x;
        for (uint32_t t = first; t < ntiles; t += stride) prev[t] = 0;
        for (uint32_t i = first; i < n; i += stride) {
            if (i < pixels) {
                const uint32_t id = (uint32_t)fb[i];
                uint32_t color = PCR_BACKGROUND_COLOR;
                if (id < 0xFFFFFFFFu) {
                    if (show_num_points) {
                        const double div = HQS ? 512.0 : 64.0;
                        const uint32_t shade = (uint32_t)(((double)(float)(int)id / div) * 255.0);
                        color = (shade << 24) | (shade << 16) | (shade << 8) | shade;
                    } else if (colorize_chunks) {
                        color = id * 1234567u;
                    } else if (HQS) {
                        const uint64_t vrg = rg[i], vba = ba[i];
                        const uint32_t cnt = (uint32_t)vba;
                        if (cnt == 0) color = 0;
                        else color = (((uint32_t)(vba >> 32) / cnt) << 16) | (((uint32_t)vrg / cnt) << 8) | ((uint32_t)(vrg >> 32) / cnt);
                    } else {
                        color = id;
                    }
                }
                rgba[i] = color;
            }
            fb[i] = empty;
            if (HQS) { rg[i] = 0; ba[i] = 0; }
        }
        return;
    }
    for (uint32_t t = blockIdx.x - prepass_blocks; t < ntiles; t += gridDim.x - prepass_blocks) {       // (uniform)
        const bool dirty = cur[t] != 0, was = all_was || prev[t] != 0;
        if (!dirty && !was) continue;
        const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
        const uint32_t x = (tx << TILE_W_SHIFT) + col, y = (ty << TILE_H_SHIFT) + row;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t i = y * width + x + k;
            if (x + k >= width || i >= n) continue;
            if (!dirty) { if (i < pixels) rgba[i] = PCR_BACKGROUND_COLOR; continue; }
            if (i < pixels) {
                const uint32_t id = (uint32_t)fb[i];
                uint32_t color = PCR_BACKGROUND_COLOR;
                if (id < 0xFFFFFFFFu) {
                    if (show_num_points) {
                        const double div = HQS ? 512.0 : 64.0;
                        const uint32_t shade = (uint32_t)(((double)(float)(int)id / div) * 255.0);
                        color = (shade << 24) | (shade << 16) | (shade << 8) | shade;
                    } else if (colorize_chunks) {
                        color = id * 1234567u;
                    } else if (HQS) {
                        const uint64_t vrg = rg[i], vba = ba[i];
                        const uint32_t cnt = (uint32_t)vba;
                        if (cnt == 0) color = 0;
                        else color = (((uint32_t)(vba >> 32) / cnt) << 16) | (((uint32_t)vrg / cnt) << 8) | ((uint32_t)(vrg >> 32) / cnt);
                    } else {
                        color = id;
                    }
                }
                rgba[i] = color;
            }
            fb[i] = empty;
            if (HQS) { rg[i] = 0; ba[i] = 0; }
        }
        if (threadIdx.x == 0 && was) prev[t] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// multi-GPU merges
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_merge_min(uint64_t *dst, const uint64_t *src, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] = min(dst[i], src[i]);
}
__global__ void __launch_bounds__(256) k_merge_sum(uint64_t *dst, const uint64_t *src, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] += src[i];
}
__global__ void __launch_bounds__(256) k_flip_sign(uint64_t *dst, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] ^= 0x8000000000000000ull;
}

} // namespace pcr
