// Issue cost of individual gfx950 VALU instructions, measured: 16 waves on one CU (4 per SIMD) each run
// REPS x 16 independent copies of the instruction; cycles per wave-instruction per SIMD = elapsed / (REPS*16*4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REPS 2048

#define KERNEL(NAME, ASM)                                                                                 \
    __global__ void __launch_bounds__(1024) NAME(unsigned long long *out, unsigned *sink)                 \
    {                                                                                                     \
        unsigned v0 = threadIdx.x, v1 = threadIdx.x * 3 + 1, v2 = 7, v3 = 9;                               \
        unsigned long long w0 = threadIdx.x, w1 = 5;                                                      \
        float f0 = 1.0f + threadIdx.x, f1 = 2.0f;                                                         \
        typedef float v2f __attribute__((ext_vector_type(2)));                                            \
        v2f p0 = {f0, f1}, p1 = {f1, f0};                                                                 \
        double d0 = f0, d1 = f1;                                                                          \
        __syncthreads();                                                                                  \
        unsigned long long t0 = __builtin_readcyclecounter();                                             \
        for (int r = 0; r < REPS; ++r) {                                                                  \
            asm volatile(ASM ASM ASM ASM ASM ASM ASM ASM ASM ASM ASM ASM ASM ASM ASM ASM                  \
                         : "+v"(v0), "+v"(v1), "+v"(w0), "+v"(f0), "+v"(p0), "+v"(d0)                     \
                         : "v"(v2), "v"(v3), "v"(w1), "v"(f1), "v"(p1), "v"(d1) : "vcc");                  \
        }                                                                                                 \
        unsigned long long t1 = __builtin_readcyclecounter();                                             \
        __syncthreads();                                                                                  \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                           \
        if (v0 + v1 + (unsigned)w0 + (unsigned)f0 + (unsigned)p0.x + (unsigned)d0 == 0x12345) sink[0] = 1;\
    }

// operands: %0 v0, %1 v1, %2 w0 (64-bit), %3 f0, %4 p0 (64-bit pair), %5 d0; inputs %6 v2, %7 v3, %8 w1, %9 f1, %10 p1, %11 d1
KERNEL(k_add_u32,        "v_add_u32 %0, %0, %6\n")
KERNEL(k_and_b32,        "v_and_b32 %0, %0, %6\n")
KERNEL(k_lshrrev_b32,    "v_lshrrev_b32 %0, %6, %0\n")
KERNEL(k_lshrrev_b64,    "v_lshrrev_b64 %2, %6, %2\n")
KERNEL(k_lshlrev_b64,    "v_lshlrev_b64 %2, %6, %2\n")
KERNEL(k_alignbit,       "v_alignbit_b32 %0, %0, %1, %6\n")
KERNEL(k_bfe_u32,        "v_bfe_u32 %0, %0, %6, %7\n")
KERNEL(k_bcnt,           "v_bcnt_u32_b32 %0, %1, %0\n")
KERNEL(k_mbcnt_lo,       "v_mbcnt_lo_u32_b32 %0, %1, %0\n")
KERNEL(k_mul_lo_u32,     "v_mul_lo_u32 %0, %0, %6\n")
KERNEL(k_mul_u32_u24,    "v_mul_u32_u24 %0, %0, %6\n")
KERNEL(k_mad_u32_u24,    "v_mad_u32_u24 %0, %0, %6, %7\n")
KERNEL(k_mad_u64_u32,    "v_mad_u64_u32 %2, vcc, %0, %6, %2\n")
KERNEL(k_lshl_add_u32,   "v_lshl_add_u32 %0, %0, 2, %6\n")
KERNEL(k_lshl_add_u64,   "v_lshl_add_u64 %2, %2, 3, %8\n")
KERNEL(k_sub_sdwa,       "v_sub_u32_sdwa %0, %0, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n")
KERNEL(k_cmp_lt_u64,     "v_cmp_lt_u64 vcc, %2, %8\n")
KERNEL(k_cmp_lt_u32,     "v_cmp_lt_u32 vcc, %0, %6\n")
KERNEL(k_cndmask,        "v_cndmask_b32 %0, %0, %6, vcc\n")
KERNEL(k_fma_f32,        "v_fma_f32 %3, %3, %9, %9\n")
KERNEL(k_pk_fma_f32,     "v_pk_fma_f32 %4, %4, %10, %10\n")
KERNEL(k_pk_mul_f32,     "v_pk_mul_f32 %4, %4, %10\n")
KERNEL(k_rcp_f32,        "v_rcp_f32 %3, %3\n")
KERNEL(k_cvt_f32_i32,    "v_cvt_f32_i32 %3, %0\n")
KERNEL(k_cvt_i32_f32,    "v_cvt_i32_f32 %0, %3\n")
KERNEL(k_cvt_f64_i32,    "v_cvt_f64_i32 %5, %0\n")
KERNEL(k_cvt_f32_f64,    "v_cvt_f32_f64 %3, %5\n")
KERNEL(k_fma_f64,        "v_fma_f64 %5, %5, %11, %11\n")
KERNEL(k_mov_b64,        "v_mov_b64 %2, %8\n")
KERNEL(k_perm,           "v_perm_b32 %0, %0, %1, %6\n")
KERNEL(k_and_or,         "v_and_or_b32 %0, %0, %6, %7\n")
KERNEL(k_cndmask_ind,    "v_cndmask_b32 %0, %1, %6, vcc\n v_cndmask_b32 %3, %1, %7, vcc\n")
KERNEL(k_add_ind,        "v_add_u32 %0, %1, %6\n v_add_u32 %3, %1, %7\n")
KERNEL(k_rcp_ind,        "v_rcp_f32 %3, %9\n v_rcp_f32 %0, %9\n")
KERNEL(k_cndmask_sgpr,   "v_cndmask_b32_e64 %0, %0, %6, s[20:21]\n")
KERNEL(k_cmp_cnd,        "v_cmp_lt_u32 vcc, %0, %6\n v_cndmask_b32 %0, %0, %6, vcc\n")
KERNEL(k_ds_read,        "ds_read_b32 %0, %1\n")
KERNEL(k_readlane_mix,   "v_add_u32 %0, %0, %6\n s_bcnt1_i32_b64 s20, vcc\n")

int main()
{
    unsigned long long *out; unsigned *sink;
    hipMalloc(&out, 8); hipMalloc(&sink, 4);
    struct K { const char *name; void (*fn)(unsigned long long *, unsigned *); };
    std::vector<K> ks = {
#define E(n) {#n, n}
        E(k_add_u32), E(k_and_b32), E(k_lshrrev_b32), E(k_lshrrev_b64), E(k_lshlrev_b64), E(k_alignbit), E(k_bfe_u32), E(k_bcnt),
        E(k_mbcnt_lo), E(k_mul_lo_u32), E(k_mul_u32_u24), E(k_mad_u32_u24), E(k_mad_u64_u32), E(k_lshl_add_u32), E(k_lshl_add_u64),
        E(k_sub_sdwa), E(k_cmp_lt_u64), E(k_cmp_lt_u32), E(k_cndmask), E(k_fma_f32), E(k_pk_fma_f32), E(k_pk_mul_f32), E(k_rcp_f32),
        E(k_cvt_f32_i32), E(k_cvt_i32_f32), E(k_cvt_f64_i32), E(k_cvt_f32_f64), E(k_fma_f64), E(k_mov_b64), E(k_perm), E(k_and_or),
        E(k_cndmask_ind), E(k_add_ind), E(k_rcp_ind), E(k_cndmask_sgpr), E(k_cmp_cnd), E(k_readlane_mix),
    };
    for (auto &k : ks) {
        unsigned long long best = ~0ull;
        float best_ms = 1e9f;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.fn, dim3(1), dim3(1024), 0, 0, out, sink);
            hipEventRecord(e1, 0);
            unsigned long long c = 0;
            hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (c < best) best = c;
            if (ms < best_ms) best_ms = ms;
        }
        printf("[%.1f us] ", best_ms * 1000.0f);
        // 16 waves = 4 per SIMD; each wave issues REPS*16 instructions
        printf("%-18s %7.2f cycles per wave-instruction per SIMD (elapsed %llu)\n", k.name, (double)best / (REPS * 16.0 * 4.0), best);
    }
    return 0;
}
