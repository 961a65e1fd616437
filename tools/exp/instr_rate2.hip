// Issue cost of gfx950 vector instructions, second take (round 2; answers VERDICT r01 "settle the issue-rate question").
//
// instr_rate.hip chained every copy of the instruction through ONE destination register per wave. Here every wave
// runs REPS x 16 copies that write 16 DISTINCT registers (16 independent chains, each of dependency distance 16), at
// 1, 2, 4 and 8 waves per SIMD, on every CU of the chip at once. Reported per instruction and occupancy:
//   cyc  = shader cycles (s_memtime) one workgroup took / wave-instructions issued per SIMD by that workgroup's CU share
//   wall = chip-wide wall time x clock / wave-instructions per SIMD -- the same figure from hipEvents and the clock
//          measured in the same launch (s_memtime ticks per s_memrealtime tick x 100 MHz)
// If a wave64 VALU instruction occupied its SIMD for 2 cycles, `cyc` would read 2 at >= 2 waves per SIMD.
// Mixed streams (VALU + MFMA 4x4x1) tell whether a small MFMA takes a VALU issue slot or runs beside the VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REPS 1024

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long cyc, real; };

#define DECL16(T, n, init) T n##0 = init, n##1 = init, n##2 = init, n##3 = init, n##4 = init, n##5 = init, n##6 = init, n##7 = init, \
                             n##8 = init, n##9 = init, n##10 = init, n##11 = init, n##12 = init, n##13 = init, n##14 = init, n##15 = init
#define OUT16(n) "+v"(n##0), "+v"(n##1), "+v"(n##2), "+v"(n##3), "+v"(n##4), "+v"(n##5), "+v"(n##6), "+v"(n##7), \
                 "+v"(n##8), "+v"(n##9), "+v"(n##10), "+v"(n##11), "+v"(n##12), "+v"(n##13), "+v"(n##14), "+v"(n##15)
#define SUM16(n) (n##0 + n##1 + n##2 + n##3 + n##4 + n##5 + n##6 + n##7 + n##8 + n##9 + n##10 + n##11 + n##12 + n##13 + n##14 + n##15)

// one instruction template applied to 16 destination registers %0..%15; %16.. are inputs
#define REP16(P, S) P "%0" S P "%1" S P "%2" S P "%3" S P "%4" S P "%5" S P "%6" S P "%7" S \
                    P "%8" S P "%9" S P "%10" S P "%11" S P "%12" S P "%13" S P "%14" S P "%15" S

#define PROLOGUE()                                                      \
    __syncthreads();                                                    \
    const unsigned long long t0 = __builtin_readcyclecounter();         \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#define EPILOGUE(sumexpr)                                               \
    const unsigned long long t1 = __builtin_readcyclecounter();         \
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();     \
    __syncthreads();                                                    \
    if (threadIdx.x == 0) { out[blockIdx.x].cyc = t1 - t0; out[blockIdx.x].real = r1 - r0; } \
    if ((sumexpr) == 0x12345) sink[0] = 1;

// --- 32-bit integer destinations ---------------------------------------------------------------------------------
#define KERNEL_U32(NAME, ASM)                                                                      \
    __global__ void __launch_bounds__(1024) NAME(Stamp *out, unsigned *sink)                       \
    {                                                                                              \
        DECL16(unsigned, v, threadIdx.x);                                                          \
        unsigned a = threadIdx.x * 3 + 1, b = 7;                                                   \
        PROLOGUE()                                                                                 \
        for (int r = 0; r < REPS; ++r) asm volatile(ASM : OUT16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21", "s22", "s23");      \
        EPILOGUE(SUM16(v))                                                                         \
    }
KERNEL_U32(k_add_u32,     REP16("v_add_u32 ", ", %16, %17\n"))                       // dst = a + b (no chain at all)
KERNEL_U32(k_add_u32_acc, "v_add_u32 %0, %0, %16\n v_add_u32 %1, %1, %16\n v_add_u32 %2, %2, %16\n v_add_u32 %3, %3, %16\n"
                          "v_add_u32 %4, %4, %16\n v_add_u32 %5, %5, %16\n v_add_u32 %6, %6, %16\n v_add_u32 %7, %7, %16\n"
                          "v_add_u32 %8, %8, %16\n v_add_u32 %9, %9, %16\n v_add_u32 %10, %10, %16\n v_add_u32 %11, %11, %16\n"
                          "v_add_u32 %12, %12, %16\n v_add_u32 %13, %13, %16\n v_add_u32 %14, %14, %16\n v_add_u32 %15, %15, %16\n")
KERNEL_U32(k_and_b32,     REP16("v_and_b32 ", ", %16, %17\n"))
KERNEL_U32(k_lshrrev_b32, REP16("v_lshrrev_b32 ", ", %17, %16\n"))
KERNEL_U32(k_bfe_u32,     REP16("v_bfe_u32 ", ", %16, %17, %17\n"))
KERNEL_U32(k_perm_b32,    REP16("v_perm_b32 ", ", %16, %17, %16\n"))
KERNEL_U32(k_mad_u32_u24, REP16("v_mad_u32_u24 ", ", %16, %17, %16\n"))
KERNEL_U32(k_mul_lo_u32,  REP16("v_mul_lo_u32 ", ", %16, %17\n"))
KERNEL_U32(k_add3_u32,    REP16("v_add3_u32 ", ", %16, %17, %16\n"))
KERNEL_U32(k_cndmask,     REP16("v_cndmask_b32 ", ", %16, %17, vcc\n"))
KERNEL_U32(k_sub_sdwa,    REP16("v_sub_u32_sdwa ", ", %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"))
KERNEL_U32(k_add_sdwa_w1, REP16("v_add_u32_sdwa ", ", %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"))
KERNEL_U32(k_cvt_i32_f32, REP16("v_cvt_i32_f32 ", ", %16\n"))
KERNEL_U32(k_cmp_lt_u32,  "v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n"
                          "v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n"
                          "v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n"
                          "v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n v_cmp_lt_u32 vcc, %16, %17\n")


KERNEL_U32(k_sub_u32,     REP16("v_sub_u32 ", ", %16, %17\n"))
KERNEL_U32(k_or_b32,      REP16("v_or_b32 ", ", %16, %17\n"))
KERNEL_U32(k_xor_b32,     REP16("v_xor_b32 ", ", %16, %17\n"))
KERNEL_U32(k_lshlrev_b32, REP16("v_lshlrev_b32 ", ", %17, %16\n"))
KERNEL_U32(k_ashrrev_i32, REP16("v_ashrrev_i32 ", ", %17, %16\n"))
KERNEL_U32(k_min_u32,     REP16("v_min_u32 ", ", %16, %17\n"))
KERNEL_U32(k_max_i32,     REP16("v_max_i32 ", ", %16, %17\n"))
KERNEL_U32(k_mov_b32,     REP16("v_mov_b32 ", ", %16\n"))
KERNEL_U32(k_mul_u32_u24, REP16("v_mul_u32_u24 ", ", %16, %17\n"))
KERNEL_U32(k_lshl_add_u32,REP16("v_lshl_add_u32 ", ", %16, 2, %17\n"))
KERNEL_U32(k_add_lshl_u32,REP16("v_add_lshl_u32 ", ", %16, %17, 2\n"))
KERNEL_U32(k_lshl_or_b32, REP16("v_lshl_or_b32 ", ", %16, 2, %17\n"))
KERNEL_U32(k_and_or_b32,  REP16("v_and_or_b32 ", ", %16, %17, %16\n"))
KERNEL_U32(k_or3_b32,     REP16("v_or3_b32 ", ", %16, %17, %16\n"))
KERNEL_U32(k_alignbit,    REP16("v_alignbit_b32 ", ", %16, %17, %17\n"))
KERNEL_U32(k_bfi_b32,     REP16("v_bfi_b32 ", ", %16, %17, %16\n"))
KERNEL_U32(k_med3_u32,    REP16("v_med3_u32 ", ", %16, %17, %16\n"))
KERNEL_U32(k_min3_u32,    REP16("v_min3_u32 ", ", %16, %17, %16\n"))
KERNEL_U32(k_mad_i32_i24, REP16("v_mad_i32_i24 ", ", %16, %17, %16\n"))
KERNEL_U32(k_bcnt,        REP16("v_bcnt_u32_b32 ", ", %16, %17\n"))
KERNEL_U32(k_mbcnt_lo,    REP16("v_mbcnt_lo_u32_b32 ", ", %16, %17\n"))
KERNEL_U32(k_add_co,      REP16("v_add_co_u32 ", ", vcc, %16, %17\n"))
KERNEL_U32(k_mov_dpp,     REP16("v_mov_b32_dpp ", ", %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"))
KERNEL_U32(k_add_dpp,     REP16("v_add_u32_dpp ", ", %16, %17 row_shr:1 row_mask:0xf bank_mask:0xf\n"))
KERNEL_U32(k_cvt_f32_u32_,REP16("v_cvt_f32_u32 ", ", %16\n"))
KERNEL_U32(k_cvt_u32_f32, REP16("v_cvt_u32_f32 ", ", %16\n"))
KERNEL_U32(k_cmp_e64,     "v_cmp_lt_u32_e64 s[20:21], %16, %17\n v_cmp_lt_u32_e64 s[22:23], %16, %17\n v_cmp_lt_u32_e64 s[20:21], %16, %17\n v_cmp_lt_u32_e64 s[22:23], %16, %17\n"
                          "v_cmp_lt_u32_e64 s[20:21], %16, %17\n v_cmp_lt_u32_e64 s[22:23], %16, %17\n v_cmp_lt_u32_e64 s[20:21], %16, %17\n v_cmp_lt_u32_e64 s[22:23], %16, %17\n"
                          "v_cmp_lt_u32_e64 s[20:21], %16, %17\n v_cmp_lt_u32_e64 s[22:23], %16, %17\n v_cmp_lt_u32_e64 s[20:21], %16, %17\n v_cmp_lt_u32_e64 s[22:23], %16, %17\n"
                          "v_cmp_lt_u32_e64 s[20:21], %16, %17\n v_cmp_lt_u32_e64 s[22:23], %16, %17\n v_cmp_lt_u32_e64 s[20:21], %16, %17\n v_cmp_lt_u32_e64 s[22:23], %16, %17\n")
KERNEL_U32(k_cndmask_vccset, "s_mov_b64 vcc, 0x5555\n" REP16("v_cndmask_b32 ", ", %16, %17, vcc\n"))
KERNEL_U32(k_cndmask_e64, "s_mov_b64 s[20:21], 0x5555\n" REP16("v_cndmask_b32_e64 ", ", %16, %17, s[20:21]\n"))
KERNEL_U32(k_cndmask_acc, "s_mov_b64 vcc, 0x5555\n"
                          "v_cndmask_b32 %0, %0, %16, vcc\n v_cndmask_b32 %1, %1, %16, vcc\n v_cndmask_b32 %2, %2, %16, vcc\n v_cndmask_b32 %3, %3, %16, vcc\n"
                          "v_cndmask_b32 %4, %4, %16, vcc\n v_cndmask_b32 %5, %5, %16, vcc\n v_cndmask_b32 %6, %6, %16, vcc\n v_cndmask_b32 %7, %7, %16, vcc\n"
                          "v_cndmask_b32 %8, %8, %16, vcc\n v_cndmask_b32 %9, %9, %16, vcc\n v_cndmask_b32 %10, %10, %16, vcc\n v_cndmask_b32 %11, %11, %16, vcc\n"
                          "v_cndmask_b32 %12, %12, %16, vcc\n v_cndmask_b32 %13, %13, %16, vcc\n v_cndmask_b32 %14, %14, %16, vcc\n v_cndmask_b32 %15, %15, %16, vcc\n")
KERNEL_U32(k_cmp_cnd_pairs, "v_cmp_lt_u32 vcc, %16, %17\n v_cndmask_b32 %0, %16, %17, vcc\n v_cmp_lt_u32 vcc, %17, %16\n v_cndmask_b32 %1, %16, %17, vcc\n"
                          "v_cmp_lt_u32 vcc, %16, %17\n v_cndmask_b32 %2, %16, %17, vcc\n v_cmp_lt_u32 vcc, %17, %16\n v_cndmask_b32 %3, %16, %17, vcc\n"
                          "v_cmp_lt_u32 vcc, %16, %17\n v_cndmask_b32 %4, %16, %17, vcc\n v_cmp_lt_u32 vcc, %17, %16\n v_cndmask_b32 %5, %16, %17, vcc\n"
                          "v_cmp_lt_u32 vcc, %16, %17\n v_cndmask_b32 %6, %16, %17, vcc\n v_cmp_lt_u32 vcc, %17, %16\n v_cndmask_b32 %7, %16, %17, vcc\n")

#define CMPV "v_cmp_lt_u32 vcc, %16, %17\n"
KERNEL_U32(k_cmp_cnd_cnd,     CMPV "v_cndmask_b32 %0, %16, %17, vcc\n v_cndmask_b32 %1, %16, %17, vcc\n v_add_u32 %2, %16, %17\n"
                              CMPV "v_cndmask_b32 %3, %16, %17, vcc\n v_cndmask_b32 %4, %16, %17, vcc\n v_add_u32 %5, %16, %17\n"
                              CMPV "v_cndmask_b32 %6, %16, %17, vcc\n v_cndmask_b32 %7, %16, %17, vcc\n v_add_u32 %8, %16, %17\n"
                              CMPV "v_cndmask_b32 %9, %16, %17, vcc\n v_cndmask_b32 %10, %16, %17, vcc\n v_add_u32 %11, %16, %17\n")
KERNEL_U32(k_cmp_cnd_add_add, CMPV "v_cndmask_b32 %0, %16, %17, vcc\n v_add_u32 %1, %16, %17\n v_add_u32 %2, %16, %17\n"
                              CMPV "v_cndmask_b32 %3, %16, %17, vcc\n v_add_u32 %4, %16, %17\n v_add_u32 %5, %16, %17\n"
                              CMPV "v_cndmask_b32 %6, %16, %17, vcc\n v_add_u32 %7, %16, %17\n v_add_u32 %8, %16, %17\n"
                              CMPV "v_cndmask_b32 %9, %16, %17, vcc\n v_add_u32 %10, %16, %17\n v_add_u32 %11, %16, %17\n")
KERNEL_U32(k_cmp_add_cnd_add, CMPV "v_add_u32 %0, %16, %17\n v_cndmask_b32 %1, %16, %17, vcc\n v_add_u32 %2, %16, %17\n"
                              CMPV "v_add_u32 %3, %16, %17\n v_cndmask_b32 %4, %16, %17, vcc\n v_add_u32 %5, %16, %17\n"
                              CMPV "v_add_u32 %6, %16, %17\n v_cndmask_b32 %7, %16, %17, vcc\n v_add_u32 %8, %16, %17\n"
                              CMPV "v_add_u32 %9, %16, %17\n v_cndmask_b32 %10, %16, %17, vcc\n v_add_u32 %11, %16, %17\n")
KERNEL_U32(k_cmp_cnd3_nop,    CMPV "v_cndmask_b32 %0, %16, %17, vcc\n s_nop 0\n v_cndmask_b32 %1, %16, %17, vcc\n s_nop 1\n v_cndmask_b32 %2, %16, %17, vcc\n"
                              CMPV "v_cndmask_b32 %3, %16, %17, vcc\n s_nop 0\n v_cndmask_b32 %4, %16, %17, vcc\n s_nop 1\n v_cndmask_b32 %5, %16, %17, vcc\n"
                              CMPV "v_cndmask_b32 %6, %16, %17, vcc\n s_nop 0\n v_cndmask_b32 %7, %16, %17, vcc\n s_nop 1\n v_cndmask_b32 %8, %16, %17, vcc\n"
                              CMPV "v_cndmask_b32 %9, %16, %17, vcc\n s_nop 0\n v_cndmask_b32 %10, %16, %17, vcc\n s_nop 1\n v_cndmask_b32 %11, %16, %17, vcc\n")
KERNEL_U32(k_add_literal,  REP16("v_add_u32 ", ", 0xa0800000, %17\n"))
KERNEL_U32(k_and_literal,  REP16("v_and_b32 ", ", 0x3ffc, %17\n"))
KERNEL_U32(k_lshr_const,   REP16("v_lshrrev_b32 ", ", 10, %17\n"))
KERNEL_U32(k_ashr_const,   REP16("v_ashrrev_i32 ", ", 10, %17\n"))
KERNEL_U32(k_add_sgpr,     REP16("v_add_u32 ", ", s20, %17\n"))
KERNEL_U32(k_sub_sgpr,     REP16("v_subrev_u32 ", ", s20, %17\n"))
KERNEL_U32(k_add_inline,   REP16("v_add_u32 ", ", 1, %17\n"))
KERNEL_U32(k_cmp_abs_f32,  "v_cmp_le_f32_e64 s[20:21], |%16|, %17\n v_cmp_le_f32_e64 s[22:23], |%16|, %17\n v_cmp_le_f32_e64 s[20:21], |%16|, %17\n v_cmp_le_f32_e64 s[22:23], |%16|, %17\n"
                           "v_cmp_le_f32_e64 s[20:21], |%16|, %17\n v_cmp_le_f32_e64 s[22:23], |%16|, %17\n v_cmp_le_f32_e64 s[20:21], |%16|, %17\n v_cmp_le_f32_e64 s[22:23], |%16|, %17\n"
                           "v_cmp_le_f32_e64 s[20:21], |%16|, %17\n v_cmp_le_f32_e64 s[22:23], |%16|, %17\n v_cmp_le_f32_e64 s[20:21], |%16|, %17\n v_cmp_le_f32_e64 s[22:23], |%16|, %17\n"
                           "v_cmp_le_f32_e64 s[20:21], |%16|, %17\n v_cmp_le_f32_e64 s[22:23], |%16|, %17\n v_cmp_le_f32_e64 s[20:21], |%16|, %17\n v_cmp_le_f32_e64 s[22:23], |%16|, %17\n")

// --- f32 destinations --------------------------------------------------------------------------------------------
#define KERNEL_F32(NAME, ASM)                                                                      \
    __global__ void __launch_bounds__(1024) NAME(Stamp *out, unsigned *sink)                       \
    {                                                                                              \
        DECL16(float, v, 1.0f + threadIdx.x);                                                      \
        float a = 1.0f + threadIdx.x * 0.001f, b = 0.5f; unsigned c = threadIdx.x;                 \
        PROLOGUE()                                                                                 \
        for (int r = 0; r < REPS; ++r) asm volatile(ASM : OUT16(v) : "v"(a), "v"(b), "v"(c) : "s20");      \
        EPILOGUE((unsigned)SUM16(v))                                                               \
    }
KERNEL_F32(k_fma_f32,     REP16("v_fma_f32 ", ", %16, %17, %16\n"))
KERNEL_F32(k_mul_f32,     REP16("v_mul_f32 ", ", %16, %17\n"))
KERNEL_F32(k_rcp_f32,     REP16("v_rcp_f32 ", ", %16\n"))
KERNEL_F32(k_cvt_f32_i32, REP16("v_cvt_f32_i32 ", ", %18\n"))


KERNEL_F32(k_add_f32,     REP16("v_add_f32 ", ", %16, %17\n"))
KERNEL_F32(k_fmac_f32,    "v_fmac_f32 %0, %16, %17\n v_fmac_f32 %1, %16, %17\n v_fmac_f32 %2, %16, %17\n v_fmac_f32 %3, %16, %17\n"
                          "v_fmac_f32 %4, %16, %17\n v_fmac_f32 %5, %16, %17\n v_fmac_f32 %6, %16, %17\n v_fmac_f32 %7, %16, %17\n"
                          "v_fmac_f32 %8, %16, %17\n v_fmac_f32 %9, %16, %17\n v_fmac_f32 %10, %16, %17\n v_fmac_f32 %11, %16, %17\n"
                          "v_fmac_f32 %12, %16, %17\n v_fmac_f32 %13, %16, %17\n v_fmac_f32 %14, %16, %17\n v_fmac_f32 %15, %16, %17\n")
KERNEL_F32(k_max_f32,     REP16("v_max_f32 ", ", %16, %17\n"))
KERNEL_F32(k_fma_f32_neg, REP16("v_fma_f32 ", ", -%16, %17, |%16|\n"))
KERNEL_F32(k_mul_f32_sgpr,REP16("v_mul_f32 ", ", s20, %17\n"))
KERNEL_F32(k_fma_f32_sgpr,REP16("v_fma_f32 ", ", s20, %17, %16\n"))

KERNEL_F32(k_mul_f32_inline, REP16("v_mul_f32 ", ", 0.5, %17\n"))
KERNEL_F32(k_fma_f32_inline, REP16("v_fma_f32 ", ", %16, 0.5, 0.5\n"))
KERNEL_F32(k_fract_f32,      REP16("v_fract_f32 ", ", %16\n"))
KERNEL_F32(k_floor_f32,      REP16("v_floor_f32 ", ", %16\n"))
KERNEL_F32(k_sub_f32,        REP16("v_sub_f32 ", ", %16, %17\n"))
KERNEL_F32(k_mac_chain,      "v_mul_f32 %0, %16, %17\n v_fmac_f32 %0, %17, %16\n v_fmac_f32 %0, %16, %16\n v_add_f32 %0, %17, %0\n"
                             "v_mul_f32 %1, %16, %17\n v_fmac_f32 %1, %17, %16\n v_fmac_f32 %1, %16, %16\n v_add_f32 %1, %17, %1\n"
                             "v_mul_f32 %2, %16, %17\n v_fmac_f32 %2, %17, %16\n v_fmac_f32 %2, %16, %16\n v_add_f32 %2, %17, %2\n"
                             "v_mul_f32 %3, %16, %17\n v_fmac_f32 %3, %17, %16\n v_fmac_f32 %3, %16, %16\n v_add_f32 %3, %17, %3\n")

// --- 64-bit destinations -----------------------------------------------------------------------------------------
#define KERNEL_B64(NAME, T, INIT, ASM)                                                             \
    __global__ void __launch_bounds__(1024) NAME(Stamp *out, unsigned *sink)                       \
    {                                                                                              \
        DECL16(T, v, INIT);                                                                        \
        T a = INIT, b = INIT; unsigned c = 5;                                                      \
        PROLOGUE()                                                                                 \
        for (int r = 0; r < REPS; ++r) asm volatile(ASM : OUT16(v) : "v"(a), "v"(b), "v"(c) : "vcc"); \
        EPILOGUE(0)                                                                                \
        if (threadIdx.x == 12345) *(T *)sink = v0;                                                 \
        if (threadIdx.x == 12346) *(T *)sink = v1; if (threadIdx.x == 12347) *(T *)sink = v2; if (threadIdx.x == 12348) *(T *)sink = v3; \
        if (threadIdx.x == 12349) *(T *)sink = v4; if (threadIdx.x == 12350) *(T *)sink = v5; if (threadIdx.x == 12351) *(T *)sink = v6; \
        if (threadIdx.x == 12352) *(T *)sink = v7; if (threadIdx.x == 12353) *(T *)sink = v8; if (threadIdx.x == 12354) *(T *)sink = v9; \
        if (threadIdx.x == 12355) *(T *)sink = v10; if (threadIdx.x == 12356) *(T *)sink = v11; if (threadIdx.x == 12357) *(T *)sink = v12; \
        if (threadIdx.x == 12358) *(T *)sink = v13; if (threadIdx.x == 12359) *(T *)sink = v14; if (threadIdx.x == 12360) *(T *)sink = v15; \
    }
KERNEL_B64(k_lshrrev_b64, unsigned long long, threadIdx.x, REP16("v_lshrrev_b64 ", ", %18, %16\n"))
KERNEL_B64(k_mad_u64_u32, unsigned long long, threadIdx.x, REP16("v_mad_u64_u32 ", ", vcc, %18, %18, %16\n"))
KERNEL_B64(k_cmp_lt_u64,  unsigned long long, threadIdx.x, REP16("v_cmp_lt_u64 vcc, %16, %17 ; ", "\n"))
KERNEL_B64(k_pk_fma_f32,  v2f, (v2f{1.0f, 2.0f}), REP16("v_pk_fma_f32 ", ", %16, %17, %16\n"))
KERNEL_B64(k_pk_mul_f32,  v2f, (v2f{1.0f, 2.0f}), REP16("v_pk_mul_f32 ", ", %16, %17\n"))
KERNEL_B64(k_fma_f64,     double, 1.0 + threadIdx.x, REP16("v_fma_f64 ", ", %16, %17, %16\n"))


KERNEL_B64(k_lshlrev_b64, unsigned long long, threadIdx.x, REP16("v_lshlrev_b64 ", ", %18, %16\n"))
KERNEL_B64(k_mov_b64,     unsigned long long, threadIdx.x, REP16("v_mov_b64 ", ", %16\n"))
KERNEL_B64(k_lshl_add_u64,unsigned long long, threadIdx.x, REP16("v_lshl_add_u64 ", ", %16, 3, %17\n"))
KERNEL_B64(k_pk_add_f32,  v2f, (v2f{1.0f, 2.0f}), REP16("v_pk_add_f32 ", ", %16, %17\n"))
KERNEL_B64(k_pk_mov_b32,  v2f, (v2f{1.0f, 2.0f}), REP16("v_pk_mov_b32 ", ", %16, %17\n"))
KERNEL_B64(k_cvt_f64_i32, double, 1.0 + threadIdx.x, REP16("v_cvt_f64_i32 ", ", %18\n"))

// --- MFMA 4x4x1 (16 blocks): every lane supplies one B value and gets its own column of four results --------------
// Mixed streams: 16 slots per repetition; MIX_M of them are MFMAs (4 independent accumulators, each MFMA accumulating
// onto its own previous result), the others v_add_u32 on distinct registers.
#define MFMA(acc) "v_mfma_f32_4x4x1_16b_f32 " acc ", %20, %21, " acc "\n"
#define ADDS4(i0, i1, i2, i3) "v_add_u32 %" #i0 ", %22, %23\n v_add_u32 %" #i1 ", %22, %23\n v_add_u32 %" #i2 ", %22, %23\n v_add_u32 %" #i3 ", %22, %23\n"
#define KERNEL_MIX(NAME, ASM)                                                                      \
    __global__ void __launch_bounds__(1024) NAME(Stamp *out, unsigned *sink)                       \
    {                                                                                              \
        DECL16(unsigned, v, threadIdx.x);                                                          \
        v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;                                          \
        float a = 1.0f + threadIdx.x * 0.001f, b = 0.5f; unsigned ia = threadIdx.x, ib = 7;        \
        PROLOGUE()                                                                                 \
        for (int r = 0; r < REPS; ++r)                                                             \
            asm volatile(ASM : OUT16(v), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b), "v"(ia), "v"(ib)); \
        EPILOGUE(SUM16(v) + (unsigned)(c0.x + c1.y + c2.z + c3.w))                                 \
    }
KERNEL_MIX(k_mfma4x4_16,        MFMA("%16") MFMA("%17") MFMA("%18") MFMA("%19") MFMA("%16") MFMA("%17") MFMA("%18") MFMA("%19")
                                MFMA("%16") MFMA("%17") MFMA("%18") MFMA("%19") MFMA("%16") MFMA("%17") MFMA("%18") MFMA("%19"))
KERNEL_MIX(k_mfma4x4_chain16,   MFMA("%16") MFMA("%16") MFMA("%16") MFMA("%16") MFMA("%16") MFMA("%16") MFMA("%16") MFMA("%16")
                                MFMA("%16") MFMA("%16") MFMA("%16") MFMA("%16") MFMA("%16") MFMA("%16") MFMA("%16") MFMA("%16"))
KERNEL_MIX(k_add12,             ADDS4(0, 1, 2, 3) ADDS4(4, 5, 6, 7) ADDS4(8, 9, 10, 11))
KERNEL_MIX(k_add12_mfma4,       ADDS4(0, 1, 2, 3) MFMA("%16") ADDS4(4, 5, 6, 7) MFMA("%17") ADDS4(8, 9, 10, 11) MFMA("%18") MFMA("%19"))
KERNEL_MIX(k_add12_mfma4_chain, ADDS4(0, 1, 2, 3) MFMA("%16") ADDS4(4, 5, 6, 7) MFMA("%16") ADDS4(8, 9, 10, 11) MFMA("%16") MFMA("%16"))
KERNEL_MIX(k_add8_mfma8,        ADDS4(0, 1, 2, 3) MFMA("%16") MFMA("%17") MFMA("%18") MFMA("%19") ADDS4(4, 5, 6, 7) MFMA("%16") MFMA("%17") MFMA("%18") MFMA("%19"))

// --- LDS: ds_read_b32 of a 4096-entry table at (pseudo-)random keys, as the decode step does -----------------------
template <int MODE>   // 0: every lane its own random key, 1: lane-linear (conflict free), 2: skewed (half the lanes share a key)
__global__ void __launch_bounds__(1024) k_lds_read(Stamp *out, unsigned *sink)
{
    __shared__ unsigned tab[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = (i * 2654435761u) >> 8;
    unsigned x = threadIdx.x * 2654435761u + 12345u, acc = 0;
    PROLOGUE()
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            x = x * 1664525u + 1013904223u;
            unsigned key = MODE == 1 ? ((threadIdx.x + r + k) & 4095u) : (x >> 20);
            if (MODE == 2 && (threadIdx.x & 1)) key = 17;
            acc += tab[key];
        }
    }
    EPILOGUE(acc)
}

struct K { const char *name; void (*fn)(Stamp *, unsigned *); int wave_instr_per_rep; };

int main(int argc, char **argv)
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    Stamp *out; unsigned *sink;
    const int max_blocks = cus * 2;
    hipMalloc(&out, sizeof(Stamp) * max_blocks); hipMalloc(&sink, 64);
    std::vector<K> ks = {
#define E(n) {#n, n, 16}
        E(k_cmp_cnd_cnd), E(k_cmp_cnd_add_add), E(k_cmp_add_cnd_add), E(k_cmp_cnd3_nop), E(k_add_literal), E(k_and_literal), E(k_lshr_const), E(k_ashr_const), E(k_add_sgpr), E(k_sub_sgpr), E(k_add_inline), E(k_cmp_abs_f32), E(k_mul_f32_inline), E(k_fma_f32_inline), E(k_fract_f32), E(k_floor_f32), E(k_sub_f32), E(k_mac_chain),
        E(k_add_u32), E(k_add_u32_acc), E(k_sub_u32), E(k_or_b32), E(k_xor_b32), E(k_lshlrev_b32), E(k_ashrrev_i32), E(k_min_u32), E(k_max_i32), E(k_mov_b32), E(k_mul_u32_u24), E(k_lshl_add_u32), E(k_add_lshl_u32), E(k_lshl_or_b32), E(k_and_or_b32), E(k_or3_b32), E(k_alignbit), E(k_bfi_b32), E(k_med3_u32), E(k_min3_u32), E(k_mad_i32_i24), E(k_bcnt), E(k_mbcnt_lo), E(k_add_co), E(k_mov_dpp), E(k_add_dpp), E(k_cvt_f32_u32_), E(k_cvt_u32_f32), E(k_cmp_e64), E(k_cndmask_vccset), E(k_cndmask_e64), E(k_cndmask_acc), E(k_cmp_cnd_pairs), E(k_add_f32), E(k_fmac_f32), E(k_max_f32), E(k_fma_f32_neg), E(k_mul_f32_sgpr), E(k_fma_f32_sgpr), E(k_lshlrev_b64), E(k_mov_b64), E(k_lshl_add_u64), E(k_pk_add_f32), E(k_pk_mov_b32), E(k_cvt_f64_i32), E(k_and_b32), E(k_lshrrev_b32), E(k_bfe_u32), E(k_perm_b32), E(k_mad_u32_u24), E(k_mul_lo_u32),
        E(k_add3_u32), E(k_cndmask), E(k_sub_sdwa), E(k_add_sdwa_w1), E(k_cvt_i32_f32), E(k_cmp_lt_u32), E(k_fma_f32), E(k_mul_f32),
        E(k_rcp_f32), E(k_cvt_f32_i32), E(k_lshrrev_b64), E(k_mad_u64_u32), E(k_cmp_lt_u64), E(k_pk_fma_f32), E(k_pk_mul_f32), E(k_fma_f64),
        E(k_mfma4x4_16), E(k_mfma4x4_chain16), {"k_add12", k_add12, 12}, E(k_add12_mfma4), E(k_add12_mfma4_chain), E(k_add8_mfma8),
        {"k_lds_read<random>", k_lds_read<0>, 16}, {"k_lds_read<linear>", k_lds_read<1>, 16}, {"k_lds_read<skewed>", k_lds_read<2>, 16},
    };
    printf("%d CUs; REPS %d; columns: waves/SIMD -> cycles per wave-instruction per SIMD from s_memtime [from wall time x measured clock]\n", cus, REPS);
    printf("%-22s %22s %22s   clock GHz\n", "instruction", "1", "8");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<Stamp> h(max_blocks);
    for (auto &k : ks) {
        printf("%-22s", k.name);
        double ghz_last = 0;
        for (int wps : {1, 8}) {
            // one workgroup of 256*wps threads per CU (wps <= 4), or two of 1024 (wps = 8); every CU busy
            const int threads = wps <= 4 ? 256 * wps : 1024;
            const int blocks = wps <= 4 ? cus : cus * 2;
            double best_cyc = 1e30, best_wall = 1e30, ghz = 0;
            for (int it = 0; it < 4; ++it) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, out, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(h.data(), out, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost);
                std::vector<double> cyc(blocks), clk(blocks);
                for (int i = 0; i < blocks; ++i) { cyc[i] = (double)h[i].cyc; clk[i] = (double)h[i].cyc / (double)h[i].real * 0.1; }
                std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
                const double med_cyc = cyc[blocks / 2];
                ghz = clk[blocks / 2];
                // per SIMD: a workgroup puts threads/256 waves on each SIMD of its CU; with two workgroups per CU the
                // SIMD carries both, and both run for about the same span
                const double instr_per_simd = (double)REPS * k.wave_instr_per_rep * wps;
                best_cyc = std::min(best_cyc, med_cyc / instr_per_simd);
                best_wall = std::min(best_wall, (double)ms * 1e6 * ghz / instr_per_simd);   // ms -> ns, x GHz = cycles (includes launch overhead)
            }
            printf("     %7.2f [%7.2f]", best_cyc, best_wall);
            ghz_last = ghz;
        }
        printf("   %.2f\n", ghz_last);
        fflush(stdout);
    }
    return 0;
}
