// tools/valu_rate.hip — calibration microbenchmark for gfx950.
//
// What it settles: how many cycles a SIMD needs per wave64 vector instruction of each CLASS the compositing kernels use
// (plain fp32, transcendental, DPP, v_permlane*_swap, compare / select, packed fp32, LDS broadcast reads), with a full,
// a half and an empty EXEC mask, alone and with 2 / 4 waves sharing the SIMD — and what one `global_atomic_add_f32`
// costs on the memory side (time; WRITE_SIZE / FETCH_SIZE under `rocprofv3 --pmc`, run by tools/valu_rate.sh).
//
// Method.  Every instruction class is one kernel instantiation `chain<OP, EXEC>`: REPS iterations of a loop whose body
// is 64 copies of the instruction on 8 rotating accumulators (so a copy never reads what the previous 7 wrote), in
// ONE inline-assembly statement, bracketed by s_memtime (shader cycles) and s_memrealtime (100 MHz) — the ratio of the two deltas is
// the clock the chain ran at.  Two launches per class:
//   * ONE workgroup of 256 * W threads (W = 1, 2, 4 waves per SIMD of one CU): cycles per instruction from the wave's
//     own s_memtime stamps.  "per wave" = (t1 - t0) / instructions of that wave; "per SIMD" = per wave / W.
//   * the whole chip (2048 workgroups of 256 threads = 8 waves per SIMD, one round): wall time by hipEvents -> wave
//     instructions per second per SIMD; this is also the launch rocprofv3 --pmc looks at (SQ_INSTS_VALU,
//     SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES per kernel name).
// Atomic patterns: see `atom` below.
//
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/valu_rate.hip -o tools/build/valu_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                               \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                        \
        }                                                                                   \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum Op {
    FMA = 0,      // v_fma_f32, 16 independent accumulators
    FMA_DEP,      // v_fma_f32, ONE accumulator: dependent chain
    MUL,          // v_mul_f32 (VOP2)
    ADD,          // v_add_f32 (VOP2)
    MAXF,         // v_max_f32
    EXP,          // v_exp_f32
    RCP,          // v_rcp_f32
    LOG,          // v_log_f32
    DPP_SHR,      // v_add_f32_dpp row_shr:1
    DPP_BCAST15,  // v_add_f32_dpp row_bcast:15
    DPP_QUAD,     // v_add_f32_dpp quad_perm:[1,0,3,2]
    PERM32,       // v_permlane32_swap
    PERM16,       // v_permlane16_swap
    CNDMASK,      // v_cndmask_b32 (vcc)
    CMP,          // v_cmp_lt_f32 -> vcc
    CMP_SGPR,     // v_cmp_lt_f32 -> SGPR pair (VOP3)
    PK_FMA,       // v_pk_fma_f32 (two fp32 FMAs per lane)
    PK_MUL,       // v_pk_mul_f32
    MOV,          // v_mov_b32
    READLANE,     // v_readlane_b32
    BPERMUTE,     // ds_bpermute_b32
    SWIZZLE,      // ds_swizzle_b32
    LDS_B128_BC,  // ds_read_b128, all lanes one address (broadcast)
    LDS_B32_BC,   // ds_read_b32, all lanes one address
    LDS_B128,     // ds_read_b128, lane l reads its own 16 bytes
    LDS_B32,      // ds_read_b32, lane l reads its own dword
    MIX_FMA_DPP,  // alternating v_fma_f32 / v_add_f32_dpp row_shr
    MIX_FMA_EXP,  // 3 v_fma_f32 : 1 v_exp_f32
    ATOM_ISSUE,   // global_atomic_add_f32 (no return), 15 lanes of every row -> one 64-B line per row: ISSUE cost
    G_SUB,
    G_FMAC,
    G_MIN,
    G_MUL_E64,
    G_ADD_NEG,
    G_FMA_SGPR,
    G_FMA_CONST,
    G_MUL_LIT,
    G_AND,
    G_OR,
    G_LSHL,
    G_ADDU,
    G_SUBU,
    G_ADDCO,
    G_CVT_F_I,
    G_CVT_I_F,
    G_MULLO,
    G_BFE,
    G_MED3,
    G_MAX3,
    G_LDEXP,
    G_SQRT,
    G_RSQ,
    G_MOV_DPP,
    G_MBCNT,
    G_CND_SGPR,
    G_CND_VCC_D,
    G_CND_E64_VCC,
    P_CMP_CND,
    P_CMPS_CNDS,
    P_FMA_MAX,
    P_FMA3_MAX,
    P_FMA_MUL_ADD,
    P_FMA_DPP_ALT,
    P_FMA7_EXP,
    P_FMA_SUB_MIN,
    P_FMA_LDS,
    P_FMAC_MAX,
    P_FMAC_FMA,
    P_FMAC_MUL_ADD,
    P_CMP_CND3,
    P_CMP_FMA_CND,
    P_FMA_MAX2,
    P_FMA_MAX_1_3,
    N_OPS
};
static const char *kOpName[N_OPS] = {"v_fma_f32",           "v_fma_f32 (dependent)", "v_mul_f32",          "v_add_f32",
                                     "v_max_f32",           "v_exp_f32",             "v_rcp_f32",          "v_log_f32",
                                     "v_add_f32_dpp row_shr:1", "v_add_f32_dpp row_bcast:15", "v_add_f32_dpp quad_perm",
                                     "v_permlane32_swap",   "v_permlane16_swap",     "v_cndmask_b32",      "v_cmp_lt_f32 vcc",
                                     "v_cmp_lt_f32 sgpr",   "v_pk_fma_f32",          "v_pk_mul_f32",       "v_mov_b32",
                                     "v_readlane_b32",      "ds_bpermute_b32",       "ds_swizzle_b32",     "ds_read_b128 broadcast",
                                     "ds_read_b32 broadcast", "ds_read_b128 per lane", "ds_read_b32 per lane",
                                     "mix fma : dpp 1:1",   "mix fma : exp 3:1",     "global_atomic_add_f32 (issue)",
    "v_sub_f32", "v_fmac_f32", "v_min_f32", "v_mul_f32 e64 |x|", "v_add_f32 e64 -x", "v_fma_f32 sgpr src", "v_fma_f32 inline 0.5", "v_mul_f32 literal", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_add_u32", "v_sub_u32", "v_add_co_u32 vcc", "v_cvt_f32_i32", "v_cvt_i32_f32", "v_mul_lo_u32", "v_bfe_u32", "v_med3_f32", "v_max3_f32", "v_ldexp_f32", "v_sqrt_f32", "v_rsq_f32", "v_mov_b32_dpp row_shr:1", "v_mbcnt_lo_u32_b32", "v_cndmask_b32 e64 sgpr mask", "v_cndmask_b32 vcc, dst != src", "v_cndmask_b32 e64 vcc", "cmp vcc ; cndmask vcc (1:1)", "cmp sgpr ; cndmask sgpr (1:1)", "fma ; max alternating", "3 fma ; 1 max", "fma ; mul ; add ; mov", "fma ; dpp alternating", "7 fma ; 1 exp", "fma ; sub ; min ; cmp", "7 fma ; 1 ds_read_b128 bcast", "fmac ; max alternating", "fmac ; fma alternating", "fmac ; mul ; add ; mov", "cmp vcc ; 3 cndmask vcc (VOP2)", "cmp vcc ; 4 fma ; cndmask vcc ; 2 fma", "2 fma ; 2 max", "1 fma ; 3 max"};

enum ExecMode { EX_FULL = 0, EX_LOW32, EX_EVEN, EX_LOW16, EX_ONE, EX_ZERO, EX_ROW15, N_EXEC };
static const char *kExecName[N_EXEC] = {"full", "lanes 0-31", "even lanes", "lanes 0-15", "one lane", "EMPTY", "15 of 16 per row"};
__host__ __device__ constexpr unsigned long long exec_mask(int m) {
    return m == EX_FULL    ? ~0ull
           : m == EX_LOW32 ? 0xffffffffull
           : m == EX_EVEN  ? 0x5555555555555555ull
           : m == EX_LOW16 ? 0xffffull
           : m == EX_ONE   ? 1ull
           : m == EX_ZERO  ? 0ull
                           : 0x7fff7fff7fff7fffull;
}

constexpr int kUnroll = 64;  // instructions per loop body

struct Stamp {
    unsigned long long cyc0, cyc1, real0, real1;
};

__device__ __forceinline__ unsigned long long memtime() {
    unsigned long long t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long memrealtime() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// The loop body of class OP: 64 instructions in ONE asm statement (separate statements make the compiler's hazard
// recogniser drop an s_nop between most of them).  %0..%7: eight rotating accumulators, so an instruction reads what the
// eighth instruction before it wrote (>= 16 issue cycles earlier: beyond every ALU latency); FMA_DEP uses %0 only.
#define FS_B8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define FS_B64(F) FS_B8(F) FS_B8(F) FS_B8(F) FS_B8(F) FS_B8(F) FS_B8(F) FS_B8(F) FS_B8(F)
#define FS_ACC8 "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define FS_PK8 "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7])
#define FS_Q8 "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3]), "=v"(q[4]), "=v"(q[5]), "=v"(q[6]), "=v"(q[7])
#define FS_OUT8 "=v"(a[0]), "=v"(a[1]), "=v"(a[2]), "=v"(a[3]), "=v"(a[4]), "=v"(a[5]), "=v"(a[6]), "=v"(a[7])

#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_FMA_DEP(n) "v_fma_f32 %0, %0, %8, %9\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_ADD(n) "v_add_f32 %" #n ", %" #n ", %9\n"
#define I_MAX(n) "v_max_f32 %" #n ", %" #n ", %9\n"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_LOG(n) "v_log_f32 %" #n ", %" #n "\n"
#define I_DPP_SHR(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_DPP_BC(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define I_DPP_QUAD(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define I_CMP_SGPR(n) "v_cmp_lt_f32 %8, %" #n ", %9\n"  /* %8 = the SGPR pair, %9 = b */
#define I_PK_FMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PK_MUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_READLANE(n) "v_readlane_b32 %8, %" #n ", 3\n"
#define I_BPERMUTE(n) "ds_bpermute_b32 %" #n ", %8, %9\n"
#define I_SWIZZLE(n) "ds_swizzle_b32 %" #n ", %8 offset:swizzle(SWAP,16)\n"
#define I_DSREAD128(n) "ds_read_b128 %" #n ", %8\n"
#define I_DSREAD32(n) "ds_read_b32 %" #n ", %8\n"
// v_permlane{32,16}_swap exchange halves / odd rows of two registers: both operands are read and written
#define I_P32A "v_permlane32_swap_b32 %0, %4\n v_permlane32_swap_b32 %1, %5\n v_permlane32_swap_b32 %2, %6\n v_permlane32_swap_b32 %3, %7\n"
#define I_P32B "v_permlane32_swap_b32 %4, %0\n v_permlane32_swap_b32 %5, %1\n v_permlane32_swap_b32 %6, %2\n v_permlane32_swap_b32 %7, %3\n"
#define I_P16A "v_permlane16_swap_b32 %0, %4\n v_permlane16_swap_b32 %1, %5\n v_permlane16_swap_b32 %2, %6\n v_permlane16_swap_b32 %3, %7\n"
#define I_P16B "v_permlane16_swap_b32 %4, %0\n v_permlane16_swap_b32 %5, %1\n v_permlane16_swap_b32 %6, %2\n v_permlane16_swap_b32 %7, %3\n"
#define FS_X8(A, B) A B A B A B A B A B A B A B A B
#define FS_X8R(A) A A A A A A A A
// lanes of row r add into line r of a group of four 64-byte lines; sixteen groups (immediate offsets) per wave
#define I_ATOM(n) "global_atomic_add_f32 %0, %1, %2 offset:" #n "*256\n global_atomic_add_f32 %0, %1, %2 offset:" #n "*256+2048\n"
#define FS_ATOM32 FS_B8(I_ATOM) FS_B8(I_ATOM)
#define I_MIX4 I_FMA(0) I_FMA(1) I_FMA(2) I_EXP(3) I_FMA(4) I_FMA(5) I_FMA(6) I_EXP(7)

template <int OP>
__device__ __forceinline__ void body(float (&a)[8], f2 (&p)[8], f4 (&q)[8], float b, float c, f2 pb, f2 pc, unsigned lds_addr_bc,
                                     unsigned lds_addr_lane, unsigned bperm_addr, unsigned long long &sg, unsigned &sg32, float *gp,
                                     unsigned goff, float sb, unsigned long long sm, unsigned long long &smw) {
    if constexpr (OP == FMA) asm volatile(FS_B64(I_FMA) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == FMA_DEP) asm volatile(FS_B64(I_FMA_DEP) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == MUL) asm volatile(FS_B64(I_MUL) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == ADD) asm volatile(FS_B64(I_ADD) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == MAXF) asm volatile(FS_B64(I_MAX) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == EXP) asm volatile(FS_B64(I_EXP) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == RCP) asm volatile(FS_B64(I_RCP) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == LOG) asm volatile(FS_B64(I_LOG) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == DPP_SHR) asm volatile(FS_B64(I_DPP_SHR) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == DPP_BCAST15) asm volatile(FS_B64(I_DPP_BC) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == DPP_QUAD) asm volatile(FS_B64(I_DPP_QUAD) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == PERM32) asm volatile(FS_X8(I_P32A, I_P32B) : FS_ACC8);
    if constexpr (OP == PERM16) asm volatile(FS_X8(I_P16A, I_P16B) : FS_ACC8);
    if constexpr (OP == CNDMASK) asm volatile(FS_B64(I_CNDMASK) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == CMP) asm volatile(FS_B64(I_CMP) : FS_ACC8 : "v"(b), "v"(c) : "vcc");
    if constexpr (OP == CMP_SGPR) asm volatile(FS_B64(I_CMP_SGPR) : FS_ACC8, "+s"(sg) : "v"(b), "v"(c));
    if constexpr (OP == PK_FMA) asm volatile(FS_B64(I_PK_FMA) : FS_PK8 : "v"(pb), "v"(pc));
    if constexpr (OP == PK_MUL) asm volatile(FS_B64(I_PK_MUL) : FS_PK8 : "v"(pb), "v"(pc));
    if constexpr (OP == MOV) asm volatile(FS_B64(I_MOV) : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == READLANE) asm volatile(FS_B64(I_READLANE) : FS_ACC8, "+s"(sg32) : "v"(b), "v"(c));
    if constexpr (OP == BPERMUTE) asm volatile(FS_B64(I_BPERMUTE) "s_waitcnt lgkmcnt(0)\n" : FS_ACC8 : "v"(bperm_addr), "v"(b));
    if constexpr (OP == SWIZZLE) asm volatile(FS_B64(I_SWIZZLE) "s_waitcnt lgkmcnt(0)\n" : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == LDS_B128_BC) asm volatile(FS_B64(I_DSREAD128) "s_waitcnt lgkmcnt(0)\n" : FS_Q8 : "v"(lds_addr_bc));
    if constexpr (OP == LDS_B32_BC) asm volatile(FS_B64(I_DSREAD32) "s_waitcnt lgkmcnt(0)\n" : FS_OUT8 : "v"(lds_addr_bc));
    if constexpr (OP == LDS_B128) asm volatile(FS_B64(I_DSREAD128) "s_waitcnt lgkmcnt(0)\n" : FS_Q8 : "v"(lds_addr_lane));
    if constexpr (OP == LDS_B32) asm volatile(FS_B64(I_DSREAD32) "s_waitcnt lgkmcnt(0)\n" : FS_OUT8 : "v"(lds_addr_lane));
    if constexpr (OP == MIX_FMA_DPP)
        asm volatile(FS_B8(I_FMA) FS_B8(I_DPP_SHR) FS_B8(I_FMA) FS_B8(I_DPP_SHR) FS_B8(I_FMA) FS_B8(I_DPP_SHR) FS_B8(I_FMA) FS_B8(I_DPP_SHR)
                     : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == MIX_FMA_EXP) asm volatile(I_MIX4 I_MIX4 I_MIX4 I_MIX4 I_MIX4 I_MIX4 I_MIX4 I_MIX4 : FS_ACC8 : "v"(b), "v"(c));
    if constexpr (OP == G_SUB) asm volatile(FS_X8R("v_sub_f32 %0, %0, %8\n" "v_sub_f32 %1, %1, %8\n" "v_sub_f32 %2, %2, %8\n" "v_sub_f32 %3, %3, %8\n" "v_sub_f32 %4, %4, %8\n" "v_sub_f32 %5, %5, %8\n" "v_sub_f32 %6, %6, %8\n" "v_sub_f32 %7, %7, %8\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_FMAC) asm volatile(FS_X8R("v_fmac_f32 %0, %8, %9\n" "v_fmac_f32 %1, %8, %9\n" "v_fmac_f32 %2, %8, %9\n" "v_fmac_f32 %3, %8, %9\n" "v_fmac_f32 %4, %8, %9\n" "v_fmac_f32 %5, %8, %9\n" "v_fmac_f32 %6, %8, %9\n" "v_fmac_f32 %7, %8, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_MIN) asm volatile(FS_X8R("v_min_f32 %0, %0, %9\n" "v_min_f32 %1, %1, %9\n" "v_min_f32 %2, %2, %9\n" "v_min_f32 %3, %3, %9\n" "v_min_f32 %4, %4, %9\n" "v_min_f32 %5, %5, %9\n" "v_min_f32 %6, %6, %9\n" "v_min_f32 %7, %7, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_MUL_E64) asm volatile(FS_X8R("v_mul_f32_e64 %0, |%0|, %8\n" "v_mul_f32_e64 %1, |%1|, %8\n" "v_mul_f32_e64 %2, |%2|, %8\n" "v_mul_f32_e64 %3, |%3|, %8\n" "v_mul_f32_e64 %4, |%4|, %8\n" "v_mul_f32_e64 %5, |%5|, %8\n" "v_mul_f32_e64 %6, |%6|, %8\n" "v_mul_f32_e64 %7, |%7|, %8\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_ADD_NEG) asm volatile(FS_X8R("v_add_f32_e64 %0, -%0, %9\n" "v_add_f32_e64 %1, -%1, %9\n" "v_add_f32_e64 %2, -%2, %9\n" "v_add_f32_e64 %3, -%3, %9\n" "v_add_f32_e64 %4, -%4, %9\n" "v_add_f32_e64 %5, -%5, %9\n" "v_add_f32_e64 %6, -%6, %9\n" "v_add_f32_e64 %7, -%7, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_FMA_SGPR) asm volatile(FS_X8R("v_fma_f32 %0, %0, %10, %9\n" "v_fma_f32 %1, %1, %10, %9\n" "v_fma_f32 %2, %2, %10, %9\n" "v_fma_f32 %3, %3, %10, %9\n" "v_fma_f32 %4, %4, %10, %9\n" "v_fma_f32 %5, %5, %10, %9\n" "v_fma_f32 %6, %6, %10, %9\n" "v_fma_f32 %7, %7, %10, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_FMA_CONST) asm volatile(FS_X8R("v_fma_f32 %0, %0, 0.5, %9\n" "v_fma_f32 %1, %1, 0.5, %9\n" "v_fma_f32 %2, %2, 0.5, %9\n" "v_fma_f32 %3, %3, 0.5, %9\n" "v_fma_f32 %4, %4, 0.5, %9\n" "v_fma_f32 %5, %5, 0.5, %9\n" "v_fma_f32 %6, %6, 0.5, %9\n" "v_fma_f32 %7, %7, 0.5, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_MUL_LIT) asm volatile(FS_X8R("v_mul_f32 %0, 0x3f8ccccd, %0\n" "v_mul_f32 %1, 0x3f8ccccd, %1\n" "v_mul_f32 %2, 0x3f8ccccd, %2\n" "v_mul_f32 %3, 0x3f8ccccd, %3\n" "v_mul_f32 %4, 0x3f8ccccd, %4\n" "v_mul_f32 %5, 0x3f8ccccd, %5\n" "v_mul_f32 %6, 0x3f8ccccd, %6\n" "v_mul_f32 %7, 0x3f8ccccd, %7\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_AND) asm volatile(FS_X8R("v_and_b32 %0, %0, %8\n" "v_and_b32 %1, %1, %8\n" "v_and_b32 %2, %2, %8\n" "v_and_b32 %3, %3, %8\n" "v_and_b32 %4, %4, %8\n" "v_and_b32 %5, %5, %8\n" "v_and_b32 %6, %6, %8\n" "v_and_b32 %7, %7, %8\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_OR) asm volatile(FS_X8R("v_or_b32 %0, %0, %9\n" "v_or_b32 %1, %1, %9\n" "v_or_b32 %2, %2, %9\n" "v_or_b32 %3, %3, %9\n" "v_or_b32 %4, %4, %9\n" "v_or_b32 %5, %5, %9\n" "v_or_b32 %6, %6, %9\n" "v_or_b32 %7, %7, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_LSHL) asm volatile(FS_X8R("v_lshlrev_b32 %0, 1, %0\n" "v_lshlrev_b32 %1, 1, %1\n" "v_lshlrev_b32 %2, 1, %2\n" "v_lshlrev_b32 %3, 1, %3\n" "v_lshlrev_b32 %4, 1, %4\n" "v_lshlrev_b32 %5, 1, %5\n" "v_lshlrev_b32 %6, 1, %6\n" "v_lshlrev_b32 %7, 1, %7\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_ADDU) asm volatile(FS_X8R("v_add_u32 %0, %0, %8\n" "v_add_u32 %1, %1, %8\n" "v_add_u32 %2, %2, %8\n" "v_add_u32 %3, %3, %8\n" "v_add_u32 %4, %4, %8\n" "v_add_u32 %5, %5, %8\n" "v_add_u32 %6, %6, %8\n" "v_add_u32 %7, %7, %8\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_SUBU) asm volatile(FS_X8R("v_sub_u32 %0, %0, %8\n" "v_sub_u32 %1, %1, %8\n" "v_sub_u32 %2, %2, %8\n" "v_sub_u32 %3, %3, %8\n" "v_sub_u32 %4, %4, %8\n" "v_sub_u32 %5, %5, %8\n" "v_sub_u32 %6, %6, %8\n" "v_sub_u32 %7, %7, %8\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_ADDCO) asm volatile(FS_X8R("v_add_co_u32 %0, vcc, %0, %8\n" "v_add_co_u32 %1, vcc, %1, %8\n" "v_add_co_u32 %2, vcc, %2, %8\n" "v_add_co_u32 %3, vcc, %3, %8\n" "v_add_co_u32 %4, vcc, %4, %8\n" "v_add_co_u32 %5, vcc, %5, %8\n" "v_add_co_u32 %6, vcc, %6, %8\n" "v_add_co_u32 %7, vcc, %7, %8\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm) : "vcc");
    if constexpr (OP == G_CVT_F_I) asm volatile(FS_X8R("v_cvt_f32_i32 %0, %0\n" "v_cvt_f32_i32 %1, %1\n" "v_cvt_f32_i32 %2, %2\n" "v_cvt_f32_i32 %3, %3\n" "v_cvt_f32_i32 %4, %4\n" "v_cvt_f32_i32 %5, %5\n" "v_cvt_f32_i32 %6, %6\n" "v_cvt_f32_i32 %7, %7\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_CVT_I_F) asm volatile(FS_X8R("v_cvt_i32_f32 %0, %0\n" "v_cvt_i32_f32 %1, %1\n" "v_cvt_i32_f32 %2, %2\n" "v_cvt_i32_f32 %3, %3\n" "v_cvt_i32_f32 %4, %4\n" "v_cvt_i32_f32 %5, %5\n" "v_cvt_i32_f32 %6, %6\n" "v_cvt_i32_f32 %7, %7\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_MULLO) asm volatile(FS_X8R("v_mul_lo_u32 %0, %0, %8\n" "v_mul_lo_u32 %1, %1, %8\n" "v_mul_lo_u32 %2, %2, %8\n" "v_mul_lo_u32 %3, %3, %8\n" "v_mul_lo_u32 %4, %4, %8\n" "v_mul_lo_u32 %5, %5, %8\n" "v_mul_lo_u32 %6, %6, %8\n" "v_mul_lo_u32 %7, %7, %8\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_BFE) asm volatile(FS_X8R("v_bfe_u32 %0, %0, 3, 8\n" "v_bfe_u32 %1, %1, 3, 8\n" "v_bfe_u32 %2, %2, 3, 8\n" "v_bfe_u32 %3, %3, 3, 8\n" "v_bfe_u32 %4, %4, 3, 8\n" "v_bfe_u32 %5, %5, 3, 8\n" "v_bfe_u32 %6, %6, 3, 8\n" "v_bfe_u32 %7, %7, 3, 8\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_MED3) asm volatile(FS_X8R("v_med3_f32 %0, %0, %8, %9\n" "v_med3_f32 %1, %1, %8, %9\n" "v_med3_f32 %2, %2, %8, %9\n" "v_med3_f32 %3, %3, %8, %9\n" "v_med3_f32 %4, %4, %8, %9\n" "v_med3_f32 %5, %5, %8, %9\n" "v_med3_f32 %6, %6, %8, %9\n" "v_med3_f32 %7, %7, %8, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_MAX3) asm volatile(FS_X8R("v_max3_f32 %0, %0, %8, %9\n" "v_max3_f32 %1, %1, %8, %9\n" "v_max3_f32 %2, %2, %8, %9\n" "v_max3_f32 %3, %3, %8, %9\n" "v_max3_f32 %4, %4, %8, %9\n" "v_max3_f32 %5, %5, %8, %9\n" "v_max3_f32 %6, %6, %8, %9\n" "v_max3_f32 %7, %7, %8, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_LDEXP) asm volatile(FS_X8R("v_ldexp_f32 %0, %0, 1\n" "v_ldexp_f32 %1, %1, 1\n" "v_ldexp_f32 %2, %2, 1\n" "v_ldexp_f32 %3, %3, 1\n" "v_ldexp_f32 %4, %4, 1\n" "v_ldexp_f32 %5, %5, 1\n" "v_ldexp_f32 %6, %6, 1\n" "v_ldexp_f32 %7, %7, 1\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_SQRT) asm volatile(FS_X8R("v_sqrt_f32 %0, %0\n" "v_sqrt_f32 %1, %1\n" "v_sqrt_f32 %2, %2\n" "v_sqrt_f32 %3, %3\n" "v_sqrt_f32 %4, %4\n" "v_sqrt_f32 %5, %5\n" "v_sqrt_f32 %6, %6\n" "v_sqrt_f32 %7, %7\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_RSQ) asm volatile(FS_X8R("v_rsq_f32 %0, %0\n" "v_rsq_f32 %1, %1\n" "v_rsq_f32 %2, %2\n" "v_rsq_f32 %3, %3\n" "v_rsq_f32 %4, %4\n" "v_rsq_f32 %5, %5\n" "v_rsq_f32 %6, %6\n" "v_rsq_f32 %7, %7\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_MOV_DPP) asm volatile(FS_X8R("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_MBCNT) asm volatile(FS_X8R("v_mbcnt_lo_u32_b32 %0, -1, %0\n" "v_mbcnt_lo_u32_b32 %1, -1, %1\n" "v_mbcnt_lo_u32_b32 %2, -1, %2\n" "v_mbcnt_lo_u32_b32 %3, -1, %3\n" "v_mbcnt_lo_u32_b32 %4, -1, %4\n" "v_mbcnt_lo_u32_b32 %5, -1, %5\n" "v_mbcnt_lo_u32_b32 %6, -1, %6\n" "v_mbcnt_lo_u32_b32 %7, -1, %7\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_CND_SGPR) asm volatile(FS_X8R("v_cndmask_b32_e64 %0, %0, %8, %11\n" "v_cndmask_b32_e64 %1, %1, %8, %11\n" "v_cndmask_b32_e64 %2, %2, %8, %11\n" "v_cndmask_b32_e64 %3, %3, %8, %11\n" "v_cndmask_b32_e64 %4, %4, %8, %11\n" "v_cndmask_b32_e64 %5, %5, %8, %11\n" "v_cndmask_b32_e64 %6, %6, %8, %11\n" "v_cndmask_b32_e64 %7, %7, %8, %11\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_CND_VCC_D) asm volatile(FS_X8R("v_cndmask_b32 %0, %8, %9, vcc\n" "v_cndmask_b32 %1, %8, %9, vcc\n" "v_cndmask_b32 %2, %8, %9, vcc\n" "v_cndmask_b32 %3, %8, %9, vcc\n" "v_cndmask_b32 %4, %8, %9, vcc\n" "v_cndmask_b32 %5, %8, %9, vcc\n" "v_cndmask_b32 %6, %8, %9, vcc\n" "v_cndmask_b32 %7, %8, %9, vcc\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == G_CND_E64_VCC) asm volatile(FS_X8R("v_cndmask_b32_e64 %0, %0, %8, vcc\n" "v_cndmask_b32_e64 %1, %1, %8, vcc\n" "v_cndmask_b32_e64 %2, %2, %8, vcc\n" "v_cndmask_b32_e64 %3, %3, %8, vcc\n" "v_cndmask_b32_e64 %4, %4, %8, vcc\n" "v_cndmask_b32_e64 %5, %5, %8, vcc\n" "v_cndmask_b32_e64 %6, %6, %8, vcc\n" "v_cndmask_b32_e64 %7, %7, %8, vcc\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == P_CMP_CND) asm volatile(FS_X8R("v_cmp_lt_f32 vcc, %0, %8\n" "v_cndmask_b32 %1, %1, %9, vcc\n" "v_cmp_lt_f32 vcc, %2, %8\n" "v_cndmask_b32 %3, %3, %9, vcc\n" "v_cmp_lt_f32 vcc, %4, %8\n" "v_cndmask_b32 %5, %5, %9, vcc\n" "v_cmp_lt_f32 vcc, %6, %8\n" "v_cndmask_b32 %7, %7, %9, vcc\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm) : "vcc");
    if constexpr (OP == P_CMPS_CNDS) asm volatile(FS_X8R("v_cmp_lt_f32 %8, %0, %9\n" "v_cndmask_b32_e64 %1, %1, %10, %8\n" "v_cmp_lt_f32 %8, %2, %9\n" "v_cndmask_b32_e64 %3, %3, %10, %8\n" "v_cmp_lt_f32 %8, %4, %9\n" "v_cndmask_b32_e64 %5, %5, %10, %8\n" "v_cmp_lt_f32 %8, %6, %9\n" "v_cndmask_b32_e64 %7, %7, %10, %8\n" ) : FS_ACC8, "+s"(smw) : "v"(b), "v"(c));
    if constexpr (OP == P_FMA_MAX) asm volatile(FS_X8R("v_fma_f32 %0, %0, %8, %9\n" "v_max_f32 %1, %1, %9\n" "v_fma_f32 %2, %2, %8, %9\n" "v_max_f32 %3, %3, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_max_f32 %5, %5, %9\n" "v_fma_f32 %6, %6, %8, %9\n" "v_max_f32 %7, %7, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == P_FMA3_MAX) asm volatile(FS_X8R("v_fma_f32 %0, %0, %8, %9\n" "v_fma_f32 %1, %1, %8, %9\n" "v_fma_f32 %2, %2, %8, %9\n" "v_max_f32 %3, %3, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_fma_f32 %5, %5, %8, %9\n" "v_fma_f32 %6, %6, %8, %9\n" "v_max_f32 %7, %7, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == P_FMA_MUL_ADD) asm volatile(FS_X8R("v_fma_f32 %0, %0, %8, %9\n" "v_mul_f32 %1, %1, %8\n" "v_add_f32 %2, %2, %9\n" "v_mov_b32 %3, %8\n" "v_fma_f32 %4, %4, %8, %9\n" "v_mul_f32 %5, %5, %8\n" "v_add_f32 %6, %6, %9\n" "v_mov_b32 %7, %8\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == P_FMA_DPP_ALT) asm volatile(FS_X8R("v_fma_f32 %0, %0, %8, %9\n" "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" "v_fma_f32 %2, %2, %8, %9\n" "v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" "v_fma_f32 %4, %4, %8, %9\n" "v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" "v_fma_f32 %6, %6, %8, %9\n" "v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == P_FMA7_EXP) asm volatile(FS_X8R("v_fma_f32 %0, %0, %8, %9\n" "v_fma_f32 %1, %1, %8, %9\n" "v_fma_f32 %2, %2, %8, %9\n" "v_fma_f32 %3, %3, %8, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_fma_f32 %5, %5, %8, %9\n" "v_fma_f32 %6, %6, %8, %9\n" "v_exp_f32 %7, %7\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == P_FMA_SUB_MIN) asm volatile(FS_X8R("v_fma_f32 %0, %0, %8, %9\n" "v_sub_f32 %1, %1, %8\n" "v_min_f32 %2, %2, %9\n" "v_cmp_lt_f32 vcc, %3, %8\n" "v_fma_f32 %4, %4, %8, %9\n" "v_sub_f32 %5, %5, %8\n" "v_min_f32 %6, %6, %9\n" "v_cmp_lt_f32 vcc, %7, %8\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm) : "vcc");
    if constexpr (OP == P_FMA_LDS) asm volatile(FS_X8R("v_fma_f32 %0, %0, %9, %10\n" "v_fma_f32 %1, %1, %9, %10\n" "v_fma_f32 %2, %2, %9, %10\n" "v_fma_f32 %3, %3, %9, %10\n" "v_fma_f32 %4, %4, %9, %10\n" "v_fma_f32 %5, %5, %9, %10\n" "v_fma_f32 %6, %6, %9, %10\n" "ds_read_b128 %8, %11\n" ) "s_waitcnt lgkmcnt(0)\n" : FS_ACC8, "=v"(q[0]) : "v"(b), "v"(c), "v"(lds_addr_bc));
    if constexpr (OP == P_FMAC_MAX) asm volatile(FS_X8R("v_fmac_f32 %0, %8, %9\n" "v_max_f32 %1, %1, %9\n" "v_fmac_f32 %2, %8, %9\n" "v_max_f32 %3, %3, %9\n" "v_fmac_f32 %4, %8, %9\n" "v_max_f32 %5, %5, %9\n" "v_fmac_f32 %6, %8, %9\n" "v_max_f32 %7, %7, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == P_FMAC_FMA) asm volatile(FS_X8R("v_fmac_f32 %0, %8, %9\n" "v_fma_f32 %1, %1, %8, %9\n" "v_fmac_f32 %2, %8, %9\n" "v_fma_f32 %3, %3, %8, %9\n" "v_fmac_f32 %4, %8, %9\n" "v_fma_f32 %5, %5, %8, %9\n" "v_fmac_f32 %6, %8, %9\n" "v_fma_f32 %7, %7, %8, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == P_FMAC_MUL_ADD) asm volatile(FS_X8R("v_fmac_f32 %0, %8, %9\n" "v_mul_f32 %1, %1, %8\n" "v_add_f32 %2, %2, %9\n" "v_mov_b32 %3, %8\n" "v_fmac_f32 %4, %8, %9\n" "v_mul_f32 %5, %5, %8\n" "v_add_f32 %6, %6, %9\n" "v_mov_b32 %7, %8\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == P_CMP_CND3) asm volatile(FS_X8R("v_cmp_lt_f32 vcc, %0, %8\n" "v_cndmask_b32 %1, %1, %9, vcc\n" "v_cndmask_b32 %2, %2, %9, vcc\n" "v_cndmask_b32 %3, %3, %9, vcc\n" "v_cmp_lt_f32 vcc, %4, %8\n" "v_cndmask_b32 %5, %5, %9, vcc\n" "v_cndmask_b32 %6, %6, %9, vcc\n" "v_cndmask_b32 %7, %7, %9, vcc\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm) : "vcc");
    if constexpr (OP == P_CMP_FMA_CND) asm volatile(FS_X8R("v_cmp_lt_f32 vcc, %0, %8\n" "v_fma_f32 %1, %1, %8, %9\n" "v_fma_f32 %2, %2, %8, %9\n" "v_fma_f32 %3, %3, %8, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_cndmask_b32 %5, %5, %9, vcc\n" "v_fma_f32 %6, %6, %8, %9\n" "v_fma_f32 %7, %7, %8, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm) : "vcc");
    if constexpr (OP == P_FMA_MAX2) asm volatile(FS_X8R("v_fma_f32 %0, %0, %8, %9\n" "v_fma_f32 %1, %1, %8, %9\n" "v_max_f32 %2, %2, %9\n" "v_max_f32 %3, %3, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_fma_f32 %5, %5, %8, %9\n" "v_max_f32 %6, %6, %9\n" "v_max_f32 %7, %7, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == P_FMA_MAX_1_3) asm volatile(FS_X8R("v_fma_f32 %0, %0, %8, %9\n" "v_max_f32 %1, %1, %9\n" "v_max_f32 %2, %2, %9\n" "v_max_f32 %3, %3, %9\n" "v_fma_f32 %4, %4, %8, %9\n" "v_max_f32 %5, %5, %9\n" "v_max_f32 %6, %6, %9\n" "v_max_f32 %7, %7, %9\n" ) : FS_ACC8 : "v"(b), "v"(c), "s"(sb), "s"(sm));
    if constexpr (OP == ATOM_ISSUE) asm volatile(FS_ATOM32 FS_ATOM32 : : "v"(goff), "v"(b), "s"(gp) : "memory");
}

template <int OP, int EXEC>
__global__ void __launch_bounds__(1024) chain(Stamp *stamps, float *sink, float *gbuf, int reps, float b, float c) {
    __shared__ f4 lds[1024];  // 16 KB; the only LDS object of the kernel: starts at LDS address 0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    lds[tid] = f4{(float)tid, 1.f, 2.f, 3.f};
    __syncthreads();
    float a[8];
    f2 p[8];
    f4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 1.0f + 1e-3f * (float)(lane + i);
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f2{1.0f + 1e-3f * i, 1.0f - 1e-3f * lane}, q[i] = f4{0.f, 0.f, 0.f, 0.f};
    const unsigned l0 = 64u, l1 = (unsigned)tid * ((OP == LDS_B32) ? 4u : 16u), bp = (unsigned)((lane ^ 17) * 4);
    unsigned long long sg = 0;
    unsigned sg32 = 0;
    const float sb = __builtin_amdgcn_readfirstlane(__float_as_uint(b)) ? 1.0001f : b;  // (an SGPR copy of b)
    unsigned long long sm = 0x5555aaaa3333ccccull ^ (unsigned long long)reps, smw = sm;
    asm volatile("s_mov_b64 %0, %0" : "+s"(sm));
    const f2 pb = f2{b, b}, pc = f2{c, c};
    // ATOM_ISSUE: wave w of block k owns 16 groups of 4 lines (4 KB); lane -> (row's line, own dword)
    float *gp;
    {
        const unsigned long long g = (unsigned long long)(gbuf + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 1024);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)g), hi = __builtin_amdgcn_readfirstlane((unsigned)(g >> 32));
        gp = (float *)(((unsigned long long)hi << 32) | lo);  // wave-uniform: an SGPR pair
    }
    const unsigned goff = (unsigned)((lane >> 4) * 64 + (lane & 15) * 4);
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(a[1]) : "vcc");
    __syncthreads();
    unsigned long long saved;
    constexpr unsigned long long M = exec_mask(EXEC);
    const unsigned long long real0 = memrealtime();
    const unsigned long long t0 = memtime();
    if constexpr (EXEC != EX_FULL) asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=s"(saved) : "s"(M));
    for (int r = 0; r < reps; ++r) {
        body<OP>(a, p, q, b, c, pb, pc, l0, l1, bp, sg, sg32, gp, goff, sb, sm, smw);
    }
    if constexpr (EXEC != EX_FULL) asm volatile("s_mov_b64 exec, %0" : : "s"(saved));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long real1 = memrealtime();
    float s = (float)(sg & 1) + (float)(sg32 & 1) + (float)(smw & 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y + q[i].x + q[i].w;
    if (s == 12345.678f) sink[0] = s;  // keeps everything alive, never true in practice
    if (lane == 0) stamps[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = Stamp{t0, t1, real0, real1};
}

// ---- memory-side atomics ------------------------------------------------------------------------------------------
// One wave instruction = 64 lanes.  Patterns (what the 64 lanes of ONE instruction address):
enum AtomPattern {
    AP_ROW15 = 0,  // lanes 0..14 of each 16-lane row -> the 15 leading dwords of ONE 64-B line per row (4 lines / instruction)
    AP_ROW16,      // 16 lanes of each row -> a full 64-B line per row
    AP_LINE64,     // 64 lanes -> 64 DIFFERENT lines, one dword each
    AP_CONTIG,     // 64 lanes -> 256 contiguous bytes (4 lines), line chosen per wave
    AP_ONE,        // 64 lanes -> one address
    AP_STORE16,    // no atomic: plain 16-B per-lane streaming stores (WRITE_SIZE calibration)
    AP_STORE4,     // no atomic: plain 4-B per-lane streaming stores
    AP_STORE_ROW15,// no atomic: plain 4-B stores in the ROW15 pattern
    AP_LOAD16,     // no atomic: plain 16-B per-lane streaming loads (FETCH_SIZE calibration)
    N_AP
};
static const char *kApName[N_AP] = {"atomic: 15 lanes/row -> 1 line/row", "atomic: 16 lanes/row -> 1 line/row",
                                    "atomic: 64 lanes -> 64 lines",       "atomic: 64 lanes -> 256 B contiguous",
                                    "atomic: 64 lanes -> 1 address",      "plain store 16 B/lane",
                                    "plain store 4 B/lane",               "plain store, 15 lanes/row pattern",
                                    "plain load 16 B/lane"};

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// lines: footprint in 64-B lines (power of two).  unique: every (wave, iteration, row) gets a line of its own
// (streaming: footprint must be >= instructions * lines per instruction); else lines are drawn by hash (re-use, like
// the per-Gaussian gradient lines of a frame).
// MODE only names the instantiation (rocprofv3 reports per kernel name): 0 stream, 1 / 2 / 3 = hashed over 32 MB / 8 MB / 256 KB
template <int PAT, int MODE>
__global__ void __launch_bounds__(256) atom(float *buf, unsigned lines_mask, int iters, int unique, float *sink) {
    const unsigned lane = threadIdx.x & 63, row = lane >> 4, col = lane & 15;
    const unsigned gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // global wave id
    const unsigned nw = (gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const unsigned inst = (unsigned)it * nw + gw;  // instruction id, consecutive waves adjacent
        if constexpr (PAT == AP_ROW15 || PAT == AP_ROW16 || PAT == AP_STORE_ROW15) {
            const unsigned key = inst * 4u + row;
            const unsigned line = (unique ? key : hash32(key)) & lines_mask;
            float *p = buf + (size_t)line * 16 + col;
            if (PAT == AP_ROW16 || col < 15) {
                if constexpr (PAT == AP_STORE_ROW15)
                    __builtin_nontemporal_store(1.0f, p);
                else
                    atomicAdd(p, 1.0f);
            }
        } else if constexpr (PAT == AP_LINE64) {
            const unsigned key = inst * 64u + lane;
            const unsigned line = (unique ? key : hash32(key)) & lines_mask;
            atomicAdd(buf + (size_t)line * 16 + (lane & 15), 1.0f);
        } else if constexpr (PAT == AP_CONTIG) {
            const unsigned key = inst * 4u;
            const unsigned line = (unique ? key : (hash32(inst) * 4u)) & lines_mask;
            atomicAdd(buf + (size_t)line * 16 + lane, 1.0f);
        } else if constexpr (PAT == AP_ONE) {
            atomicAdd(buf + ((unique ? inst : 0u) & lines_mask) * 16, 1.0f);
        } else if constexpr (PAT == AP_STORE16) {
            const size_t o = ((size_t)inst * 64 + lane) & (((size_t)lines_mask + 1) * 4 - 1);  // in float4 units
            reinterpret_cast<f4 *>(buf)[o] = f4{1.f, 2.f, 3.f, 4.f};
        } else if constexpr (PAT == AP_STORE4) {
            const size_t o = ((size_t)inst * 64 + lane) & (((size_t)lines_mask + 1) * 16 - 1);
            buf[o] = 1.0f;
        } else if constexpr (PAT == AP_LOAD16) {
            const size_t o = ((size_t)inst * 64 + lane) & (((size_t)lines_mask + 1) * 4 - 1);
            f4 v = __builtin_nontemporal_load(reinterpret_cast<f4 *>(buf) + o);
            acc += v.x + v.w;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// ---- host ---------------------------------------------------------------------------------------------------------
struct ChainResult {
    double cyc_per_inst_wave[3];  // W = 1, 2, 4 waves per SIMD, one CU: per wave
    double clock_ghz[3];
    double chip_inst_per_ns_per_simd, chip_ms, chip_clock_ghz;
};

static int g_num_cu = 256;

template <int OP, int EXEC>
static ChainResult run_chain(Stamp *d_stamps, float *d_sink, float *d_gbuf, int reps, bool chip) {
    ChainResult R{};
    std::vector<Stamp> h(8192);
    const int Ws[3] = {1, 2, 4};
    for (int k = 0; k < 3; ++k) {
        const int threads = 256 * Ws[k];
        double best = 1e30, clk = 0;
        for (int rep = 0; rep < 3; ++rep) {  // best of 3 (first launch loads the code object)
            hipLaunchKernelGGL((chain<OP, EXEC>), dim3(1), dim3(threads), 0, 0, d_stamps, d_sink, d_gbuf, reps, 1.0001f, 1e-7f);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), d_stamps, sizeof(Stamp) * (threads / 64), hipMemcpyDeviceToHost));
            // the workgroup's span: first start to last end (waves of a SIMD interleave)
            unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0;
            for (int w = 0; w < threads / 64; ++w) {
                t0 = std::min(t0, h[w].cyc0), t1 = std::max(t1, h[w].cyc1);
                r0 = std::min(r0, h[w].real0), r1 = std::max(r1, h[w].real1);
            }
            const double c = (double)(t1 - t0) / ((double)reps * kUnroll);
            if (c < best) best = c, clk = (double)(t1 - t0) / ((double)(r1 - r0) * 10.0);  // 100 MHz ticks -> ns
        }
        R.cyc_per_inst_wave[k] = best;
        R.clock_ghz[k] = clk;
    }
    if (chip) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        const int blocks = g_num_cu * 8;
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((chain<OP, EXEC>), dim3(blocks), dim3(256), 0, 0, d_stamps, d_sink, d_gbuf, reps, 1.0001f, 1e-7f);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, (double)ms);
        }
        CK(hipMemcpy(h.data(), d_stamps, sizeof(Stamp) * 4 * 64, hipMemcpyDeviceToHost));
        double clk = 0;
        for (int w = 0; w < 256; ++w) clk += (double)(h[w].cyc1 - h[w].cyc0) / ((double)(h[w].real1 - h[w].real0) * 10.0) / 256.0;
        R.chip_ms = best;
        R.chip_clock_ghz = clk;
        const double wave_insts = (double)blocks * 4 * reps * kUnroll;
        R.chip_inst_per_ns_per_simd = wave_insts / (best * 1e6) / (g_num_cu * 4);
        CK(hipEventDestroy(e0));
        CK(hipEventDestroy(e1));
    }
    return R;
}

static void print_chain(const char *op, const char *ex, const ChainResult &r, bool chip) {
    printf("%-30s %-17s | %7.2f %7.2f %7.2f | %7.2f %7.2f %7.2f | %5.2f", op, ex, r.cyc_per_inst_wave[0], r.cyc_per_inst_wave[1],
           r.cyc_per_inst_wave[2], r.cyc_per_inst_wave[0] / 1, r.cyc_per_inst_wave[1] / 2, r.cyc_per_inst_wave[2] / 4, r.clock_ghz[0]);
    if (chip)
        printf(" | %8.3f ms  %6.3f inst/ns/SIMD  = %5.2f cyc/inst at %4.2f GHz", r.chip_ms, r.chip_inst_per_ns_per_simd,
               r.chip_clock_ghz / r.chip_inst_per_ns_per_simd, r.chip_clock_ghz);
    printf("\n");
    fflush(stdout);
}

#define RUN(OP, EX)                                                              \
    do {                                                                         \
        ChainResult r_ = run_chain<OP, EX>(d_stamps, d_sink, d_gbuf, (OP) == ATOM_ISSUE ? reps / 16 : reps, chip); \
        print_chain(kOpName[OP], kExecName[EX], r_, chip);                       \
    } while (0)

template <int PAT, int MODE = 0>
static void run_atom(float *buf, size_t buf_lines, size_t lines, int iters, int unique, float *d_sink, const char *what) {
    if (lines > buf_lines) {
        printf("%-38s %-28s skipped (footprint)\n", kApName[PAT], what);
        return;
    }
    const int blocks = g_num_cu * 8;
    const double insts = (double)blocks * 4 * iters;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(buf, 0, lines * 64, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((atom<PAT, MODE>), dim3(blocks), dim3(256), 0, 0, buf, (unsigned)(lines - 1), iters, unique, d_sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms);
    }
    const int lines_per_inst = (PAT == AP_LINE64) ? 64 : (PAT == AP_ONE) ? 1 : (PAT == AP_STORE16 || PAT == AP_LOAD16) ? 16 : 4;
    const int lanes = (PAT == AP_ROW15 || PAT == AP_STORE_ROW15) ? 60 : 64;
    const double bytes = insts * lanes * ((PAT == AP_STORE16 || PAT == AP_LOAD16) ? 16.0 : 4.0);
    printf("%-38s %-28s | %9.0f wave-inst x %2d lines | %8.3f ms | %7.2f G wave-inst/s | %7.2f G line-ops/s | %8.1f GB/s of operands\n",
           kApName[PAT], what, insts, lines_per_inst, best, insts / best / 1e6, insts * lines_per_inst / best / 1e6,
           bytes / best / 1e6);
    fflush(stdout);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    bool do_chains = true, do_atoms = true, chip = true, quick = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--only-chains")) do_atoms = false;
        if (!strcmp(argv[i], "--only-atoms")) do_chains = false;
        if (!strcmp(argv[i], "--no-chip")) chip = false;
        if (!strcmp(argv[i], "--quick")) quick = true;
    }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    g_num_cu = prop.multiProcessorCount;
    printf("# device: %s (%s), %d CUs, clockRate %.3f GHz, wavefront %d\n", prop.name, prop.gcnArchName, g_num_cu,
           prop.clockRate * 1e-6, prop.warpSize);
    Stamp *d_stamps;
    float *d_sink, *d_gbuf;
    CK(hipMalloc(&d_stamps, sizeof(Stamp) * 65536));
    CK(hipMalloc(&d_sink, 64));
    CK(hipMalloc(&d_gbuf, (size_t)g_num_cu * 8 * 4 * 1024 * sizeof(float)));  // 4 KB per wave of the chip-wide launch
    CK(hipMemset(d_gbuf, 0, (size_t)g_num_cu * 8 * 4 * 1024 * sizeof(float)));
    const int reps = quick ? 256 : 2048;  // x 64 instructions per wave

    if (do_chains) {
        printf("# chains: %d x %d instructions per wave.  Columns: cycles per instruction PER WAVE with 1 / 2 / 4 waves on the SIMD\n",
               reps, kUnroll);
        printf("#   (one workgroup of 256 / 512 / 1024 threads on one CU, s_memtime); the same PER SIMD (= per wave / waves);\n");
        printf("#   clock = s_memtime ticks per ns of s_memrealtime; chip-wide: %d workgroups of 256 (8 waves per SIMD), hipEvents\n", g_num_cu * 8);
        printf("%-30s %-17s | %7s %7s %7s | %7s %7s %7s | %5s\n", "instruction", "EXEC", "wave W1", "wave W2", "wave W4", "SIMD W1",
               "SIMD W2", "SIMD W4", "GHz");
        RUN(FMA, EX_FULL);
        RUN(FMA_DEP, EX_FULL);
        RUN(MUL, EX_FULL);
        RUN(ADD, EX_FULL);
        RUN(MAXF, EX_FULL);
        RUN(MOV, EX_FULL);
        RUN(EXP, EX_FULL);
        RUN(RCP, EX_FULL);
        RUN(LOG, EX_FULL);
        RUN(DPP_SHR, EX_FULL);
        RUN(DPP_BCAST15, EX_FULL);
        RUN(DPP_QUAD, EX_FULL);
        RUN(PERM32, EX_FULL);
        RUN(PERM16, EX_FULL);
        RUN(CNDMASK, EX_FULL);
        RUN(CMP, EX_FULL);
        RUN(CMP_SGPR, EX_FULL);
        RUN(PK_FMA, EX_FULL);
        RUN(PK_MUL, EX_FULL);
        RUN(READLANE, EX_FULL);
        RUN(BPERMUTE, EX_FULL);
        RUN(SWIZZLE, EX_FULL);
        RUN(LDS_B128_BC, EX_FULL);
        RUN(LDS_B32_BC, EX_FULL);
        RUN(LDS_B128, EX_FULL);
        RUN(LDS_B32, EX_FULL);
        RUN(MIX_FMA_DPP, EX_FULL);
        RUN(MIX_FMA_EXP, EX_FULL);
        RUN(G_SUB, EX_FULL);
        RUN(G_FMAC, EX_FULL);
        RUN(G_MIN, EX_FULL);
        RUN(G_MUL_E64, EX_FULL);
        RUN(G_ADD_NEG, EX_FULL);
        RUN(G_FMA_SGPR, EX_FULL);
        RUN(G_FMA_CONST, EX_FULL);
        RUN(G_MUL_LIT, EX_FULL);
        RUN(G_AND, EX_FULL);
        RUN(G_OR, EX_FULL);
        RUN(G_LSHL, EX_FULL);
        RUN(G_ADDU, EX_FULL);
        RUN(G_SUBU, EX_FULL);
        RUN(G_ADDCO, EX_FULL);
        RUN(G_CVT_F_I, EX_FULL);
        RUN(G_CVT_I_F, EX_FULL);
        RUN(G_MULLO, EX_FULL);
        RUN(G_BFE, EX_FULL);
        RUN(G_MED3, EX_FULL);
        RUN(G_MAX3, EX_FULL);
        RUN(G_LDEXP, EX_FULL);
        RUN(G_SQRT, EX_FULL);
        RUN(G_RSQ, EX_FULL);
        RUN(G_MOV_DPP, EX_FULL);
        RUN(G_MBCNT, EX_FULL);
        RUN(G_CND_SGPR, EX_FULL);
        RUN(G_CND_VCC_D, EX_FULL);
        RUN(G_CND_E64_VCC, EX_FULL);
        RUN(P_CMP_CND, EX_FULL);
        RUN(P_CMPS_CNDS, EX_FULL);
        RUN(P_FMA_MAX, EX_FULL);
        RUN(P_FMA3_MAX, EX_FULL);
        RUN(P_FMA_MUL_ADD, EX_FULL);
        RUN(P_FMA_DPP_ALT, EX_FULL);
        RUN(P_FMA7_EXP, EX_FULL);
        RUN(P_FMA_SUB_MIN, EX_FULL);
        RUN(P_FMA_LDS, EX_FULL);
        RUN(P_FMAC_MAX, EX_FULL);
        RUN(P_FMAC_FMA, EX_FULL);
        RUN(P_FMAC_MUL_ADD, EX_FULL);
        RUN(P_CMP_CND3, EX_FULL);
        RUN(P_CMP_FMA_CND, EX_FULL);
        RUN(P_FMA_MAX2, EX_FULL);
        RUN(P_FMA_MAX_1_3, EX_FULL);
        // EXEC dependence
        RUN(FMA, EX_LOW32);
        RUN(FMA, EX_EVEN);
        RUN(FMA, EX_LOW16);
        RUN(FMA, EX_ONE);
        RUN(FMA, EX_ZERO);
        RUN(EXP, EX_LOW32);
        RUN(EXP, EX_ZERO);
        RUN(DPP_SHR, EX_LOW32);
        RUN(DPP_SHR, EX_ZERO);
        RUN(PERM32, EX_ZERO);
        RUN(CNDMASK, EX_ZERO);
        RUN(LDS_B128_BC, EX_LOW32);
        RUN(LDS_B128_BC, EX_ZERO);
        RUN(ATOM_ISSUE, EX_ROW15);
        RUN(ATOM_ISSUE, EX_ZERO);
    }
    if (do_atoms) {
        const size_t buf_lines = (size_t)1 << 24;  // 1 GiB of 64-B lines
        float *buf;
        CK(hipMalloc(&buf, buf_lines * 64));
        CK(hipMemset(buf, 0, buf_lines * 64));
        const int it = quick ? 32 : 128;  // x 8192 waves = 1.05 M wave instructions (x 4 lines = 4.2 M row atomics)
        printf("# memory-side atomics and plain accesses, chip-wide (%d workgroups of 256): %d instructions per wave\n", g_num_cu * 8, it);
        // footprints: streaming (every line once), 32 MB / 8 MB / 256 KB drawn by hash (re-use)
        const size_t n_inst = (size_t)g_num_cu * 8 * 4 * it;
        size_t stream4 = 1;
        while (stream4 < n_inst * 4) stream4 <<= 1;
        size_t stream64 = 1;
        while (stream64 < n_inst / 4 * 64) stream64 <<= 1;  // (the 64-lines pattern runs it / 4 instructions per wave)
        run_atom<AP_ROW15>(buf, buf_lines, stream4, it, 1, d_sink, "every line once (stream)");
        run_atom<AP_ROW15, 1>(buf, buf_lines, (size_t)1 << 19, it, 0, d_sink, "hashed over 32 MB");
        run_atom<AP_ROW15, 2>(buf, buf_lines, (size_t)1 << 17, it, 0, d_sink, "hashed over 8 MB");
        run_atom<AP_ROW15, 3>(buf, buf_lines, (size_t)1 << 12, it, 0, d_sink, "hashed over 256 KB");
        run_atom<AP_ROW16>(buf, buf_lines, stream4, it, 1, d_sink, "every line once (stream)");
        run_atom<AP_ROW16, 1>(buf, buf_lines, (size_t)1 << 19, it, 0, d_sink, "hashed over 32 MB");
        run_atom<AP_LINE64>(buf, buf_lines, stream64, it / 4, 1, d_sink, "every line once (stream)");
        run_atom<AP_LINE64, 1>(buf, buf_lines, (size_t)1 << 19, it / 4, 0, d_sink, "hashed over 32 MB");
        run_atom<AP_CONTIG>(buf, buf_lines, stream4, it, 1, d_sink, "every line once (stream)");
        run_atom<AP_CONTIG, 1>(buf, buf_lines, (size_t)1 << 19, it, 0, d_sink, "hashed over 32 MB");
        run_atom<AP_ONE, 3>(buf, buf_lines, 1, it / 4, 0, d_sink, "one address for all");
        run_atom<AP_ONE>(buf, buf_lines, stream4, it, 1, d_sink, "one address per instruction");
        run_atom<AP_STORE16>(buf, buf_lines, buf_lines, it * 4, 1, d_sink, "stream");
        run_atom<AP_STORE4>(buf, buf_lines, buf_lines, it * 4, 1, d_sink, "stream");
        run_atom<AP_STORE_ROW15>(buf, buf_lines, stream4, it, 1, d_sink, "every line once (stream)");
        run_atom<AP_STORE_ROW15, 1>(buf, buf_lines, (size_t)1 << 19, it, 0, d_sink, "hashed over 32 MB");
        run_atom<AP_LOAD16>(buf, buf_lines, buf_lines, it * 4, 1, d_sink, "stream");
        CK(hipFree(buf));
    }
    return 0;
}
