// Shared helpers for the gfx950 kernels of libfsgs.so.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/fsgs.h"

// The hand-rolled inter-workgroup hand-offs of this library (raster_quad.hip: the forward's tail queue; sh.hip: the scan
// launch's last workgroup forming isect_offsets) publish data with relaxed agent-scope atomic stores — which gfx942 / gfx950
// compile to write-through (sc1) stores — followed by `s_waitcnt vmcnt(0)` and a tagged word or a relaxed ticket, and read
// it back with relaxed agent-scope loads.  That is OUTSIDE the HIP memory model: it relies on stores being counted by
// vmcnt and on sc1 accesses bypassing the non-coherent caches, as they do on these two targets (MI355X_MICROARCH.md,
// "Inter-workgroup visibility"; an architecture with a separate store counter, or a compiler that stops emitting sc1 for
// agent scope, would let a reader see stale data with no error).  Refuse to build for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "libfsgs's write-through hand-off protocols are validated for gfx942 / gfx950 only (see common.h)"
#endif


namespace fsgs {

constexpr int kWave = 64;
constexpr float kAlphaMax = 0.999f;       // gsplat: alpha = min(0.999, opac*exp(-sigma))
constexpr float kAlphaMin = 1.0f / 255.f; // skip below
constexpr float kTMin = 1e-4f;            // stop before T would reach this

extern thread_local int g_last_hip_error;

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return FSGS_ELAUNCH;
    }
    return FSGS_OK;
}

inline hipStream_t as_stream(fsgs_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): remembered per device of the calling
// thread, with atomics (the library is re-entrant across threads and a process may drive several GPUs).  A lost race
// sets the same attribute twice, which is harmless.
constexpr int kMaxDevices = 64;
template <auto Kernel>
inline int ensure_dynamic_lds(size_t need) {
    static std::atomic<size_t> have[kMaxDevices] = {};
    int dev = -1;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices;
    if (known && need <= have[dev].load(std::memory_order_acquire)) return FSGS_OK;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return FSGS_ELAUNCH;
    }
    if (known) {
        size_t cur = have[dev].load(std::memory_order_relaxed);
        while (cur < need && !have[dev].compare_exchange_weak(cur, need, std::memory_order_release)) {}
    }
    return FSGS_OK;
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- wave64 reductions on DPP (no LDS traffic) ------------------------------------------
// dpp_ctrl encodings (GCN3+/CDNA): quad_perm = sel0|sel1<<2|sel2<<4|sel3<<6,
// row_half_mirror 0x141, row_mirror 0x140, row_bcast15 0x142, row_bcast31 0x143.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}

// Sum over the 64 lanes; the total is valid in lanes 48..63 (read it from lane 63).
__device__ __forceinline__ float wave_sum_to_last_row(float v) {
    v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);       // row_half_mirror: 8-lane sums
    v += dpp_f<0x140>(v);       // row_mirror: every lane holds its 16-lane row sum
    v += dpp_f<0x142, 0xA>(v);  // row_bcast15 into rows 1,3
    v += dpp_f<0x143, 0xC>(v);  // row_bcast31 into rows 2,3
    return v;
}

// Sum 12 values within each 16-lane row (every lane of a row ends with the 12 row sums).
// Hand-written: under -O3 hipcc SLP-packs the adds into v_pk_add_f32, which cannot carry a DPP
// operand, and emits v_mov 0 + v_mov_dpp + v_pk_add per step (~2.5 instructions per value per
// step).  Here each step is one v_add_f32_dpp per value.  The 12 chains are independent, so within
// a block every DPP source was written >= 11 instructions earlier (the VALU-write -> DPP-read
// hazard needs 2 wait states); the leading s_nop covers the first block's inputs.
#define FSGS_DPP12(CTRL)                                                                                   \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_add_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_add_f32_dpp %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_add_f32_dpp %8, %8, %8 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_add_f32_dpp %9, %9, %9 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                      \
    "v_add_f32_dpp %10, %10, %10 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                                   \
    "v_add_f32_dpp %11, %11, %11 " CTRL " row_mask:0xf bank_mask:0xf\n\t"

__device__ __forceinline__ void row_sum16_x12(float (&g)[12]) {
    asm volatile("s_nop 1\n\t" FSGS_DPP12("quad_perm:[1,0,3,2]") FSGS_DPP12("quad_perm:[2,3,0,1]")
                     FSGS_DPP12("row_half_mirror") FSGS_DPP12("row_mirror")
                 : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]),
                   "+v"(g[7]), "+v"(g[8]), "+v"(g[9]), "+v"(g[10]), "+v"(g[11]));
}

// three more values (the normal plane's colour gradients); consecutive steps of one value are
// separated by the other two values' instructions = the 2 wait states the DPP read needs.
#define FSGS_DPP3(CTRL)                                                  \
    "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"    \
    "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"    \
    "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"

__device__ __forceinline__ void row_sum16_x3(float &a, float &b, float &c) {
    asm volatile("s_nop 1\n\t" FSGS_DPP3("quad_perm:[1,0,3,2]") FSGS_DPP3("quad_perm:[2,3,0,1]")
                     FSGS_DPP3("row_half_mirror") FSGS_DPP3("row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c));
}

// Sixteen values summed over each 16-lane row, TRANSPOSED: afterwards lane l holds (in v[0]) the row
// sum of value (l & 15).  An all-reduce of 16 values costs 16 x 4 DPP adds; here every round halves the
// number of values a lane still carries: lanes of the lower half keep v[k], lanes of the upper half
// take over v[k + half] INTO register v[k] (two bank-masked DPP adds per surviving value; inside a
// quad, where bank masks cannot tell lanes apart, two adds and a select): 16 + 8 + 6 + 3 = 33.
// Hand-written for the same reason as above (and for the 2-wait-state VALU-write -> DPP-read hazard).
#define FSGS_T1(K, K8) \
    "v_add_f32_dpp %" #K ", %" #K ", %" #K " row_shl:8 row_mask:0xf bank_mask:0x3\n\t" \
    "v_add_f32_dpp %" #K ", %" #K8 ", %" #K8 " row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
#define FSGS_T2(K, K4) \
    "v_add_f32_dpp %" #K ", %" #K ", %" #K " row_shl:4 row_mask:0xf bank_mask:0x5\n\t" \
    "v_add_f32_dpp %" #K ", %" #K4 ", %" #K4 " row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
__device__ __forceinline__ void row_transpose_sum16(float (&v)[16]) {
    float ta, tb;
    const unsigned long long m2 = 0xCCCCCCCCCCCCCCCCull, m1 = 0xAAAAAAAAAAAAAAAAull;
    asm volatile(
        "s_nop 1\n\t"
        FSGS_T1(0, 8) FSGS_T1(1, 9) FSGS_T1(2, 10) FSGS_T1(3, 11) FSGS_T1(4, 12) FSGS_T1(5, 13) FSGS_T1(6, 14) FSGS_T1(7, 15)
        FSGS_T2(0, 4) FSGS_T2(1, 5) FSGS_T2(2, 6) FSGS_T2(3, 7)
        "v_add_f32_dpp %16, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %17, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %0, %16, %17, %18\n\t"
        "v_add_f32_dpp %16, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %17, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %1, %16, %17, %18\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %16, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %17, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %0, %16, %17, %19\n\t"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
          "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]),
          "=&v"(ta), "=&v"(tb)
        : "s"(m2), "s"(m1));
}

// Sixteen values summed over ALL 64 lanes, transposed: afterwards lane (row r, column c) holds the wave total of value
// 4 r + (c >> 2) (every lane of a quad the same one).  For a wave whose four rows work on the SAME record: the halves
// and the row pairs meet through gfx950's v_permlane32_swap / v_permlane16_swap (swap lanes 32..63 of the first operand
// with lanes 0..31 of the second / the odd rows of the first with the even rows of the second; each exchange + add halves
// the values a lane carries: 16 -> 8 -> 4), the 16 columns through bank-masked DPP adds as in row_transpose_sum16
// (4 -> 2 -> 1) and two quad butterflies: 12 swaps + 12 adds + 4 + 2 + 2 = 32 instructions, against 36 + 6 for the row
// transpose followed by a cross-row sum.  hipcc's hazard recognizer places the wait states around the swaps.
__device__ __forceinline__ float wave_transpose_sum16(float (&v)[16]) {
    auto u = [](float x) { return __builtin_bit_cast(unsigned, x); };
    auto f = [](unsigned x) { return __builtin_bit_cast(float, x); };
    float s[8], t[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const auto h = __builtin_amdgcn_permlane32_swap(u(v[k]), u(v[k + 8]), false, false);
        s[k] = f(h[0]) + f(h[1]);  // lanes 0..31: value k over (l, l + 32); lanes 32..63: value k + 8
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const auto r = __builtin_amdgcn_permlane16_swap(u(s[k]), u(s[k + 4]), false, false);
        t[k] = f(r[0]) + f(r[1]);  // row r: value 4 r + k over the four rows
    }
    asm volatile(
        "s_nop 1\n\t"
        FSGS_T1(0, 2) FSGS_T1(1, 3)
        "s_nop 1\n\t"  // (a DPP read needs two wait states after the VALU write of its source)
        FSGS_T2(0, 1)
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
    return t[0];
}

// one Adam update (torch.optim.Adam, amsgrad = False, weight_decay = 0): ss = lr / (1 - b1^t), isb2 = 1 / sqrt(1 - b2^t)
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float b1, float b2, float omb1,
                                         float omb2, float ss, float isb2, float eps) {
    // (no FMA contraction: torch rounds every product, and the kernels that inline this — the Adam launch, the SH
    // rebuild + Adam, the count pass that carries a riding step — must agree to the bit)
#pragma clang fp contract(off)
    m = b1 * m + omb1 * g;
    v = b2 * v + omb2 * g * g;
    p -= ss * m / (sqrtf(v) * isb2 + eps);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- spreading the gradient atomics of large Gaussians --------------------------------------------------------
// The compositing backward adds every (row, record) total into the Gaussian's packed 64-byte gradient line with one
// memory-side atomic transaction; transactions on one line are serial.  A Gaussian that covers many tiles receives
// them from hundreds of workgroups at once: on BASELINE config #3 (large background Gaussians behind a small
// object) a third of the kernel was spent there.  Gaussians whose 2-D footprint is large (det(conic) small, i.e.
// sigma_1 sigma_2 > 64 px^2) therefore own kGradReplicas lines, `replica_rows` lines apart; a workgroup picks the
// replica from its tile and quadrant, the SH backward sums the colour parts and the per-Gaussian backward (the last
// reader) folds and clears them.  Writer and last reader evaluate the same predicate on the same fp32 conic, so they
// always agree.  The threshold is measured (profiles/README.md): lower ones cost the two readers more than the
// compositing backward gains on config #2, higher ones lose the gain on config #3.
constexpr int kGradReplicas = 4;
// A reader that has the radius at hand but not the conic may use the superset `radius >= kSpreadMinRadius`:
// det(cov) > 4096 means sigma_max > 8 px, i.e. radius = ceil(3 sigma_max) >= 25; replicas never written hold zeros.
constexpr int kSpreadMinRadius = 24;
__device__ __forceinline__ bool grad_spread(float ca, float cb, float cc) {
    return __builtin_fmaf(-cb, cb, ca * cc) < (1.f / 4096.f);  // (one fixed rounding sequence in every kernel)
}
// ---- the SH forward riding in the binning's scan launch (sh.hip: scan_rows_sh_pack_kernel) ---------------------
// What fsgs_sh_fwd_pack needs, handed to the projecting count pass so that it can launch the colours + packing
// together with its table scan (16 stored coefficients, split features, one camera).
struct ShPackRider {
    int N, degree, attr_half;
    const float *means, *campos;
    const void *features_dc, *features_rest;   // fp32, or IEEE-half mirrors when attr_half
    const int32_t *radii;
    const float *depths, *means2d, *conics, *opacities;
    const void *quats, *log_scales;            // fp32 / half like the features
    const float *c2w;
    float *packed, *normals_world, *zero_cells;
    int n_zero;
    const uint8_t *kept;                       // nullable [N]: 0 = the Gaussian has no binned pair (no colours, no record)
};

// ---- attribute storage (BASELINE config #5) ---------------------------------------------------------------------
// Per-Gaussian attributes other than the means (SH features, log-scales, quaternions, opacity logits) are read either
// as fp32 or as IEEE half mirrors of the fp32 master parameters (`half` != 0, uniform per launch): all arithmetic
// stays fp32, only the bytes per Gaussian on the render path shrink (236 -> 124).  Adam keeps the masters and
// refreshes the mirrors in the same launch (glue.hip).
__device__ __forceinline__ float ld_attr(const void *p, int64_t i, int half) {
    return half ? __half2float(reinterpret_cast<const __half *>(p)[i]) : reinterpret_cast<const float *>(p)[i];
}
__device__ __forceinline__ float4 ld_attr4(const void *p, int64_t n, int half) {  // record n of four elements
    if (half) {
        const uint2 r = reinterpret_cast<const uint2 *>(p)[n];
        const float2 a = __half22float2(*reinterpret_cast<const __half2 *>(&r.x));
        const float2 b = __half22float2(*reinterpret_cast<const __half2 *>(&r.y));
        return make_float4(a.x, a.y, b.x, b.y);
    }
    return reinterpret_cast<const float4 *>(p)[n];
}
__device__ __forceinline__ void ld_attr3(const void *p, int64_t n, int half, float (&o)[3]) {  // record n of three
    o[0] = ld_attr(p, n * 3 + 0, half);
    o[1] = ld_attr(p, n * 3 + 1, half);
    o[2] = ld_attr(p, n * 3 + 2, half);
}

// Non-temporal 16-byte accesses for data that is written once and read once (record streams, Adam streams of large
// scenes): they leave the caches to the data that is re-used (gathered records, gradient lines).
typedef float fsgs_nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load_f4(const float4 *p) {
    const fsgs_nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const fsgs_nt_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store_f4(const float4 &v, float4 *p) {
    fsgs_nt_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<fsgs_nt_f4 *>(p));
}

}  // namespace fsgs
