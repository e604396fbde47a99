// E3: view-dependent colour from real spherical harmonics, forward and VJP.
// Replaces gsplat 1.0.0 `compute_sh_{fwd,bwd}` plus the torch glue that rasterization() wraps
// around it (`dirs = means - campos`, `+0.5`, `clamp_min(0)`, depth-channel concat for
// render_mode "RGB+ED") — call site /root/reference/dn_splatter/dn_model.py:570-591.
//
// Layout: coeffs[N,K,3] (K*12 = 192 B per Gaussian at degree 3) is the widest per-Gaussian
// stream on the whole path.  A 256-thread workgroup stages its 256 consecutive Gaussians'
// records (one contiguous 48 KiB span) through LDS with 16-byte coalesced loads, then each
// lane consumes its own record; LDS rows are padded by one dword so the per-lane stride is odd.
#include <math.h>

#include <atomic>
#include <cstdlib>
#include "common.h"
#include "normal_math.h"
#include "gauss_bwd_body.h"
#include "tile_scan.h"

namespace fsgs {

constexpr int kMaxK = 25;

// 1 / |d| with one fixed rounding sequence (see sh_basis)
__device__ __forceinline__ float sh_inv_norm(float dx, float dy, float dz) {
#pragma clang fp contract(off)
    return 1.f / fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
}

// basis values b[k] and (optionally) their gradients w.r.t. the unit direction.
template <bool GRAD>
__device__ __forceinline__ void sh_basis(int degree, float x, float y, float z, float *b, float *bx,
                                         float *by, float *bz) {
    // (no FMA contraction: the kernels that evaluate the basis — forward, backward, and the data-parallel rebuild of
    // the backward from its factors — are separate inlines of these statements and must agree to the bit)
#pragma clang fp contract(off)
    b[0] = 0.2820947917738781f;
    if (GRAD) bx[0] = by[0] = bz[0] = 0.f;
    if (degree < 1) return;
    const float C1 = 0.48860251190292f;
    b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
    if (GRAD) {
        bx[1] = 0.f; by[1] = -C1; bz[1] = 0.f;
        bx[2] = 0.f; by[2] = 0.f; bz[2] = C1;
        bx[3] = -C1; by[3] = 0.f; bz[3] = 0.f;
    }
    if (degree < 2) return;
    const float z2 = z * z;
    const float fTmp0B = -1.092548430592079f * z;
    const float fC1 = x * x - y * y;
    const float fS1 = 2.f * x * y;
    const float kA = 0.5462742152960395f;
    const float pSH6 = 0.9461746957575601f * z2 - 0.3153915652525201f;
    b[4] = kA * fS1; b[5] = fTmp0B * y; b[6] = pSH6; b[7] = fTmp0B * x; b[8] = kA * fC1;
    if (GRAD) {
        bx[4] = kA * 2.f * y; by[4] = kA * 2.f * x; bz[4] = 0.f;
        bx[5] = 0.f; by[5] = fTmp0B; bz[5] = -1.092548430592079f * y;
        bx[6] = 0.f; by[6] = 0.f; bz[6] = 2.f * 0.9461746957575601f * z;
        bx[7] = fTmp0B; by[7] = 0.f; bz[7] = -1.092548430592079f * x;
        bx[8] = kA * 2.f * x; by[8] = -kA * 2.f * y; bz[8] = 0.f;
    }
    if (degree < 3) return;
    const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    const float fTmp1B = 1.445305721320277f * z;
    const float fC2 = x * fC1 - y * fS1;
    const float fS2 = x * fS1 + y * fC1;
    const float pSH12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    const float kB = 0.5900435899266435f;
    b[9] = -kB * fS2; b[10] = fTmp1B * fS1; b[11] = fTmp0C * y; b[12] = pSH12;
    b[13] = fTmp0C * x; b[14] = fTmp1B * fC1; b[15] = -kB * fC2;
    const float fTmp0C_z = -2.f * 2.285228997322329f * z;
    const float pSH12_z = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
    if (GRAD) {
        bx[9] = -kB * 3.f * fS1; by[9] = -kB * 3.f * fC1; bz[9] = 0.f;
        bx[10] = fTmp1B * 2.f * y; by[10] = fTmp1B * 2.f * x; bz[10] = 1.445305721320277f * fS1;
        bx[11] = 0.f; by[11] = fTmp0C; bz[11] = fTmp0C_z * y;
        bx[12] = 0.f; by[12] = 0.f; bz[12] = pSH12_z;
        bx[13] = fTmp0C; by[13] = 0.f; bz[13] = fTmp0C_z * x;
        bx[14] = fTmp1B * 2.f * x; by[14] = -fTmp1B * 2.f * y; bz[14] = 1.445305721320277f * fC1;
        bx[15] = -kB * 3.f * fC1; by[15] = kB * 3.f * fS1; bz[15] = 0.f;
    }
    if (degree < 4) return;
    const float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    const float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
    const float fTmp2B = -1.770130769779931f * z;
    const float fC3 = x * fC2 - y * fS2;
    const float fS3 = x * fS2 + y * fC2;
    const float kC = 0.6258357354491763f;
    b[16] = kC * fS3; b[17] = fTmp2B * fS2; b[18] = fTmp1C * fS1; b[19] = fTmp0D * y;
    b[20] = 1.984313483298443f * z * pSH12 - 1.006230589874905f * pSH6;
    b[21] = fTmp0D * x; b[22] = fTmp1C * fC1; b[23] = fTmp2B * fC2; b[24] = kC * fC3;
    if (GRAD) {
        const float fTmp0D_z = -3.f * 4.683325804901025f * z2 + 2.007139630671868f;
        const float fTmp1C_z = 2.f * 3.31161143515146f * z;
        bx[16] = kC * 4.f * fS2; by[16] = kC * 4.f * fC2; bz[16] = 0.f;
        bx[17] = fTmp2B * 3.f * fS1; by[17] = fTmp2B * 3.f * fC1; bz[17] = -1.770130769779931f * fS2;
        bx[18] = fTmp1C * 2.f * y; by[18] = fTmp1C * 2.f * x; bz[18] = fTmp1C_z * fS1;
        bx[19] = 0.f; by[19] = fTmp0D; bz[19] = fTmp0D_z * y;
        bx[20] = 0.f; by[20] = 0.f;
        bz[20] = 1.984313483298443f * (pSH12 + z * pSH12_z) -
                 1.006230589874905f * 2.f * 0.9461746957575601f * z;
        bx[21] = fTmp0D; by[21] = 0.f; bz[21] = fTmp0D_z * x;
        bx[22] = fTmp1C * 2.f * x; by[22] = -fTmp1C * 2.f * y; bz[22] = fTmp1C_z * fC1;
        bx[23] = fTmp2B * 3.f * fC1; by[23] = -fTmp2B * 3.f * fS1; bz[23] = -1.770130769779931f * fC2;
        bx[24] = kC * 4.f * fC2; by[24] = -kC * 4.f * fS2; bz[24] = 0.f;
    }
}

#ifndef FSGS_SH_BLOCK
#define FSGS_SH_BLOCK 256
#endif
constexpr int kShBlock = FSGS_SH_BLOCK;  // Gaussians (= threads) per workgroup

// Stage `rows` consecutive records of `row_floats` floats (contiguous in HBM) into LDS with a row pitch of
// `pitch` dwords, 16 bytes per lane per load.  RF > 0 fixes row_floats at compile time: the (row, column)
// of a float4's first element then costs a multiply-shift instead of a ~20-instruction integer division
// per ELEMENT, and the other three follow by increment-and-wrap.  Loading and parking are separate calls so
// that a kernel can issue EVERY global load it needs (both coefficient arrays, means, radii, gradients)
// before it waits for the first one: written as one loop the compiler waits for each load in turn, and
// a workgroup then spends a dozen serialised HBM round trips (~20 us of a 23 us kernel) staging.
template <int RF>
struct Staged {
    float4 v[RF ? (RF + 3) / 4 : 1];
};

template <int RF, bool HALF = false>
__device__ __forceinline__ void stage_load(Staged<RF> &st, const void *__restrict__ src_v, int rows) {
    if (!RF) return;
    if (HALF) {
        // IEEE-half storage (BASELINE config #5): 8 elements per 16-byte load; a block's span starts at
        // n0 * RF halves, n0 a multiple of 256: 16-byte aligned whenever the tensor base is
        const uint4 *src8 = reinterpret_cast<const uint4 *>(src_v);
        const int total8 = (rows * RF) >> 3;
#pragma unroll
        for (int it = 0; it < (RF + 7) / 8; ++it) {
            const int i = threadIdx.x + it * kShBlock;
            if (i < total8) {
                const uint4 r = src8[i];
                st.v[it] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            }
        }
        return;
    }
    const float *src = reinterpret_cast<const float *>(src_v);
    // src is 4-byte aligned only in general (row start = n0*K*3 floats); n0 is a multiple of 256
    // so the span start is 16-byte aligned whenever the tensor base is.
    const float4 *src4 = reinterpret_cast<const float4 *>(src);
    const int total4 = (rows * RF) >> 2;
#pragma unroll
    for (int it = 0; it < (RF + 3) / 4; ++it) {
        const int i = threadIdx.x + it * kShBlock;
        if (i < total4) st.v[it] = src4[i];  // (a select between float4s would go through scratch memory)
    }
}

__device__ __forceinline__ void unpack_half8(const float4 &v, float (&o)[8]) {
    const unsigned w[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float2 f = __half22float2(*reinterpret_cast<const __half2 *>(&w[k]));
        o[2 * k] = f.x;
        o[2 * k + 1] = f.y;
    }
}

// the half-storage counterpart of stage_store (rows parked in LDS as fp32)
template <int RF>
__device__ __forceinline__ void stage_store_half(const Staged<RF> &st, const void *__restrict__ src_v, int rows,
                                                 int row_floats_rt, float *lds, int pitch, int col_off) {
    const int row_floats = RF ? RF : row_floats_rt;
    const int total = rows * row_floats;
    lds += col_off;
    const int total8 = total >> 3;
    if (RF) {
#pragma unroll
        for (int it = 0; it < (RF + 7) / 8; ++it) {
            const int i = threadIdx.x + it * kShBlock;
            if (i < total8) {
                float vv[8];
                unpack_half8(st.v[it], vv);
                const int e = i << 3;
                int r = e / row_floats, col = e - r * row_floats;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    lds[r * pitch + col] = vv[k];
                    if (++col == row_floats) { col = 0; ++r; }
                }
            }
        }
    } else {
        const uint4 *src8 = reinterpret_cast<const uint4 *>(src_v);
        for (int i = threadIdx.x; i < total8; i += kShBlock) {
            const uint4 rr = src8[i];
            float vv[8];
            unpack_half8(make_float4(__uint_as_float(rr.x), __uint_as_float(rr.y), __uint_as_float(rr.z), __uint_as_float(rr.w)), vv);
            const int e = i << 3;
            int r = e / row_floats, col = e - r * row_floats;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                lds[r * pitch + col] = vv[k];
                if (++col == row_floats) { col = 0; ++r; }
            }
        }
    }
    const __half *srch = reinterpret_cast<const __half *>(src_v);
    for (int i = (total8 << 3) + threadIdx.x; i < total; i += kShBlock) {
        const int r = i / row_floats, col = i - r * row_floats;
        lds[r * pitch + col] = __half2float(srch[i]);
    }
}

template <int RF>
__device__ __forceinline__ void stage_store(const Staged<RF> &st, const float *__restrict__ src, int rows,
                                            int row_floats_rt, float *lds, int pitch, int col_off) {
    const int row_floats = RF ? RF : row_floats_rt;
    const int total = rows * row_floats;
    lds += col_off;
    const int total4 = total >> 2;
    if (RF) {
#pragma unroll
        for (int it = 0; it < (RF + 3) / 4; ++it) {
            const int i = threadIdx.x + it * kShBlock;
            if (i < total4) {
                const int e = i << 2;
                const float vv[4] = {st.v[it].x, st.v[it].y, st.v[it].z, st.v[it].w};
                int r = e / row_floats, col = e - r * row_floats;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lds[r * pitch + col] = vv[k];
                    if (++col == row_floats) { col = 0; ++r; }
                }
            }
        }
    } else {
        const float4 *src4 = reinterpret_cast<const float4 *>(src);
        for (int i = threadIdx.x; i < total4; i += kShBlock) {
            const float4 v = src4[i];
            const int e = i << 2;
            const float vv[4] = {v.x, v.y, v.z, v.w};
            int r = e / row_floats, col = e - r * row_floats;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lds[r * pitch + col] = vv[k];
                if (++col == row_floats) { col = 0; ++r; }
            }
        }
    }
    for (int i = (total4 << 2) + threadIdx.x; i < total; i += kShBlock) {
        const int r = i / row_floats, col = i - r * row_floats;
        lds[r * pitch + col] = src[i];
    }
}

// The coefficients come either split (features_dc [N,3] + features_rest [N,K-1,3], SPLIT) or as one
// [N,K,3] array.  (SPLIT is a template parameter: a struct of register arrays that is filled on one of
// two run-time paths is kept in scratch memory by the compiler.)
template <int KT, bool SPLIT>
struct StagedCoeffs {
    Staged<SPLIT ? 3 : KT * 3> a;            // features_dc, or the whole array
    Staged<(SPLIT && KT) ? KT * 3 - 3 : 0> b;  // features_rest
};

// element offset into a coefficient array stored as fp32 or (HALF) as IEEE half
template <bool HALF>
__device__ __forceinline__ const void *coeff_at(const void *base, int64_t elems) {
    return HALF ? static_cast<const void *>(reinterpret_cast<const __half *>(base) + elems)
                : static_cast<const void *>(reinterpret_cast<const float *>(base) + elems);
}

template <int KT, bool SPLIT, bool HALF = false>
__device__ __forceinline__ void coeffs_load(StagedCoeffs<KT, SPLIT> &st, const void *__restrict__ coeffs,
                                            const void *__restrict__ coeffs_rest, int n0, int rows, int K) {
    const int row_floats = K * 3;
    if (SPLIT) {
        stage_load<SPLIT ? 3 : KT * 3, HALF>(st.a, coeff_at<HALF>(coeffs, (int64_t)n0 * 3), rows);
        if (K > 1)
            stage_load<(SPLIT && KT) ? KT * 3 - 3 : 0, HALF>(st.b, coeff_at<HALF>(coeffs_rest, (int64_t)n0 * (row_floats - 3)), rows);
    } else {
        stage_load<SPLIT ? 3 : KT * 3, HALF>(st.a, coeff_at<HALF>(coeffs, (int64_t)n0 * row_floats), rows);
    }
}

template <int KT, bool SPLIT, bool HALF = false>
__device__ __forceinline__ void coeffs_park(const StagedCoeffs<KT, SPLIT> &st, const void *__restrict__ coeffs_v,
                                            const void *__restrict__ coeffs_rest_v, int n0, int rows, int K,
                                            float *lds) {
    const int row_floats = K * 3, pitch = row_floats + 1;
    if (HALF) {
        if (SPLIT) {
            stage_store_half(st.a, coeff_at<true>(coeffs_v, (int64_t)n0 * 3), rows, 3, lds, pitch, 0);
            if (K > 1)
                stage_store_half(st.b, coeff_at<true>(coeffs_rest_v, (int64_t)n0 * (row_floats - 3)), rows,
                                 row_floats - 3, lds, pitch, 3);
        } else {
            stage_store_half(st.a, coeff_at<true>(coeffs_v, (int64_t)n0 * row_floats), rows, row_floats, lds, pitch, 0);
        }
        return;
    }
    const float *coeffs = reinterpret_cast<const float *>(coeffs_v);
    const float *coeffs_rest = reinterpret_cast<const float *>(coeffs_rest_v);
    if (SPLIT) {
        stage_store(st.a, coeffs + (int64_t)n0 * 3, rows, 3, lds, pitch, 0);
        if (K > 1)
            stage_store(st.b, coeffs_rest + (int64_t)n0 * (row_floats - 3), rows, row_floats - 3, lds, pitch, 3);
    } else {
        stage_store(st.a, coeffs + (int64_t)n0 * row_floats, rows, row_floats, lds, pitch, 0);
    }
}

// The reverse: LDS rows back to a contiguous HBM span, 16 bytes per lane per store.
template <int RF>
__device__ __forceinline__ void unstage_rows(float *__restrict__ dst, int rows, int row_floats_rt,
                                             const float *lds, int pitch, int col_off) {
    const int row_floats = RF ? RF : row_floats_rt;
    const int total = rows * row_floats;
    lds += col_off;
    float4 *dst4 = reinterpret_cast<float4 *>(dst);
    const int total4 = total >> 2;
    for (int i = threadIdx.x; i < total4; i += kShBlock) {
        const int e = i << 2;
        int r = e / row_floats, col = e - r * row_floats;
        float vv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            vv[k] = lds[r * pitch + col];
            if (++col == row_floats) { col = 0; ++r; }
        }
        dst4[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    for (int i = (total4 << 2) + threadIdx.x; i < total; i += kShBlock) {
        const int r = i / row_floats, col = i - r * row_floats;
        dst[i] = lds[r * pitch + col];
    }
}

// KT > 0: the coefficient count per Gaussian is the compile-time constant KT (16 = degree-3 storage,
// FusionSense's configuration); KT = 0: any K <= kMaxK.
// PACK (single camera, depth channel): the kernel also writes what fsgs_live_pack_normals would — the 64-byte
// record per Gaussian the compositing kernels gather (centre, opacity, conic, colour + depth, camera-space normal)
// and normals_world — instead of the colour array: one launch and a round trip of the colours less.
struct ShPackArgs {
    const float *means2d, *conics, *opacities;
    const void *quats, *log_scales;  // fp32, or IEEE-half mirrors in the HALF instantiation
    const float *c2w;
    float4 *packed;
    float *normals_world, *zero_cells;
    int n_zero;
    const uint8_t *kept;  // nullable [N] (sh_fwd_pack_direct_body only): 0 = in no list, see ShPackRider
};

template <int KT, bool SPLIT, bool PACK, bool HALF = false>
__global__ void __launch_bounds__(kShBlock)
sh_fwd_kernel(int C, int N, int K, int degree, const float *__restrict__ means,
              const float *__restrict__ campos, const void *__restrict__ coeffs,
              const void *__restrict__ coeffs_rest, const int32_t *__restrict__ radii,
              const float *__restrict__ depths, float *__restrict__ colors_out, ShPackArgs pk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n0 = blockIdx.x * kShBlock;
    const int rows = min(kShBlock, N - n0);
    const int kk = (degree + 1) * (degree + 1);
    if (KT) K = KT;
    const int row_floats = K * 3;
    const int n = n0 + threadIdx.x;
    const bool in_range = n < N;
    // every global load of the first camera is in flight before the first wait.  (A persistent variant that
    // prefetches the next block's coefficients into registers while this one is evaluated measured no
    // faster: 20.8 vs 19.6 us at N = 300 k.)
    StagedCoeffs<KT, SPLIT> st;
    coeffs_load<KT, SPLIT, HALF>(st, coeffs, coeffs_rest, n0, rows, K);
    float mx = 0.f, myy = 0.f, mz = 0.f, dep0 = 0.f;
    int rad0 = 0;
    float4 pq = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 pxy = make_float2(0.f, 0.f);
    float pls[3] = {0.f, 0.f, 0.f}, pcon[3] = {0.f, 0.f, 0.f}, pop = 0.f;
    if (PACK && blockIdx.x == 0)  // e.g. the forward's running image maxima
        for (int k = threadIdx.x; k < pk.n_zero; k += kShBlock) pk.zero_cells[k] = 0.f;
    if (in_range) {
        mx = means[n * 3 + 0]; myy = means[n * 3 + 1]; mz = means[n * 3 + 2];
        rad0 = radii[n];
        if (depths) dep0 = depths[n];
        if (PACK) {
            pq = ld_attr4(pk.quats, n, HALF ? 1 : 0);
            pxy = reinterpret_cast<const float2 *>(pk.means2d)[n];
            pop = pk.opacities[n];
            ld_attr3(pk.log_scales, n, HALF ? 1 : 0, pls);
#pragma unroll
            for (int k = 0; k < 3; ++k) pcon[k] = pk.conics[n * 3 + k];
        }
    }
    coeffs_park<KT, SPLIT, HALF>(st, coeffs, coeffs_rest, n0, rows, K, lds);
    __syncthreads();
    if (!in_range) return;
    const float *my = lds + threadIdx.x * (row_floats + 1);
    const int D = depths ? 4 : 3;
    for (int c = 0; c < C; ++c) {
        const int64_t idx = (int64_t)c * N + n;
        float r = 0.5f, g = 0.5f, bl = 0.5f;
        if ((c ? radii[idx] : rad0) > 0) {
            float dx = mx - campos[c * 3 + 0], dy = myy - campos[c * 3 + 1], dz = mz - campos[c * 3 + 2];
            const float inorm = sh_inv_norm(dx, dy, dz);
            dx *= inorm; dy *= inorm; dz *= inorm;
            float b[kMaxK];
            sh_basis<false>(degree, dx, dy, dz, b, nullptr, nullptr, nullptr);
            float ar = 0.f, ag = 0.f, ab = 0.f;
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) {
                if (k < kk) {
                    ar = __builtin_fmaf(b[k], my[k * 3 + 0], ar);
                    ag = __builtin_fmaf(b[k], my[k * 3 + 1], ag);
                    ab = __builtin_fmaf(b[k], my[k * 3 + 2], ab);
                }
            }
            r = fmaxf(ar + 0.5f, 0.f);
            g = fmaxf(ag + 0.5f, 0.f);
            bl = fmaxf(ab + 0.5f, 0.f);
        }
        if (PACK) {
            // the Gaussian's camera-space normal (dn_model.py:618-636), as fsgs_live_pack_normals computes it
            const float mean[3] = {mx, myy, mz};
            NormalCtx o;
            normal_forward(pq, pls, mean, pk.c2w, o);
            float ex[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) pk.normals_world[n * 3 + k] = o.n[k];
            normal_to_camera(o.n, pk.c2w, ex);
            pk.packed[n * 4 + 0] = make_float4(pxy.x, pxy.y, pop, pcon[0]);
            pk.packed[n * 4 + 1] = make_float4(pcon[1], pcon[2], 0.f, 0.f);
            pk.packed[n * 4 + 2] = make_float4(r, g, bl, dep0);
            pk.packed[n * 4 + 3] = make_float4(ex[0], ex[1], ex[2], 0.f);
        } else if (D == 4) {
            reinterpret_cast<float4 *>(colors_out)[idx] = make_float4(r, g, bl, c ? depths[idx] : dep0);
        } else {
            colors_out[idx * 3 + 0] = r;
            colors_out[idx * 3 + 1] = g;
            colors_out[idx * 3 + 2] = bl;
        }
    }
}

template <bool SINGLE_CAM, int KT, bool SPLIT, bool HALF = false>
__global__ void __launch_bounds__(kShBlock)
sh_bwd_kernel(int C, int N, int K, int degree, const float *__restrict__ means,
              const float *__restrict__ campos, const void *__restrict__ coeffs,
              const void *__restrict__ coeffs_rest, const int32_t *__restrict__ radii, int D,
              const float *__restrict__ v_colors, float *__restrict__ v_coeffs,
              float *__restrict__ v_coeffs_rest, float *__restrict__ v_means,
              float *__restrict__ v_depths, int overwrite_means, float4 *__restrict__ v_rgb_masked,
              int64_t replica_rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n0 = blockIdx.x * kShBlock;
    const int rows = min(kShBlock, N - n0);
    const int kk = (degree + 1) * (degree + 1);
    if (KT) K = KT;
    const int row_floats = K * 3;
    const int pitch = row_floats + 1;
    const int n = n0 + threadIdx.x;
    StagedCoeffs<KT, SPLIT> st;
    coeffs_load<KT, SPLIT, HALF>(st, coeffs, coeffs_rest, n0, rows, K);
    // (single camera) this Gaussian's own inputs, in flight together with the coefficients
    float mx = 0.f, myy = 0.f, mz = 0.f, vc0 = 0.f, vc1 = 0.f, vc2 = 0.f, vc3 = 0.f;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    int rad0 = 0;
    if (SINGLE_CAM && n < N) {
        mx = means[n * 3 + 0]; myy = means[n * 3 + 1]; mz = means[n * 3 + 2];
        rad0 = radii[n];
        vc0 = v_colors[(int64_t)n * D + 0]; vc1 = v_colors[(int64_t)n * D + 1]; vc2 = v_colors[(int64_t)n * D + 2];
        if (D == 4 && v_depths) vc3 = v_colors[(int64_t)n * 4 + 3];
        if (replica_rows > 0 && rad0 >= kSpreadMinRadius) {
            // a large Gaussian's gradient atomics may have been spread over kGradReplicas lines (common.h): sum their
            // colour parts (the replicas of a Gaussian that was not spread hold zeros)
#pragma unroll
            for (int r = 1; r < kGradReplicas; ++r) {
                const float *line = v_colors + ((int64_t)r * replica_rows + n) * D;
                vc0 += line[0]; vc1 += line[1]; vc2 += line[2];
            }
        }
        if (!overwrite_means) { o0 = v_means[n * 3 + 0]; o1 = v_means[n * 3 + 1]; o2 = v_means[n * 3 + 2]; }
    }
    coeffs_park<KT, SPLIT, HALF>(st, coeffs, coeffs_rest, n0, rows, K, lds);
    __syncthreads();
    float *my = lds + threadIdx.x * pitch;
    if (SINGLE_CAM) {
      if (n < N) {
        // single-camera fast path (the training case): the coefficient gradient is formed in place in
        // this lane's LDS row, no per-lane accumulator array (75 VGPRs) is needed
        if (D == 4 && v_depths) v_depths[n] = vc3;
        float vr = 0.f, vg = 0.f, vb = 0.f;
        float b[kMaxK], bx[kMaxK], by[kMaxK], bz[kMaxK];
        float dx = 0.f, dy = 0.f, dz = 0.f, inorm = 0.f;
        const bool vis = rad0 > 0;
        if (vis) {
            dx = mx - campos[0]; dy = myy - campos[1]; dz = mz - campos[2];
            inorm = sh_inv_norm(dx, dy, dz);
            dx *= inorm; dy *= inorm; dz *= inorm;
            sh_basis<true>(degree, dx, dy, dz, b, bx, by, bz);
            float ar = 0.f, ag = 0.f, ab = 0.f;
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) {
                if (k < kk) {
                    ar = __builtin_fmaf(b[k], my[k * 3 + 0], ar);
                    ag = __builtin_fmaf(b[k], my[k * 3 + 1], ag);
                    ab = __builtin_fmaf(b[k], my[k * 3 + 2], ab);
                }
            }
            vr = (ar + 0.5f > 0.f) ? vc0 : 0.f;
            vg = (ag + 0.5f > 0.f) ? vc1 : 0.f;
            vb = (ab + 0.5f > 0.f) ? vc2 : 0.f;
        }
        float gdx = 0.f, gdy = 0.f, gdz = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxK; ++k) {
            if (k < K) {
                const bool on = vis && (k < kk);
                const float c0 = my[k * 3 + 0], c1 = my[k * 3 + 1], c2 = my[k * 3 + 2];
                if (on) {
                    const float w = c0 * vr + c1 * vg + c2 * vb;
                    gdx += bx[k] * w; gdy += by[k] * w; gdz += bz[k] * w;
                }
                my[k * 3 + 0] = on ? b[k] * vr : 0.f;
                my[k * 3 + 1] = on ? b[k] * vg : 0.f;
                my[k * 3 + 2] = on ? b[k] * vb : 0.f;
            }
        }
        const float dp = gdx * dx + gdy * dy + gdz * dz;
        const float g0 = (gdx - dp * dx) * inorm, g1 = (gdy - dp * dy) * inorm, g2 = (gdz - dp * dz) * inorm;
        v_means[n * 3 + 0] = o0 + g0;
        v_means[n * 3 + 1] = o1 + g1;
        v_means[n * 3 + 2] = o2 + g2;
        // data-parallel runs exchange this instead of the 48 coefficient gradients it determines
        // (basis(dir) x colour gradient, fsgs_sh_coeff_grad): 16 B instead of 192 B per Gaussian
        if (v_rgb_masked) {
            v_rgb_masked[n] = make_float4(vr, vg, vb, 0.f);
            if (n == 0) v_rgb_masked[N] = make_float4(campos[0], campos[1], campos[2], 0.f);  // row N: the camera centre
        }
      }
      if (v_rgb_masked) return;  // (no coefficient gradient is written)
    } else if (n < N) {
        const float mx = means[n * 3 + 0], myy = means[n * 3 + 1], mz = means[n * 3 + 2];
        float acc[kMaxK * 3];
#pragma unroll
        for (int k = 0; k < kMaxK * 3; ++k) acc[k] = 0.f;
        float gmx = 0.f, gmy = 0.f, gmz = 0.f;
        for (int c = 0; c < C; ++c) {
            const int64_t idx = (int64_t)c * N + n;
            if (D == 4 && v_depths) v_depths[idx] = v_colors[idx * 4 + 3];
            if (radii[idx] <= 0) continue;
            float dx = mx - campos[c * 3 + 0], dy = myy - campos[c * 3 + 1], dz = mz - campos[c * 3 + 2];
            const float inorm = sh_inv_norm(dx, dy, dz);
            dx *= inorm; dy *= inorm; dz *= inorm;
            float b[kMaxK], bx[kMaxK], by[kMaxK], bz[kMaxK];
            sh_basis<true>(degree, dx, dy, dz, b, bx, by, bz);
            float ar = 0.f, ag = 0.f, ab = 0.f;
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) {
                if (k < kk) {
                    ar = __builtin_fmaf(b[k], my[k * 3 + 0], ar);
                    ag = __builtin_fmaf(b[k], my[k * 3 + 1], ag);
                    ab = __builtin_fmaf(b[k], my[k * 3 + 2], ab);
                }
            }
            // clamp_min(colour + 0.5, 0): gradient passes only where the clamp is inactive
            const float vr = (ar + 0.5f > 0.f) ? v_colors[idx * D + 0] : 0.f;
            const float vg = (ag + 0.5f > 0.f) ? v_colors[idx * D + 1] : 0.f;
            const float vb = (ab + 0.5f > 0.f) ? v_colors[idx * D + 2] : 0.f;
            float gdx = 0.f, gdy = 0.f, gdz = 0.f;
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) {
                if (k < kk) {
                    acc[k * 3 + 0] += b[k] * vr;
                    acc[k * 3 + 1] += b[k] * vg;
                    acc[k * 3 + 2] += b[k] * vb;
                    const float w = my[k * 3 + 0] * vr + my[k * 3 + 1] * vg + my[k * 3 + 2] * vb;
                    gdx += bx[k] * w;
                    gdy += by[k] * w;
                    gdz += bz[k] * w;
                }
            }
            // through normalisation: v_dir = (g - (g.d) d) / |dir|
            const float dp = gdx * dx + gdy * dy + gdz * dz;
            gmx += (gdx - dp * dx) * inorm;
            gmy += (gdy - dp * dy) * inorm;
            gmz += (gdz - dp * dz) * inorm;
        }
        v_means[n * 3 + 0] = overwrite_means ? gmx : v_means[n * 3 + 0] + gmx;
        v_means[n * 3 + 1] = overwrite_means ? gmy : v_means[n * 3 + 1] + gmy;
        v_means[n * 3 + 2] = overwrite_means ? gmz : v_means[n * 3 + 2] + gmz;
        // reuse this lane's LDS row for the coefficient gradient, then stream it out coalesced
#pragma unroll
        for (int k = 0; k < kMaxK; ++k) {
            if (k < K) {
                my[k * 3 + 0] = (k < kk) ? acc[k * 3 + 0] : 0.f;
                my[k * 3 + 1] = (k < kk) ? acc[k * 3 + 1] : 0.f;
                my[k * 3 + 2] = (k < kk) ? acc[k * 3 + 2] : 0.f;
            }
        }
    }
    __syncthreads();
    if (SPLIT) {  // split gradient storage, mirroring the split inputs
        unstage_rows<3>(v_coeffs + (int64_t)n0 * 3, rows, 3, lds, pitch, 0);
        if (K > 1)
            unstage_rows<KT ? KT * 3 - 3 : 0>(v_coeffs_rest + (int64_t)n0 * (row_floats - 3), rows, row_floats - 3, lds,
                                              pitch, 3);
        return;
    }
    unstage_rows<KT * 3>(v_coeffs + (int64_t)n0 * row_floats, rows, row_floats, lds, pitch, 0);
}

// Coefficient gradients from their factors, for R views at once (data-parallel training: every rank all-gathers
// the ranks' masked colour gradients [N,4] + camera centre, 16 B per Gaussian and rank, instead of all-reducing the
// 192 B per Gaussian they determine):  v_coeffs[n,k,:] = scale * sum_r basis_k(dir_r(n)) * v_rgb_r[n,:].
// `gathered` = R blocks of (N + 1) float4: rows 0..N-1 = (v_r, v_g, v_b, -), row N = the rank's camera centre.
// The sum runs over r = 0..R-1 in this order on every rank: all replicas get bit-identical gradients.
// ADAM: the rebuilt gradients are not written out but applied: the Adam update of features_dc / features_rest runs
// on the rows as they leave LDS (same arithmetic as adam_kernel of glue.hip), saving the gradient's write + read.
struct ShAdamArgs {
    float *p_dc, *m_dc, *v_dc, *p_rest, *m_rest, *v_rest;
    float ss_dc, ss_rest, b1, b2, omb1, omb2, inv_sqrt_bc2, eps;
};

template <int RF>
__device__ __forceinline__ void adam_rows(float *__restrict__ P, float *__restrict__ M, float *__restrict__ V, int rows,
                                          int row_floats_rt, const float *lds, int pitch, int col_off, float ss,
                                          const ShAdamArgs &a) {
    const int row_floats = RF ? RF : row_floats_rt;
    const int total = rows * row_floats;
    lds += col_off;
    const int total4 = total >> 2;
    for (int i = threadIdx.x; i < total4; i += kShBlock) {
        const int e = i << 2;
        int r = e / row_floats, col = e - r * row_floats;
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g[k] = lds[r * pitch + col];
            if (++col == row_floats) { col = 0; ++r; }
        }
        float4 p = reinterpret_cast<float4 *>(P)[i], m = reinterpret_cast<float4 *>(M)[i], v = reinterpret_cast<float4 *>(V)[i];
        adam_one(p.x, g[0], m.x, v.x, a.b1, a.b2, a.omb1, a.omb2, ss, a.inv_sqrt_bc2, a.eps);
        adam_one(p.y, g[1], m.y, v.y, a.b1, a.b2, a.omb1, a.omb2, ss, a.inv_sqrt_bc2, a.eps);
        adam_one(p.z, g[2], m.z, v.z, a.b1, a.b2, a.omb1, a.omb2, ss, a.inv_sqrt_bc2, a.eps);
        adam_one(p.w, g[3], m.w, v.w, a.b1, a.b2, a.omb1, a.omb2, ss, a.inv_sqrt_bc2, a.eps);
        reinterpret_cast<float4 *>(P)[i] = p; reinterpret_cast<float4 *>(M)[i] = m; reinterpret_cast<float4 *>(V)[i] = v;
    }
    for (int i = (total4 << 2) + threadIdx.x; i < total; i += kShBlock) {
        const int r = i / row_floats, col = i - r * row_floats;
        float p = P[i], m = M[i], v = V[i];
        adam_one(p, lds[r * pitch + col], m, v, a.b1, a.b2, a.omb1, a.omb2, ss, a.inv_sqrt_bc2, a.eps);
        P[i] = p; M[i] = m; V[i] = v;
    }
}

// the same on a group of a GaussAdam block (fsgs_gauss_sh_bwd_adam), with the group's half mirror if it has one; U
// float4 triples (parameter, both moments) are in flight per thread before the first store.  SKIP (GaussAdam::skip_idle):
// the moments are read first and the parameter only where something changes — a second, dependent round trip; without
// it all three streams are read at once and every element is stepped (identical parameters: a skipped update is a no-op).
template <int RF, int U, bool SKIP>
__device__ __forceinline__ void adam_rows_g(const GaussAdam &a, int grp, int64_t first, int rows, const float *lds,
                                            int pitch, int col_off) {
    float4 *__restrict__ const P4 = reinterpret_cast<float4 *>(a.p[grp] + first);
    float4 *__restrict__ const M4 = reinterpret_cast<float4 *>(a.m[grp] + first);
    float4 *__restrict__ const V4 = reinterpret_cast<float4 *>(a.v[grp] + first);
    float *const P = a.p[grp] + first, *const M = a.m[grp] + first, *const V = a.v[grp] + first;
    __half *const H = a.h[grp] ? a.h[grp] + first : nullptr;
    const float ss = a.ss[grp];
    const int total = rows * RF;
    lds += col_off;
    const int total4 = total >> 2;
    for (int i0 = threadIdx.x; i0 < total4; i0 += kShBlock * U) {
        float4 p[U], m[U], v[U];
        float g[U][4];
        bool idle[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kShBlock;
            idle[u] = true;
            if (i < total4) {
                // (non-temporal: each element of the three streams is read once and written once per frame, and at 6 M
                // Gaussians they are 3.5 GB — kept out of the L2 / MALL they stop evicting the gradient lines this launch
                // gathers: 2.14-2.18 ms -> 1.65-1.92 ms at config #4, four runs each; no change at 300 k)
                m[u] = nt_load_f4(&M4[i]); v[u] = nt_load_f4(&V4[i]);
                if (!SKIP) p[u] = nt_load_f4(&P4[i]);
                const int e = i << 2;
                int r = e / RF, col = e - r * RF;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    g[u][k] = lds[r * pitch + col];
                    if (++col == RF) { col = 0; ++r; }
                }
            }
        }
        // SKIP: zero gradients on zero moments (Gaussians no view has reached yet): nothing changes — the parameters are
        // not read and nothing is written (32 instead of 96 B per float4)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kShBlock;
            if (i < total4) {
                if (SKIP) {
                    auto z4 = [](const float4 q) { return q.x == 0.f && q.y == 0.f && q.z == 0.f && q.w == 0.f; };
                    idle[u] = z4(m[u]) && z4(v[u]) && g[u][0] == 0.f && g[u][1] == 0.f && g[u][2] == 0.f && g[u][3] == 0.f;
                    if (!idle[u]) p[u] = P4[i];
                } else {
                    idle[u] = false;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kShBlock;
            if (i >= total4) break;
            if (idle[u]) continue;
            adam_one(p[u].x, g[u][0], m[u].x, v[u].x, a.b1, a.b2, a.omb1, a.omb2, ss, a.isb2, a.eps);
            adam_one(p[u].y, g[u][1], m[u].y, v[u].y, a.b1, a.b2, a.omb1, a.omb2, ss, a.isb2, a.eps);
            adam_one(p[u].z, g[u][2], m[u].z, v[u].z, a.b1, a.b2, a.omb1, a.omb2, ss, a.isb2, a.eps);
            adam_one(p[u].w, g[u][3], m[u].w, v[u].w, a.b1, a.b2, a.omb1, a.omb2, ss, a.isb2, a.eps);
            nt_store_f4(p[u], &P4[i]); nt_store_f4(m[u], &M4[i]); nt_store_f4(v[u], &V4[i]);
            if (H) {
                const __half2 lo = __floats2half2_rn(p[u].x, p[u].y), hi = __floats2half2_rn(p[u].z, p[u].w);
                uint2 o;
                o.x = *reinterpret_cast<const unsigned *>(&lo);
                o.y = *reinterpret_cast<const unsigned *>(&hi);
                reinterpret_cast<uint2 *>(H)[i] = o;
            }
        }
    }
    for (int i = (total4 << 2) + threadIdx.x; i < total; i += kShBlock) {
        const int r = i / RF, col = i - r * RF;
        float p = P[i], m = M[i], v = V[i];
        adam_one(p, lds[r * pitch + col], m, v, a.b1, a.b2, a.omb1, a.omb2, ss, a.isb2, a.eps);
        P[i] = p; M[i] = m; V[i] = v;
        if (H) H[i] = __float2half(p);
    }
}

template <int KT, bool ADAM>
__global__ void __launch_bounds__(kShBlock)
sh_coeff_grad_kernel(int R, int N, int K, int degree, const float *__restrict__ means,
                     const float4 *__restrict__ gathered, float scale, float *__restrict__ v_dc,
                     float *__restrict__ v_rest, ShAdamArgs ad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n0 = blockIdx.x * kShBlock;
    const int rows = min(kShBlock, N - n0);
    const int kk = (degree + 1) * (degree + 1);
    if (KT) K = KT;
    const int row_floats = K * 3, pitch = row_floats + 1;
    const int n = n0 + threadIdx.x;
    if (n < N) {
        const float mx = means[n * 3 + 0], myy = means[n * 3 + 1], mz = means[n * 3 + 2];
        float acc[kMaxK * 3];
#pragma unroll
        for (int k = 0; k < kMaxK * 3; ++k) acc[k] = 0.f;
        for (int r = 0; r < R; ++r) {
            const float4 *blk = gathered + (int64_t)r * (N + 1);
            const float4 v = blk[n];
            if (v.x == 0.f && v.y == 0.f && v.z == 0.f) continue;  // invisible / clamped / no gradient in this view
            const float4 cp = blk[N];
            float dx = mx - cp.x, dy = myy - cp.y, dz = mz - cp.z;
            const float inorm = sh_inv_norm(dx, dy, dz);
            dx *= inorm; dy *= inorm; dz *= inorm;
            float b[kMaxK];
            sh_basis<false>(degree, dx, dy, dz, b, nullptr, nullptr, nullptr);
#pragma unroll
            for (int k = 0; k < kMaxK; ++k) {
                if (k < kk) {
                    acc[k * 3 + 0] += b[k] * v.x;
                    acc[k * 3 + 1] += b[k] * v.y;
                    acc[k * 3 + 2] += b[k] * v.z;
                }
            }
        }
        float *my = lds + threadIdx.x * pitch;
#pragma unroll
        for (int k = 0; k < kMaxK; ++k) {
            if (k < K) {
                my[k * 3 + 0] = (k < kk) ? acc[k * 3 + 0] * scale : 0.f;
                my[k * 3 + 1] = (k < kk) ? acc[k * 3 + 1] * scale : 0.f;
                my[k * 3 + 2] = (k < kk) ? acc[k * 3 + 2] * scale : 0.f;
            }
        }
    }
    __syncthreads();
    if (ADAM) {
        adam_rows<3>(ad.p_dc + (int64_t)n0 * 3, ad.m_dc + (int64_t)n0 * 3, ad.v_dc + (int64_t)n0 * 3, rows, 3, lds, pitch, 0,
                     ad.ss_dc, ad);
        if (K > 1) {
            const int64_t o = (int64_t)n0 * (row_floats - 3);
            adam_rows<KT ? KT * 3 - 3 : 0>(ad.p_rest + o, ad.m_rest + o, ad.v_rest + o, rows, row_floats - 3, lds, pitch, 3,
                                           ad.ss_rest, ad);
        }
        return;
    }
    unstage_rows<3>(v_dc + (int64_t)n0 * 3, rows, 3, lds, pitch, 0);
    if (K > 1)
        unstage_rows<KT ? KT * 3 - 3 : 0>(v_rest + (int64_t)n0 * (row_floats - 3), rows, row_floats - 3, lds, pitch, 3);
}

// Camera centres from world-to-camera matrices: campos = -A^-1 t for [A t; 0 1] (what
// torch.inverse(viewmats)[:, :3, 3] yields in gsplat's rasterization()), one thread per camera,
// no solver library and no host synchronisation.
__global__ void campos_kernel(int C, const float *__restrict__ viewmats, float *__restrict__ campos) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float *m = viewmats + c * 16;
    const float a = m[0], b = m[1], cc = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const float tx = m[3], ty = m[7], tz = m[11];
    const float A = e * i - f * h, B = -(d * i - f * g), Cc = d * h - e * g;
    const float det = a * A + b * B + cc * Cc;
    const float id = 1.f / det;
    // inverse = adj / det ; rows of the inverse:
    const float i00 = A * id, i01 = -(b * i - cc * h) * id, i02 = (b * f - cc * e) * id;
    const float i10 = B * id, i11 = (a * i - cc * g) * id, i12 = -(a * f - cc * d) * id;
    const float i20 = Cc * id, i21 = -(a * h - b * g) * id, i22 = (a * e - b * d) * id;
    campos[c * 3 + 0] = -(i00 * tx + i01 * ty + i02 * tz);
    campos[c * 3 + 1] = -(i10 * tx + i11 * ty + i12 * tz);
    campos[c * 3 + 2] = -(i20 * tx + i21 * ty + i22 * tz);
}

}  // namespace fsgs

using namespace fsgs;

extern "C" int fsgs_campos_from_viewmats(int C, const float *viewmats, float *campos, fsgs_stream_t stream) {
    if (C < 0) return FSGS_EINVAL;
    if (C == 0) return FSGS_OK;
    if (!viewmats || !campos) return FSGS_EINVAL;
    hipLaunchKernelGGL(campos_kernel, dim3(ceil_div(C, 64)), dim3(64), 0, as_stream(stream), C, viewmats, campos);
    return check_launch();
}

static int sh_fwd_impl(int C, int N, int K, int degree, const float *means, const float *campos,
                       const float *coeffs, const float *coeffs_rest, const int32_t *radii, const float *depths,
                       float *colors_out, fsgs_stream_t stream) {
    if (C < 0 || N < 0 || degree < 0 || degree > 4 || K < (degree + 1) * (degree + 1) || K > kMaxK)
        return FSGS_EINVAL;
    if ((int64_t)C * N == 0) return FSGS_OK;
    if (!means || !campos || !coeffs || !radii || !colors_out) return FSGS_EINVAL;
    const size_t lds_bytes = (size_t)kShBlock * (K * 3 + 1) * sizeof(float);
    ShPackArgs pk = {};
#define FSGS_SH_FWD(KT, SP)                                                                               \
    hipLaunchKernelGGL((sh_fwd_kernel<KT, SP, false>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes, \
                       as_stream(stream), C, N, K, degree, means, campos, coeffs, coeffs_rest, radii, depths, \
                       colors_out, pk)
    if (coeffs_rest) {
        if (K == 16) FSGS_SH_FWD(16, true); else FSGS_SH_FWD(0, true);
    } else {
        if (K == 16) FSGS_SH_FWD(16, false); else FSGS_SH_FWD(0, false);
    }
#undef FSGS_SH_FWD
    return check_launch();
}

extern "C" int fsgs_sh_fwd(int C, int N, int K, int degree, const float *means, const float *campos,
                           const float *coeffs, const int32_t *radii, const float *depths,
                           float *colors_out, fsgs_stream_t stream) {
    return sh_fwd_impl(C, N, K, degree, means, campos, coeffs, nullptr, radii, depths, colors_out, stream);
}

extern "C" int fsgs_sh_fwd_split(int C, int N, int K, int degree, const float *means, const float *campos,
                                 const float *features_dc, const float *features_rest, const int32_t *radii,
                                 const float *depths, float *colors_out, fsgs_stream_t stream) {
    if (C >= 0 && N >= 0 && (int64_t)C * N == 0) return FSGS_OK;  // (empty tensors have no address)
    if (!features_rest && K > 1) return FSGS_EINVAL;
    return sh_fwd_impl(C, N, K, degree, means, campos, features_dc, features_rest ? features_rest : features_dc,
                       radii, depths, colors_out, stream);
}

// One Gaussian's 16 stored coefficients (split storage: dc [N,3], rest [N,15,3]) straight into registers.
// fp32: sixteen 12-byte pieces (merged into 16-byte loads by the compiler).  HALF (IEEE-half mirrors, rows of 6 and
// 90 bytes, i.e. only 2-byte aligned): the aligned dwords covering the row are loaded and the halves picked by the
// row's parity — 2 + 23 dword loads instead of 48 two-byte ones.  (The 2 bytes before an odd row belong to the row in
// front of it; the 2 bytes after an even row to the next one, except for the LAST row of the tensor, whose final
// dword is read as the 2 bytes that exist.)
struct ShF3 { float x, y, z; };
template <bool HALF>
__device__ __forceinline__ void load_sh_row(const void *__restrict__ dc, const void *__restrict__ rest, int64_t n,
                                            int64_t N, ShF3 (&cf)[16]) {
    if (!HALF) {
        cf[0] = reinterpret_cast<const ShF3 *>(dc)[n];
        const ShF3 *row = reinterpret_cast<const ShF3 *>(rest) + n * 15;
#pragma unroll
        for (int k = 0; k < 15; ++k) cf[k + 1] = row[k];
        return;
    }
    const int par = (int)(n & 1);  // rows of an odd number of halves: odd rows start in the upper half of a dword
    auto pick = [](const uint32_t *w, int i) {  // half i of the dword array (i static after unrolling)
        const uint32_t v = w[i >> 1];
        return __half2float(__ushort_as_half((unsigned short)((i & 1) ? (v >> 16) : (v & 0xFFFFu))));
    };
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(dc) + ((n * 3) >> 1);
        const bool last_even = !par && n == N - 1;
        const uint32_t w[2] = {src[0], last_even ? (uint32_t) * reinterpret_cast<const unsigned short *>(src + 1) : src[1]};
        float f[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) f[j] = par ? pick(w, j + 1) : pick(w, j);
        cf[0].x = f[0]; cf[0].y = f[1]; cf[0].z = f[2];
    }
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(rest) + ((n * 45) >> 1);
        uint32_t w[23];
#pragma unroll
        for (int k = 0; k < 22; ++k) w[k] = src[k];
        w[22] = (!par && n == N - 1) ? (uint32_t) * reinterpret_cast<const unsigned short *>(src + 22) : src[22];
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            float f[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) f[c] = par ? pick(w, 3 * k + c + 1) : pick(w, 3 * k + c);
            cf[k + 1].x = f[0]; cf[k + 1].y = f[1]; cf[k + 1].z = f[2];
        }
    }
}

// The single-camera degree-3-storage backward of the training path: the coefficients are read straight into
// registers (as in sh_fwd_pack_direct_kernel) and only the coefficient GRADIENT goes through LDS, for coalesced
// 16-byte stores — one staging pass and one barrier less than sh_bwd_kernel: 32 us against 36 us at 300 k Gaussians
// (writing the rows directly in 12-byte pieces as well: 39 us).
__global__ void __launch_bounds__(kShBlock)
sh_bwd_hybrid_kernel(int N, int degree, const float *__restrict__ means, const float *__restrict__ campos,
                     const float *__restrict__ dc, const float *__restrict__ rest, const int32_t *__restrict__ radii,
                     int D, const float *__restrict__ v_colors, float *__restrict__ v_dc, float *__restrict__ v_rest,
                     float *__restrict__ v_means, float *__restrict__ v_depths, int overwrite_means,
                     float4 *__restrict__ v_rgb_masked, int64_t replica_rows) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int pitch = 49;
    const int n0 = blockIdx.x * kShBlock;
    const int rows = min(kShBlock, N - n0);
    const int n = n0 + threadIdx.x;
    float *my = lds + threadIdx.x * pitch;
    if (n < N) {
        struct F3 { float x, y, z; };
        F3 cf[16];
        cf[0] = reinterpret_cast<const F3 *>(dc)[n];
        const F3 *row = reinterpret_cast<const F3 *>(rest) + (int64_t)n * 15;
#pragma unroll
        for (int k = 0; k < 15; ++k) cf[k + 1] = row[k];
        const float mx = means[n * 3 + 0], myy = means[n * 3 + 1], mz = means[n * 3 + 2];
        const int rad0 = radii[n];
        float vc0 = v_colors[(int64_t)n * D + 0], vc1 = v_colors[(int64_t)n * D + 1], vc2 = v_colors[(int64_t)n * D + 2];
        if (D == 4 && v_depths) v_depths[n] = v_colors[(int64_t)n * 4 + 3];
        if (replica_rows > 0 && rad0 >= kSpreadMinRadius) {
#pragma unroll
            for (int r = 1; r < kGradReplicas; ++r) {
                const float *line = v_colors + ((int64_t)r * replica_rows + n) * D;
                vc0 += line[0]; vc1 += line[1]; vc2 += line[2];
            }
        }
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        if (!overwrite_means) { o0 = v_means[n * 3 + 0]; o1 = v_means[n * 3 + 1]; o2 = v_means[n * 3 + 2]; }
        const int kk = (degree + 1) * (degree + 1);
        float vr = 0.f, vg = 0.f, vb = 0.f;
        float b[kMaxK], bx[kMaxK], by[kMaxK], bz[kMaxK];
        float dx = 0.f, dy = 0.f, dz = 0.f, inorm = 0.f;
        const bool vis = rad0 > 0;
        if (vis) {
            dx = mx - campos[0]; dy = myy - campos[1]; dz = mz - campos[2];
            inorm = sh_inv_norm(dx, dy, dz);
            dx *= inorm; dy *= inorm; dz *= inorm;
            sh_basis<true>(degree, dx, dy, dz, b, bx, by, bz);
            float ar = 0.f, ag = 0.f, ab = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < kk) { ar = __builtin_fmaf(b[k], cf[k].x, ar); ag = __builtin_fmaf(b[k], cf[k].y, ag); ab = __builtin_fmaf(b[k], cf[k].z, ab); }
            }
            vr = (ar + 0.5f > 0.f) ? vc0 : 0.f;
            vg = (ag + 0.5f > 0.f) ? vc1 : 0.f;
            vb = (ab + 0.5f > 0.f) ? vc2 : 0.f;
        }
        float gdx = 0.f, gdy = 0.f, gdz = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const bool on = vis && (k < kk);
            if (on) {
                const float w = cf[k].x * vr + cf[k].y * vg + cf[k].z * vb;
                gdx += bx[k] * w; gdy += by[k] * w; gdz += bz[k] * w;
            }
            if (!v_rgb_masked) {
                my[k * 3 + 0] = on ? b[k] * vr : 0.f;
                my[k * 3 + 1] = on ? b[k] * vg : 0.f;
                my[k * 3 + 2] = on ? b[k] * vb : 0.f;
            }
        }
        const float dp = gdx * dx + gdy * dy + gdz * dz;
        v_means[n * 3 + 0] = o0 + (gdx - dp * dx) * inorm;
        v_means[n * 3 + 1] = o1 + (gdy - dp * dy) * inorm;
        v_means[n * 3 + 2] = o2 + (gdz - dp * dz) * inorm;
        if (v_rgb_masked) {
            v_rgb_masked[n] = make_float4(vr, vg, vb, 0.f);
            if (n == 0) v_rgb_masked[N] = make_float4(campos[0], campos[1], campos[2], 0.f);
        }
    }
    if (v_rgb_masked) return;
    __syncthreads();
    unstage_rows<3>(v_dc + (int64_t)n0 * 3, rows, 3, lds, pitch, 0);
    unstage_rows<45>(v_rest + (int64_t)n0 * 45, rows, 45, lds, pitch, 3);
}

// The whole per-Gaussian backward of the training path in ONE launch: the SH backward (as sh_bwd_hybrid_kernel) and
// fsgs_gaussian_bwd's body (gauss_bwd_body.h) in the same thread — the packed gradient line is read (folded, cleared)
// once, the view-direction share of v_means stays in registers, one launch boundary less.
struct GaussShArgs {
    const float *means, *campos;
    const void *dc, *rest;
    const void *quats;
    const float *scales, *viewmat, *K;
    const int32_t *radii;
    const float *conics;
    float *v_dc, *v_rest, *v_means, *v_quats, *v_scales;
    float4 *v_rgb_masked;
    int N, degree, width, height;
    float eps2d;
    // dense scenes: the coefficients (192 of ~320 B read per Gaussian) are fetched only AFTER the gradient line, and
    // only by the Gaussians whose line carries a colour gradient — a dependent load, so only where most lines are
    // empty (large N: config #4 composites 2 % of its pairs); the outputs are the same numbers either way
    int lazy_sh;
};

#ifndef FSGS_ADAM_ROWS_U
#define FSGS_ADAM_ROWS_U 6
#endif
constexpr int kAdamRowsU = FSGS_ADAM_ROWS_U;  // float4 triples in flight per thread in the cooperative Adam of the features
template <bool HALF, bool ADAM>
__global__ void __launch_bounds__(kShBlock) __attribute__((amdgpu_waves_per_eu(3, 3)))
gauss_sh_bwd_kernel(GaussShArgs A, GaussBwdFused fz) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int pitch = 49;
    const int N = A.N;
    const int n0 = blockIdx.x * kShBlock;
    const int rows = min(kShBlock, N - n0);
    const int n = n0 + threadIdx.x;
    float *my = lds + threadIdx.x * pitch;
    if (n < N) {
        ShF3 cf[16];
        if (!A.lazy_sh) load_sh_row<HALF>(A.dc, A.rest, n, N, cf);
        const float mx = A.means[n * 3 + 0], myy = A.means[n * 3 + 1], mz = A.means[n * 3 + 2];
        const int rad0 = A.radii[n];
        float4 pa, pb, pc, pd;
        gauss_bwd_load_line(n, fz, A.radii, A.conics, pa, pb, pc, pd);
        // (lazy: a Gaussian without colour gradient has vr = vg = vb = 0 below whatever its coefficients are: every
        // coefficient gradient and the view-direction share of v_means are zero)
        const bool coloured = rad0 > 0 && (pa.x != 0.f || pa.y != 0.f || pa.z != 0.f);
        if (A.lazy_sh && coloured) load_sh_row<HALF>(A.dc, A.rest, n, N, cf);
        const int kk = (A.degree + 1) * (A.degree + 1);
        float vr = 0.f, vg = 0.f, vb = 0.f;
        float share[3] = {0.f, 0.f, 0.f};
        {
            float b[kMaxK], bx[kMaxK], by[kMaxK], bz[kMaxK];
            float dx = 0.f, dy = 0.f, dz = 0.f, inorm = 0.f;
            const bool vis = A.lazy_sh ? coloured : rad0 > 0;
            if (vis) {
                dx = mx - A.campos[0]; dy = myy - A.campos[1]; dz = mz - A.campos[2];
                inorm = sh_inv_norm(dx, dy, dz);
                dx *= inorm; dy *= inorm; dz *= inorm;
                sh_basis<true>(A.degree, dx, dy, dz, b, bx, by, bz);
                float ar = 0.f, ag = 0.f, ab = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k < kk) { ar = __builtin_fmaf(b[k], cf[k].x, ar); ag = __builtin_fmaf(b[k], cf[k].y, ag); ab = __builtin_fmaf(b[k], cf[k].z, ab); }
                }
                vr = (ar + 0.5f > 0.f) ? pa.x : 0.f;
                vg = (ag + 0.5f > 0.f) ? pa.y : 0.f;
                vb = (ab + 0.5f > 0.f) ? pa.z : 0.f;
            }
            float gdx = 0.f, gdy = 0.f, gdz = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool on = vis && (k < kk);
                if (on) {
                    const float w = cf[k].x * vr + cf[k].y * vg + cf[k].z * vb;
                    gdx += bx[k] * w; gdy += by[k] * w; gdz += bz[k] * w;
                }
                if (!A.v_rgb_masked) {
                    my[k * 3 + 0] = on ? b[k] * vr : 0.f;
                    my[k * 3 + 1] = on ? b[k] * vg : 0.f;
                    my[k * 3 + 2] = on ? b[k] * vb : 0.f;
                }
            }
            const float dp = gdx * dx + gdy * dy + gdz * dz;
            share[0] = (gdx - dp * dx) * inorm; share[1] = (gdy - dp * dy) * inorm; share[2] = (gdz - dp * dz) * inorm;
        }
        if (A.v_rgb_masked) {
            A.v_rgb_masked[n] = make_float4(vr, vg, vb, 0.f);
            if (n == 0) A.v_rgb_masked[N] = make_float4(A.campos[0], A.campos[1], A.campos[2], 0.f);
        }
        gaussian_bwd_one<true, ADAM>(n, 1, N, A.means, A.quats, A.scales, A.viewmat, A.K, A.width, A.height, A.eps2d,
                                     A.radii, A.conics, nullptr, nullptr, nullptr, nullptr, nullptr, A.v_means, A.v_quats,
                                     A.v_scales, nullptr, fz, pa, pb, pc, pd, share);
    }
    if (A.v_rgb_masked) return;
    __syncthreads();
    if (ADAM) {  // (every thread of the block has read its coefficients: the block's rows may move)
        if (fz.adam.skip_idle) {
            adam_rows_g<3, 4, true>(fz.adam, 3, (int64_t)n0 * 3, rows, lds, pitch, 0);
            adam_rows_g<45, 4, true>(fz.adam, 4, (int64_t)n0 * 45, rows, lds, pitch, 3);
        } else {
            adam_rows_g<3, 1, false>(fz.adam, 3, (int64_t)n0 * 3, rows, lds, pitch, 0);
            adam_rows_g<45, kAdamRowsU, false>(fz.adam, 4, (int64_t)n0 * 45, rows, lds, pitch, 3);
        }
        return;
    }
    unstage_rows<3>(A.v_dc + (int64_t)n0 * 3, rows, 3, lds, pitch, 0);
    unstage_rows<45>(A.v_rest + (int64_t)n0 * 45, rows, 45, lds, pitch, 3);
}

// DEPRECATED process-wide default for callers that pass flags < 0 (round-3 behaviour): from this many Gaussians on,
// fsgs_gauss_sh_bwd fetches coefficients lazily and skips idle Adam elements.  The flags argument decides per launch.
static std::atomic<int> g_lazy_sh_min_n{1 << 20};
extern "C" int fsgs_set_lazy_sh_min_n(int n) { return g_lazy_sh_min_n.exchange(n); }

static int gauss_sh_bwd_impl(int N, int degree, const float *means, const float *campos, const void *features_dc,
                             const void *features_rest, const void *quats, const void *log_scales, int attr_half,
                             const float *scales, const float *opac, const float *viewmat, const float *K,
                             const float *c2w, int width, int height, float eps2d, const int32_t *radii,
                             const float *conics, float *v_packed, float *v_features_dc, float *v_features_rest,
                             float *v_rgb_masked, float *v_means, float *v_quats, float *v_log_scales,
                             float *v_opac_logit, float *absgrad, float *xys_grad_norm, float *vis_counts,
                             float *max_2Dsize, float inv_max_hw, const uint8_t *frozen, int64_t replica_rows,
                             fsgs_stream_t stream, const fsgs_adam_groups *adam = nullptr, float min_scale_g = 0.f,
                             int flags = -1) {
    if (N < 0 || degree < 0 || degree > 3 || replica_rows < 0) return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!means || !campos || !features_dc || !features_rest || !quats || !log_scales || !scales || !opac || !viewmat ||
        !K || !c2w || !radii || !conics || !v_packed || !absgrad)
        return FSGS_EINVAL;
    if (!adam && (!v_means || !v_quats || !v_log_scales || !v_opac_logit)) return FSGS_EINVAL;
    if (!adam && !v_rgb_masked && (!v_features_dc || !v_features_rest)) return FSGS_EINVAL;
    GaussAdam ga{};
    if (adam) {
        // the six groups in the trainer's order (fsgs.h); the gradients are applied, not written
        static const int64_t kRow[6] = {3, 3, 4, 3, 45, 1};
        if (v_rgb_masked || adam->n_groups != 6 || adam->step < 1) return FSGS_EINVAL;
        const double bc1 = 1.0 - pow(adam->beta1, (double)adam->step), bc2 = 1.0 - pow(adam->beta2, (double)adam->step);
        for (int g = 0; g < 6; ++g) {
            if (!adam->params[g] || !adam->exp_avg[g] || !adam->exp_avg_sq[g] || adam->numel[g] != kRow[g] * N)
                return FSGS_EINVAL;
            ga.p[g] = adam->params[g]; ga.m[g] = adam->exp_avg[g]; ga.v[g] = adam->exp_avg_sq[g];
            ga.h[g] = reinterpret_cast<__half *>(adam->half_mirror[g]);
            ga.ss[g] = (float)((double)adam->lr[g] / bc1);  // (as build_adam_args: adam_body.h)
        }
        ga.b1 = (float)adam->beta1; ga.b2 = (float)adam->beta2; ga.isb2 = (float)(1.0 / sqrt(bc2)); ga.eps = adam->eps;
        ga.omb1 = (float)(1.0 - adam->beta1); ga.omb2 = (float)(1.0 - adam->beta2);
        ga.min_scale_g = min_scale_g;
        ga.on = 1;
    }
    if ((xys_grad_norm || vis_counts || max_2Dsize) && !(xys_grad_norm && vis_counts && max_2Dsize)) return FSGS_EINVAL;
    GaussShArgs A = {means, campos, features_dc, features_rest, quats, scales, viewmat, K, radii, conics,
                     v_features_dc, v_features_rest, v_means, v_quats, v_log_scales,
                     reinterpret_cast<float4 *>(v_rgb_masked), N, degree, width, height, eps2d, 0};
    // flags >= 0: FSGS_GSB_LAZY_COEFFS (1) | FSGS_GSB_SKIP_IDLE (2) as the caller decided for THIS launch; flags < 0
    // (deprecated): the process-wide threshold of fsgs_set_lazy_sh_min_n decides both from N
    if (flags < 0) flags = N >= g_lazy_sh_min_n.load(std::memory_order_relaxed) ? 3 : 0;
    A.lazy_sh = (flags & 1) ? 1 : 0;
    ga.skip_idle = (flags & 2) ? 1 : 0;
    GaussBwdFused fz{};
    fz.v_packed = reinterpret_cast<float4 *>(v_packed);
    fz.log_scales = log_scales; fz.opac = opac; fz.c2w = c2w; fz.absgrad = absgrad; fz.v_opac_logit = v_opac_logit;
    fz.accumulate_means = 0; fz.frozen = frozen; fz.attr_half = attr_half; fz.replica_rows = replica_rows;
    fz.xys_grad_norm = xys_grad_norm; fz.vis_counts = vis_counts; fz.max_2Dsize = max_2Dsize; fz.inv_max_hw = inv_max_hw;
    fz.adam = ga;
    const size_t lds_bytes = (size_t)kShBlock * 49 * sizeof(float);
    const dim3 grid(ceil_div(N, kShBlock)), block(kShBlock);
    hipStream_t s = as_stream(stream);
    if (attr_half && ga.on) hipLaunchKernelGGL((gauss_sh_bwd_kernel<true, true>), grid, block, lds_bytes, s, A, fz);
    else if (attr_half) hipLaunchKernelGGL((gauss_sh_bwd_kernel<true, false>), grid, block, lds_bytes, s, A, fz);
    else if (ga.on) hipLaunchKernelGGL((gauss_sh_bwd_kernel<false, true>), grid, block, lds_bytes, s, A, fz);
    else hipLaunchKernelGGL((gauss_sh_bwd_kernel<false, false>), grid, block, lds_bytes, s, A, fz);
    return check_launch();
}

// fsgs_sh_bwd_split / fsgs_sh_bwd_colors (one camera, 16 stored coefficients, D = 16 packed records) followed by
// fsgs_gaussian_bwd, in one launch.  v_features_dc / v_features_rest, or v_rgb_masked [N + 1,4] (then the two
// coefficient outputs are not written); the other arguments as in the two calls it replaces.
extern "C" int fsgs_gauss_sh_bwd(int N, int degree, const float *means, const float *campos, const float *features_dc,
                                 const float *features_rest, const float *quats, const float *log_scales,
                                 const float *scales, const float *opac, const float *viewmat, const float *K,
                                 const float *c2w, int width, int height, float eps2d, const int32_t *radii,
                                 const float *conics, float *v_packed, float *v_features_dc, float *v_features_rest,
                                 float *v_rgb_masked, float *v_means, float *v_quats, float *v_log_scales,
                                 float *v_opac_logit, float *absgrad, float *xys_grad_norm, float *vis_counts,
                                 float *max_2Dsize, float inv_max_hw, const uint8_t *frozen, int64_t replica_rows,
                                 int flags, fsgs_stream_t stream) {
    return gauss_sh_bwd_impl(N, degree, means, campos, features_dc, features_rest, quats, log_scales, 0, scales, opac,
                             viewmat, K, c2w, width, height, eps2d, radii, conics, v_packed, v_features_dc,
                             v_features_rest, v_rgb_masked, v_means, v_quats, v_log_scales, v_opac_logit, absgrad,
                             xys_grad_norm, vis_counts, max_2Dsize, inv_max_hw, frozen, replica_rows, stream, nullptr, 0.f, flags);
}

// half attribute storage (features, quats, log_scales as IEEE-half mirrors; the gradients stay fp32)
extern "C" int fsgs_gauss_sh_bwd_h16(int N, int degree, const float *means, const float *campos,
                                     const void *features_dc_h, const void *features_rest_h, const void *quats_h,
                                     const void *log_scales_h, const float *scales, const float *opac,
                                     const float *viewmat, const float *K, const float *c2w, int width, int height,
                                     float eps2d, const int32_t *radii, const float *conics, float *v_packed,
                                     float *v_features_dc, float *v_features_rest, float *v_means, float *v_quats,
                                     float *v_log_scales, float *v_opac_logit, float *absgrad, float *xys_grad_norm,
                                     float *vis_counts, float *max_2Dsize, float inv_max_hw, const uint8_t *frozen,
                                     int64_t replica_rows, int flags, fsgs_stream_t stream) {
    return gauss_sh_bwd_impl(N, degree, means, campos, features_dc_h, features_rest_h, quats_h, log_scales_h, 1, scales,
                             opac, viewmat, K, c2w, width, height, eps2d, radii, conics, v_packed, v_features_dc,
                             v_features_rest, nullptr, v_means, v_quats, v_log_scales, v_opac_logit, absgrad,
                             xys_grad_norm, vis_counts, max_2Dsize, inv_max_hw, frozen, replica_rows, stream, nullptr, 0.f, flags);
}

// fsgs_gauss_sh_bwd / _h16 with the Adam step of all six parameter groups applied where the gradients are formed (the
// gradients are never written: 236 B per Gaussian less written here, 236 B less read by an Adam launch that no longer
// exists).  adam: six groups in the order means, log-scales, quats, features_dc, features_rest, opacity logits.
extern "C" int fsgs_gauss_sh_bwd_adam(int N, int degree, const float *means, const float *campos, const float *features_dc,
                                      const float *features_rest, const float *quats, const float *log_scales,
                                      const float *scales, const float *opac, const float *viewmat, const float *K,
                                      const float *c2w, int width, int height, float eps2d, const int32_t *radii,
                                      const float *conics, float *v_packed, float *absgrad, float *xys_grad_norm,
                                      float *vis_counts, float *max_2Dsize, float inv_max_hw, const uint8_t *frozen,
                                      int64_t replica_rows, const fsgs_adam_groups *adam, float min_scale_g,
                                      int flags, fsgs_stream_t stream) {
    if (!adam) return FSGS_EINVAL;
    return gauss_sh_bwd_impl(N, degree, means, campos, features_dc, features_rest, quats, log_scales, 0, scales, opac,
                             viewmat, K, c2w, width, height, eps2d, radii, conics, v_packed, nullptr, nullptr, nullptr,
                             nullptr, nullptr, nullptr, nullptr, absgrad, xys_grad_norm, vis_counts, max_2Dsize,
                             inv_max_hw, frozen, replica_rows, stream, adam, min_scale_g, flags);
}

extern "C" int fsgs_gauss_sh_bwd_adam_h16(int N, int degree, const float *means, const float *campos,
                                          const void *features_dc_h, const void *features_rest_h, const void *quats_h,
                                          const void *log_scales_h, const float *scales, const float *opac,
                                          const float *viewmat, const float *K, const float *c2w, int width, int height,
                                          float eps2d, const int32_t *radii, const float *conics, float *v_packed,
                                          float *absgrad, float *xys_grad_norm, float *vis_counts, float *max_2Dsize,
                                          float inv_max_hw, const uint8_t *frozen, int64_t replica_rows,
                                          const fsgs_adam_groups *adam, float min_scale_g, int flags, fsgs_stream_t stream) {
    if (!adam) return FSGS_EINVAL;
    for (int g = 1; g < 6; ++g)  // (the render path reads the mirrors: every group but the means must have one)
        if (g != 0 && !adam->half_mirror[g]) return FSGS_EINVAL;
    return gauss_sh_bwd_impl(N, degree, means, campos, features_dc_h, features_rest_h, quats_h, log_scales_h, 1, scales,
                             opac, viewmat, K, c2w, width, height, eps2d, radii, conics, v_packed, nullptr, nullptr,
                             nullptr, nullptr, nullptr, nullptr, nullptr, absgrad, xys_grad_norm, vis_counts, max_2Dsize,
                             inv_max_hw, frozen, replica_rows, stream, adam, min_scale_g, flags);
}

static int sh_bwd_impl(int C, int N, int K, int degree, const float *means, const float *campos,
                       const void *coeffs, const void *coeffs_rest, const int32_t *radii, int D,
                       const float *v_colors, float *v_coeffs, float *v_coeffs_rest, float *v_means,
                       float *v_depths, int overwrite_means, fsgs_stream_t stream, float *v_rgb_masked = nullptr,
                       int attr_half = 0, int64_t replica_rows = 0) {
    // D = floats per v_colors row (the colour gradient is its first three): 3, 4 (v_depths = column 3), or a
    // wider stride such as the rasterizer's packed 16-float gradient records
    if (C < 0 || N < 0 || degree < 0 || degree > 4 || K < (degree + 1) * (degree + 1) || K > kMaxK || D < 3)
        return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!means || !campos || !coeffs || !radii || !v_colors || (!v_coeffs && !v_rgb_masked) || !v_means) return FSGS_EINVAL;
    if (v_rgb_masked && C != 1) return FSGS_EINVAL;
    if (replica_rows < 0 || (replica_rows > 0 && (C != 1 || D != 16))) return FSGS_EINVAL;
    const size_t lds_bytes = (size_t)kShBlock * (K * 3 + 1) * sizeof(float);
#define FSGS_SH_BWD(SC, KT, SP)                                                                           \
    hipLaunchKernelGGL((sh_bwd_kernel<SC, KT, SP>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,  \
                       as_stream(stream), C, N, K, degree, means, campos, coeffs, coeffs_rest, radii, D,      \
                       v_colors, v_coeffs, v_coeffs_rest, v_means, v_depths, overwrite_means,                 \
                       reinterpret_cast<float4 *>(v_rgb_masked), replica_rows)
#define FSGS_SH_BWD_K(SC, SP) do { if (K == 16) FSGS_SH_BWD(SC, 16, SP); else FSGS_SH_BWD(SC, 0, SP); } while (0)
    if (attr_half) {  // half coefficient storage: one camera, split features (the training path)
        if (C != 1 || !coeffs_rest) return FSGS_EINVAL;
        if (K == 16)
            hipLaunchKernelGGL((sh_bwd_kernel<true, 16, true, true>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,
                               as_stream(stream), C, N, K, degree, means, campos, coeffs, coeffs_rest, radii, D, v_colors,
                               v_coeffs, v_coeffs_rest, v_means, v_depths, overwrite_means,
                               reinterpret_cast<float4 *>(v_rgb_masked), replica_rows);
        else
            hipLaunchKernelGGL((sh_bwd_kernel<true, 0, true, true>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,
                               as_stream(stream), C, N, K, degree, means, campos, coeffs, coeffs_rest, radii, D, v_colors,
                               v_coeffs, v_coeffs_rest, v_means, v_depths, overwrite_means,
                               reinterpret_cast<float4 *>(v_rgb_masked), replica_rows);
    } else if (C == 1) {
        if (K == 16 && coeffs_rest && (v_rgb_masked || v_coeffs_rest))
            hipLaunchKernelGGL(sh_bwd_hybrid_kernel, dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,
                               as_stream(stream), N, degree, means, campos, reinterpret_cast<const float *>(coeffs),
                               reinterpret_cast<const float *>(coeffs_rest), radii, D, v_colors, v_coeffs,
                               v_coeffs_rest, v_means, v_depths, overwrite_means,
                               reinterpret_cast<float4 *>(v_rgb_masked), replica_rows);
        else if (coeffs_rest) FSGS_SH_BWD_K(true, true); else FSGS_SH_BWD_K(true, false);
    } else {
        if (coeffs_rest) FSGS_SH_BWD_K(false, true); else FSGS_SH_BWD_K(false, false);
    }
#undef FSGS_SH_BWD_K
#undef FSGS_SH_BWD
    return check_launch();
}

extern "C" int fsgs_sh_bwd(int C, int N, int K, int degree, const float *means, const float *campos,
                           const float *coeffs, const int32_t *radii, int D, const float *v_colors,
                           float *v_coeffs, float *v_means, float *v_depths, fsgs_stream_t stream) {
    return sh_bwd_impl(C, N, K, degree, means, campos, coeffs, nullptr, radii, D, v_colors, v_coeffs, nullptr,
                       v_means, v_depths, 0, stream);
}

extern "C" int fsgs_sh_bwd_split(int C, int N, int K, int degree, const float *means, const float *campos,
                                 const float *features_dc, const float *features_rest, const int32_t *radii,
                                 int D, const float *v_colors, float *v_features_dc, float *v_features_rest,
                                 float *v_means, float *v_depths, int overwrite_means,
                                 int64_t replica_rows, fsgs_stream_t stream) {
    if (C >= 0 && N == 0) return FSGS_OK;  // (empty tensors have no address)
    if ((!features_rest || !v_features_rest) && K > 1) return FSGS_EINVAL;
    return sh_bwd_impl(C, N, K, degree, means, campos, features_dc, features_rest ? features_rest : features_dc,
                       radii, D, v_colors, v_features_dc, v_features_rest ? v_features_rest : v_features_dc,
                       v_means, v_depths, overwrite_means, stream, nullptr, 0, replica_rows);
}

// one camera, features read from IEEE-half mirrors (BASELINE config #5); gradients fp32
extern "C" int fsgs_sh_bwd_split_h16(int N, int K, int degree, const float *means, const float *campos,
                                     const void *features_dc_h, const void *features_rest_h, const int32_t *radii,
                                     int D, const float *v_colors, float *v_features_dc, float *v_features_rest,
                                     float *v_means, float *v_depths, int overwrite_means,
                                     int64_t replica_rows, fsgs_stream_t stream) {
    if (N == 0) return FSGS_OK;
    if ((!features_rest_h || !v_features_rest) && K > 1) return FSGS_EINVAL;
    return sh_bwd_impl(1, N, K, degree, means, campos, features_dc_h, features_rest_h ? features_rest_h : features_dc_h,
                       radii, D, v_colors, v_features_dc, v_features_rest ? v_features_rest : v_features_dc, v_means,
                       v_depths, overwrite_means, stream, nullptr, 1, replica_rows);
}

// fsgs_sh_bwd_split for one camera that leaves the coefficient gradients to fsgs_sh_coeff_grad: writes v_means (as
// fsgs_sh_bwd_split) and v_rgb_masked [N + 1,4]: rows 0..N-1 = the colour gradient where the colour clamp is inactive and the Gaussian
// visible, else 0.
extern "C" int fsgs_sh_bwd_colors(int N, int K, int degree, const float *means, const float *campos,
                                  const float *features_dc, const float *features_rest, const int32_t *radii, int D,
                                  const float *v_colors, float *v_rgb_masked, float *v_means, int overwrite_means,
                                  int64_t replica_rows, fsgs_stream_t stream) {
    if (N == 0) return FSGS_OK;
    if ((!features_rest && K > 1) || !v_rgb_masked) return FSGS_EINVAL;
    return sh_bwd_impl(1, N, K, degree, means, campos, features_dc, features_rest ? features_rest : features_dc,
                       radii, D, v_colors, nullptr, nullptr, v_means, nullptr, overwrite_means, stream, v_rgb_masked, 0,
                       replica_rows);
}

extern "C" int fsgs_sh_coeff_grad(int R, int N, int K, int degree, const float *means, const float *gathered,
                                  float scale, float *v_features_dc, float *v_features_rest, fsgs_stream_t stream) {
    if (R < 1 || N < 0 || degree < 0 || degree > 4 || K < (degree + 1) * (degree + 1) || K > kMaxK) return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!means || !gathered || !v_features_dc || (K > 1 && !v_features_rest)) return FSGS_EINVAL;
    const size_t lds_bytes = (size_t)kShBlock * (K * 3 + 1) * sizeof(float);
    const float4 *g4 = reinterpret_cast<const float4 *>(gathered);
    if (K == 16)
        hipLaunchKernelGGL((sh_coeff_grad_kernel<16, false>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,
                           as_stream(stream), R, N, K, degree, means, g4, scale, v_features_dc, v_features_rest,
                           ShAdamArgs{});
    else
        hipLaunchKernelGGL((sh_coeff_grad_kernel<0, false>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,
                           as_stream(stream), R, N, K, degree, means, g4, scale, v_features_dc, v_features_rest,
                           ShAdamArgs{});
    return check_launch();
}

// fsgs_sh_coeff_grad + the Adam step of the two feature groups in one launch: the rebuilt mean gradient is applied
// to features_dc / features_rest (and their exp_avg / exp_avg_sq) as it leaves LDS; it is never written.  Arguments
// of fsgs_adam_step for the two groups; `step` = the Adam step number of this update.
extern "C" int fsgs_sh_coeff_grad_adam(int R, int N, int K, int degree, const float *means, const float *gathered,
                                       float scale, float *features_dc, float *exp_avg_dc, float *exp_avg_sq_dc,
                                       float lr_dc, float *features_rest, float *exp_avg_rest,
                                       float *exp_avg_sq_rest, float lr_rest, int step, double beta1_d, double beta2_d,
                                       float eps, fsgs_stream_t stream) {
    if (R < 1 || N < 0 || degree < 0 || degree > 4 || K < (degree + 1) * (degree + 1) || K > kMaxK || step < 1)
        return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!means || !gathered || !features_dc || !exp_avg_dc || !exp_avg_sq_dc ||
        (K > 1 && (!features_rest || !exp_avg_rest || !exp_avg_sq_rest)))
        return FSGS_EINVAL;
    const double bc1 = 1.0 - pow(beta1_d, (double)step), bc2 = 1.0 - pow(beta2_d, (double)step);
    ShAdamArgs ad = {features_dc, exp_avg_dc, exp_avg_sq_dc, features_rest, exp_avg_rest, exp_avg_sq_rest,
                     (float)((double)lr_dc / bc1), (float)((double)lr_rest / bc1), (float)beta1_d, (float)beta2_d,
                     (float)(1.0 - beta1_d), (float)(1.0 - beta2_d), (float)(1.0 / sqrt(bc2)), eps};
    const size_t lds_bytes = (size_t)kShBlock * (K * 3 + 1) * sizeof(float);
    const float4 *g4 = reinterpret_cast<const float4 *>(gathered);
    if (K == 16)
        hipLaunchKernelGGL((sh_coeff_grad_kernel<16, true>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,
                           as_stream(stream), R, N, K, degree, means, g4, scale, nullptr, nullptr, ad);
    else
        hipLaunchKernelGGL((sh_coeff_grad_kernel<0, true>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,
                           as_stream(stream), R, N, K, degree, means, g4, scale, nullptr, nullptr, ad);
    return check_launch();
}

// fsgs_sh_fwd_split (one camera, depth channel) + fsgs_live_pack_normals in one launch: the colours go straight into
// the packed per-Gaussian records [N,16] (no colour array), normals_world [N,3] and the zeroed cells as there.
// The degree-3-storage packing forward of the training path WITHOUT the LDS staging of sh_fwd_kernel: every lane
// reads its own 192 B of coefficients straight into registers (rows are 180 B apart: the compiler merges the sixteen
// 12-byte pieces into 16-byte loads, the partial lines are shared through L1 / L2), no barrier, no LDS, 5 waves per
// SIMD instead of 3: 24 us against 29-33 us at 300 k Gaussians.  (The same idea for the backward — rows also WRITTEN
// in 12-byte pieces — measured 39 us against 35 us: partial-line stores; it keeps the staged kernel.)
template <bool HALF>
__device__ __forceinline__ void sh_fwd_pack_direct_body(int block, int N, int degree, const float *__restrict__ means,
                                                        const float *__restrict__ campos, const void *__restrict__ dc,
                                                        const void *__restrict__ rest, const int32_t *__restrict__ radii,
                                                        const float *__restrict__ depths, const ShPackArgs &pk) {
    const int n = block * 256 + threadIdx.x;
    if (block == 0)
        for (int k = threadIdx.x; k < pk.n_zero; k += 256) pk.zero_cells[k] = 0.f;
    if (n >= N) return;
    // (dense scenes: most Gaussians are in no tile's list — behind their tiles' occlusion cuts, or reaching no pixel —
    // and the walk gathers records only through the lists: no colours (192 of ~350 B read) and no record for those)
    const bool listed = !pk.kept || pk.kept[n] != 0;
    ShF3 cf[16];
    if (listed) load_sh_row<HALF>(dc, rest, n, N, cf);
    const float mx = means[n * 3 + 0], myy = means[n * 3 + 1], mz = means[n * 3 + 2];
    const int rad0 = listed ? radii[n] : 0;
    const float dep0 = depths[n];
    const float4 pq = ld_attr4(pk.quats, n, HALF ? 1 : 0);
    const float2 pxy = reinterpret_cast<const float2 *>(pk.means2d)[n];
    const float pop = pk.opacities[n];
    float pls[3], pcon[3];
    ld_attr3(pk.log_scales, n, HALF ? 1 : 0, pls);
#pragma unroll
    for (int k = 0; k < 3; ++k) pcon[k] = pk.conics[n * 3 + k];
    const int kk = (degree + 1) * (degree + 1);
    float r = 0.5f, g = 0.5f, bl = 0.5f;
    if (rad0 > 0) {
        float dx = mx - campos[0], dy = myy - campos[1], dz = mz - campos[2];
        const float inorm = sh_inv_norm(dx, dy, dz);
        dx *= inorm; dy *= inorm; dz *= inorm;
        float b[kMaxK];
        sh_basis<false>(degree, dx, dy, dz, b, nullptr, nullptr, nullptr);
        float ar = 0.f, ag = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < kk) { ar = __builtin_fmaf(b[k], cf[k].x, ar); ag = __builtin_fmaf(b[k], cf[k].y, ag); ab = __builtin_fmaf(b[k], cf[k].z, ab); }
        }
        r = fmaxf(ar + 0.5f, 0.f); g = fmaxf(ag + 0.5f, 0.f); bl = fmaxf(ab + 0.5f, 0.f);
    }
    const float mean[3] = {mx, myy, mz};
    NormalCtx o;
    normal_forward(pq, pls, mean, pk.c2w, o);
    float ex[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pk.normals_world[n * 3 + k] = o.n[k];
    if (!listed) return;
    normal_to_camera(o.n, pk.c2w, ex);
    pk.packed[n * 4 + 0] = make_float4(pxy.x, pxy.y, pop, pcon[0]);
    pk.packed[n * 4 + 1] = make_float4(pcon[1], pcon[2], 0.f, 0.f);
    pk.packed[n * 4 + 2] = make_float4(r, g, bl, dep0);
    pk.packed[n * 4 + 3] = make_float4(ex[0], ex[1], ex[2], 0.f);
}

template <bool HALF>
__global__ void __launch_bounds__(256)
sh_fwd_pack_direct_kernel(int N, int degree, const float *__restrict__ means, const float *__restrict__ campos,
                          const void *__restrict__ dc, const void *__restrict__ rest,
                          const int32_t *__restrict__ radii, const float *__restrict__ depths, ShPackArgs pk) {
    sh_fwd_pack_direct_body<HALF>(blockIdx.x, N, degree, means, campos, dc, rest, radii, depths, pk);
}

// The column scan of the direct binning's table (tile_scan.h: a few hundred latency-bound workgroups, 10 us) and
// the SH forward (bandwidth-bound, 22 us) in ONE launch: both only need what the projecting count pass has written,
// neither needs the other; the scan's workgroups take the first block ids, the colours fill the machine meanwhile.
template <bool HALF>
__global__ void __launch_bounds__(256)
scan_rows_sh_pack_kernel(int T, int nb, int32_t *__restrict__ table, int32_t *__restrict__ totals, int n_scan_blocks,
                         int N, int degree, const float *__restrict__ means, const float *__restrict__ campos,
                         const void *__restrict__ dc, const void *__restrict__ rest,
                         const int32_t *__restrict__ radii, const float *__restrict__ depths, ShPackArgs pk,
                         int32_t *__restrict__ ticket, int32_t *__restrict__ isect_offsets,
                         int32_t *__restrict__ total_mapped, BucketBook bk) {
    if ((int)blockIdx.x < n_scan_blocks) {
        if (bk.cursor) {  // the count-free route: no table to scan — one workgroup does the buckets' bookkeeping
            bucket_offsets_body256(T, bk.cursor, bk.base, isect_offsets, bk.next_base, bk.growth, bk.slack, total_mapped,
                                   bk.tile_order);
            return;
        }
        tile_scan_rows_body(T, nb, table, totals, blockIdx.x);
        if (!ticket) return;
        // the scan workgroup that arrives LAST turns the totals into isect_offsets inside this launch, beside the colour
        // workgroups (round 5: tile_offsets_kernel was a one-workgroup launch of its own, 6.7 us of the step; round 2's
        // attempt at this used agent-scope fences and lost).  Totals are stored write-through, drained, then ticketed.
        __shared__ int last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_scan_blocks - 1;
        __syncthreads();
        if (last) tile_offsets_body256(T, totals, isect_offsets, total_mapped, bk.tile_order);
        return;
    }
    sh_fwd_pack_direct_body<HALF>(blockIdx.x - n_scan_blocks, N, degree, means, campos, dc, rest, radii, depths, pk);
}

namespace fsgs {
// (called by the projecting count pass of isect.hip; the arguments are checked there and in sh_pack_rider_ok)
int launch_scan_rows_sh_pack(int T, int nb, int32_t *table, int32_t *totals, const ShPackRider &r, hipStream_t s,
                             int32_t *ticket, int32_t *isect_offsets, int32_t *total_mapped, const BucketBook &bk) {
    ShPackArgs pk = {r.means2d, r.conics, r.opacities, r.quats, r.log_scales, r.c2w, reinterpret_cast<float4 *>(r.packed),
                     r.normals_world, r.zero_cells, r.zero_cells ? r.n_zero : 0, r.kept};
    const int n_scan = bk.cursor ? 1 : tile_scan_rows_blocks(T);
    const dim3 grid(n_scan + ceil_div(r.N, 256));
    if (r.attr_half)
        hipLaunchKernelGGL(scan_rows_sh_pack_kernel<true>, grid, dim3(256), 0, s, T, nb, table, totals, n_scan, r.N,
                           r.degree, r.means, r.campos, r.features_dc, r.features_rest, r.radii, r.depths, pk, ticket,
                           isect_offsets, total_mapped, bk);
    else
        hipLaunchKernelGGL(scan_rows_sh_pack_kernel<false>, grid, dim3(256), 0, s, T, nb, table, totals, n_scan, r.N,
                           r.degree, r.means, r.campos, r.features_dc, r.features_rest, r.radii, r.depths, pk, ticket,
                           isect_offsets, total_mapped, bk);
    return check_launch();
}
}  // namespace fsgs

static int sh_fwd_pack_impl(int N, int K, int degree, const float *means, const float *campos,
                                const void *features_dc, const void *features_rest, const int32_t *radii,
                                const float *depths, const float *means2d, const float *conics,
                                const float *opacities, const void *quats, const void *log_scales,
                                const float *c2w, float *packed, float *normals_world, float *zero_cells,
                                int n_zero, int attr_half, fsgs_stream_t stream) {
    if (N < 0 || degree < 0 || degree > 4 || K < (degree + 1) * (degree + 1) || K > kMaxK) return FSGS_EINVAL;
    if (N == 0) {
        if (zero_cells && n_zero > 0) {
            hipError_t e = hipMemsetAsync(zero_cells, 0, (size_t)n_zero * sizeof(float), as_stream(stream));
            if (e != hipSuccess) { g_last_hip_error = (int)e; return FSGS_ELAUNCH; }
        }
        return FSGS_OK;
    }
    if (!means || !campos || !features_dc || (!features_rest && K > 1) || !radii || !depths || !means2d || !conics ||
        !opacities || !quats || !log_scales || !c2w || !packed || !normals_world)
        return FSGS_EINVAL;
    ShPackArgs pk = {means2d, conics, opacities, quats, log_scales, c2w, reinterpret_cast<float4 *>(packed),
                     normals_world, zero_cells, zero_cells ? n_zero : 0, nullptr};
    const void *rest = features_rest ? features_rest : features_dc;
    const size_t lds_bytes = (size_t)kShBlock * (K * 3 + 1) * sizeof(float);
    if (attr_half) {
        if (K == 16 && features_rest)
            hipLaunchKernelGGL(sh_fwd_pack_direct_kernel<true>, dim3(ceil_div(N, 256)), dim3(256), 0, as_stream(stream), N,
                               degree, means, campos, features_dc, rest, radii, depths, pk);
        else if (K == 16)
            hipLaunchKernelGGL((sh_fwd_kernel<16, true, true, true>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,
                               as_stream(stream), 1, N, K, degree, means, campos, features_dc, rest, radii, depths,
                               nullptr, pk);
        else
            hipLaunchKernelGGL((sh_fwd_kernel<0, true, true, true>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,
                               as_stream(stream), 1, N, K, degree, means, campos, features_dc, rest, radii, depths,
                               nullptr, pk);
        return check_launch();
    }
    if (K == 16 && features_rest)
        hipLaunchKernelGGL(sh_fwd_pack_direct_kernel<false>, dim3(ceil_div(N, 256)), dim3(256), 0, as_stream(stream), N,
                           degree, means, campos, features_dc, rest, radii, depths, pk);
    else if (K == 16)
        hipLaunchKernelGGL((sh_fwd_kernel<16, true, true>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,
                           as_stream(stream), 1, N, K, degree, means, campos, features_dc, rest, radii, depths,
                           nullptr, pk);
    else
        hipLaunchKernelGGL((sh_fwd_kernel<0, true, true>), dim3(ceil_div(N, kShBlock)), dim3(kShBlock), lds_bytes,
                           as_stream(stream), 1, N, K, degree, means, campos, features_dc, rest, radii, depths,
                           nullptr, pk);
    return check_launch();
}

extern "C" int fsgs_sh_fwd_pack(int N, int K, int degree, const float *means, const float *campos,
                                const float *features_dc, const float *features_rest, const int32_t *radii,
                                const float *depths, const float *means2d, const float *conics,
                                const float *opacities, const float *quats, const float *log_scales,
                                const float *c2w, float *packed, float *normals_world, float *zero_cells,
                                int n_zero, fsgs_stream_t stream) {
    return sh_fwd_pack_impl(N, K, degree, means, campos, features_dc, features_rest, radii, depths, means2d, conics,
                            opacities, quats, log_scales, c2w, packed, normals_world, zero_cells, n_zero, 0, stream);
}

// features / quats / log_scales read from IEEE-half mirrors (BASELINE config #5): 124 instead of 236 parameter
// bytes per Gaussian on the render path; all arithmetic fp32
extern "C" int fsgs_sh_fwd_pack_h16(int N, int K, int degree, const float *means, const float *campos,
                                    const void *features_dc_h, const void *features_rest_h, const int32_t *radii,
                                    const float *depths, const float *means2d, const float *conics,
                                    const float *opacities, const void *quats_h, const void *log_scales_h,
                                    const float *c2w, float *packed, float *normals_world, float *zero_cells,
                                    int n_zero, fsgs_stream_t stream) {
    return sh_fwd_pack_impl(N, K, degree, means, campos, features_dc_h, features_rest_h, radii, depths, means2d, conics,
                            opacities, quats_h, log_scales_h, c2w, packed, normals_world, zero_cells, n_zero, 1, stream);
}
