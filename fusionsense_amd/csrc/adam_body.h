// Adam for several parameter groups: the kernel body and its argument block, shared by adam_kernel (glue.hip) and
// the binning count pass that carries a deferred Adam launch in its own grid (isect.hip).
#pragma once
#include <math.h>

#include "common.h"

namespace fsgs {
constexpr int kAdamMaxGroups = 8;
struct AdamArgs {
    float *p[kAdamMaxGroups];
    const float *g[kAdamMaxGroups];
    float *m[kAdamMaxGroups];
    float *v[kAdamMaxGroups];
    __half *h[kAdamMaxGroups];      // nullable per group: IEEE-half mirror of the parameter, refreshed with the update
    long long n[kAdamMaxGroups];
    int block_end[kAdamMaxGroups];  // exclusive prefix ends of each group's workgroup range
    float step_size[kAdamMaxGroups];
    int n_groups;
    int n_blocks;                   // block_end of the last group
    float b1, b2, omb1, omb2, inv_sqrt_bc2, eps;  // omb = 1 - beta, rounded from double like torch does
};

// every group owns a whole number of workgroups of `threads` threads; a thread updates U runs of 4 consecutive floats
// (`threads` floats4 apart; all of a thread's loads are issued before its first store)
template <int U = 1>
__device__ __forceinline__ void adam_body(const AdamArgs &a, int block, int threads) {
    int grp = 0;
#pragma unroll
    for (int q = 0; q < kAdamMaxGroups - 1; ++q) grp += (q < a.n_groups - 1 && block >= a.block_end[q]) ? 1 : 0;
    const int blk = block - (grp ? a.block_end[grp - 1] : 0);
    const long long n = a.n[grp];
    const long long j0 = ((long long)blk * U * threads + threadIdx.x) * 4;
    if (j0 >= n) return;
    float *const P0 = a.p[grp], *const M0 = a.m[grp], *const V0 = a.v[grp];
    const float *const G0 = a.g[grp];
    __half *const H0 = a.h[grp];
    const float ss = a.step_size[grp];
    const bool aligned = ((((uintptr_t)P0 | (uintptr_t)M0 | (uintptr_t)V0 | (uintptr_t)G0) & 15) == 0) &&
                         (((uintptr_t)H0 & 7) == 0);
    float4 p[U], m[U], v[U], g[U];
    bool vec[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long j = j0 + (long long)u * threads * 4;
        vec[u] = aligned && (j + 3 < n);
        if (vec[u]) {
            p[u] = *reinterpret_cast<const float4 *>(P0 + j); m[u] = *reinterpret_cast<const float4 *>(M0 + j);
            v[u] = *reinterpret_cast<const float4 *>(V0 + j); g[u] = *reinterpret_cast<const float4 *>(G0 + j);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long j = j0 + (long long)u * threads * 4;
        if (j >= n) break;
        if (vec[u]) {
            adam_one(p[u].x, g[u].x, m[u].x, v[u].x, a.b1, a.b2, a.omb1, a.omb2, ss, a.inv_sqrt_bc2, a.eps);
            adam_one(p[u].y, g[u].y, m[u].y, v[u].y, a.b1, a.b2, a.omb1, a.omb2, ss, a.inv_sqrt_bc2, a.eps);
            adam_one(p[u].z, g[u].z, m[u].z, v[u].z, a.b1, a.b2, a.omb1, a.omb2, ss, a.inv_sqrt_bc2, a.eps);
            adam_one(p[u].w, g[u].w, m[u].w, v[u].w, a.b1, a.b2, a.omb1, a.omb2, ss, a.inv_sqrt_bc2, a.eps);
            *reinterpret_cast<float4 *>(P0 + j) = p[u]; *reinterpret_cast<float4 *>(M0 + j) = m[u];
            *reinterpret_cast<float4 *>(V0 + j) = v[u];
            if (H0) {  // the render path's half mirror of this parameter (BASELINE config #5): 2 more bytes per float
                const __half2 lo = __floats2half2_rn(p[u].x, p[u].y), hi = __floats2half2_rn(p[u].z, p[u].w);
                uint2 o;
                o.x = *reinterpret_cast<const unsigned *>(&lo);
                o.y = *reinterpret_cast<const unsigned *>(&hi);
                *reinterpret_cast<uint2 *>(H0 + j) = o;
            }
        } else {
            for (int k = 0; k < 4 && j + k < n; ++k) {
                float pp = P0[j + k], mm = M0[j + k], vv = V0[j + k];
                adam_one(pp, G0[j + k], mm, vv, a.b1, a.b2, a.omb1, a.omb2, ss, a.inv_sqrt_bc2, a.eps);
                P0[j + k] = pp; M0[j + k] = mm; V0[j + k] = vv;
                if (H0) H0[j + k] = __float2half(pp);
            }
        }
    }
}

// Host: the argument block of one Adam step over n_groups tensors for workgroups that cover threads x U float4s.
// Returns FSGS_OK / FSGS_EINVAL; a.n_blocks = workgroups needed (0: nothing to do).
inline int build_adam_args(AdamArgs &a, int threads_x_unroll, int n_groups, float *const *params, const float *const *grads,
                           float *const *exp_avg, float *const *exp_avg_sq, void *const *half_mirrors,
                           const int64_t *numel, const float *lr, int step, double beta1_d, double beta2_d, float eps) {
    if (n_groups < 1 || n_groups > kAdamMaxGroups || step < 1 || threads_x_unroll < 64) return FSGS_EINVAL;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr) return FSGS_EINVAL;
    const long long chunk = (long long)threads_x_unroll * 4;
    long long blocks = 0;
    const double bc1 = 1.0 - pow(beta1_d, (double)step), bc2 = 1.0 - pow(beta2_d, (double)step);
    for (int k = 0; k < kAdamMaxGroups; ++k) {
        if (k < n_groups) {
            if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || numel[k] < 0) return FSGS_EINVAL;
            a.p[k] = params[k]; a.g[k] = grads[k]; a.m[k] = exp_avg[k]; a.v[k] = exp_avg_sq[k];
            a.h[k] = half_mirrors ? reinterpret_cast<__half *>(half_mirrors[k]) : nullptr;
            a.n[k] = numel[k];
            blocks += (numel[k] + chunk - 1) / chunk;
            a.step_size[k] = (float)((double)lr[k] / bc1);
        } else {
            a.p[k] = nullptr; a.g[k] = nullptr; a.m[k] = nullptr; a.v[k] = nullptr; a.h[k] = nullptr; a.n[k] = 0;
            a.step_size[k] = 0.f;
        }
        if (blocks > 0x7FFFFFFF) return FSGS_EINVAL;
        a.block_end[k] = (int)blocks;
    }
    a.n_groups = n_groups; a.n_blocks = (int)blocks;
    a.b1 = (float)beta1_d; a.b2 = (float)beta2_d; a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2)); a.eps = eps;
    a.omb1 = (float)(1.0 - beta1_d); a.omb2 = (float)(1.0 - beta2_d);
    return FSGS_OK;
}
}  // namespace fsgs
