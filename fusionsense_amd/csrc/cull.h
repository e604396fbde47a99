// Exact reach test of a projected Gaussian against the 8x8 pixel quadrants of a 16x16 tile and the 4x4
// blocks of a quadrant (shared by live.hip, the live emission kernels of isect.hip and the compositing
// kernels).
#pragma once
#include "common.h"

namespace fsgs {

constexpr float kCullMargin = 0.02f;  // slack on ln(255*opacity) against fp32 rounding of sigma / exp

// min over the rectangle [x_lo,x_hi] x [y_lo,y_hi] of q(p) = 0.5*(a dx^2 + c dy^2) + b dx dy,
// (dx,dy) = (mx,my) - p, for a positive-definite conic.  Convex => attained at the centre
// (if inside) or on one of the four edges, each a clamped 1-D parabola.
__device__ __forceinline__ float min_sigma_rect(float mx, float my, float a, float b, float c, float inv_a,
                                                float inv_c, float x_lo, float x_hi, float y_lo, float y_hi) {
    const float dx_lo = mx - x_hi, dx_hi = mx - x_lo, dy_lo = my - y_hi, dy_hi = my - y_lo;
    if (dx_lo <= 0.f && dx_hi >= 0.f && dy_lo <= 0.f && dy_hi >= 0.f) return 0.f;
    float best = 3.0e38f;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float dxe = e ? dx_hi : dx_lo;
        const float dy = fminf(fmaxf(-b * dxe * inv_c, dy_lo), dy_hi);
        best = fminf(best, 0.5f * (a * dxe * dxe + c * dy * dy) + b * dxe * dy);
        const float dye = e ? dy_hi : dy_lo;
        const float dx = fminf(fmaxf(-b * dye * inv_a, dx_lo), dx_hi);
        best = fminf(best, 0.5f * (a * dx * dx + c * dye * dye) + b * dx * dye);
    }
    return best;
}

// Per-Gaussian constants of the reach test (computed once, reused for every tile of its rectangle).
struct CullPrep {
    float mx, my, b, tau;          // tau = ln(255 * opacity) + margin: reachable iff min sigma <= tau
    float ha, hc, inv_a, inv_c;    // 0.5 * a, 0.5 * c, 1 / a, 1 / c
};

__device__ __forceinline__ CullPrep cull_prepare(float mx, float my, float opac, float a, float b, float c) {
    CullPrep p;
    p.mx = mx; p.my = my;
    if (!(opac >= kAlphaMin * 0.999f)) {
        // alpha <= opacity < 1/255 everywhere: nothing is reachable (min sigma >= 0 > -inf)
        p.b = 0.f; p.ha = 0.5f; p.hc = 0.5f; p.inv_a = 1.f; p.inv_c = 1.f;
        p.tau = -__builtin_huge_valf();
    } else if (!(a > 0.f && c > 0.f && a * c - b * b > 0.f)) {
        // not positive definite: no bound, every quadrant stays
        p.b = 0.f; p.ha = 0.5f; p.hc = 0.5f; p.inv_a = 1.f; p.inv_c = 1.f;
        p.tau = __builtin_huge_valf();
    } else {
        p.b = b; p.ha = 0.5f * a; p.hc = 0.5f * c;
        p.inv_a = __builtin_amdgcn_rcpf(a); p.inv_c = __builtin_amdgcn_rcpf(c);
        p.tau = __logf(255.f * opac) + kCullMargin;
    }
    return p;
}

// 4-bit mask: bit q = qy*2+qx set if the Gaussian can reach sub-square q of the (2*HALF)x(2*HALF) pixel
// square at (x0, y0).  HALF = 8: the 8x8 quadrants of a 16x16 tile; HALF = 4: the 4x4 blocks of a
// quadrant.  Conservative: a set bit never hides a contribution.
// The four sub-squares' rectangles of pixel centres are bounded by 4 vertical and 4 horizontal lines
// (x0 + {0.5, HALF-0.5, HALF+0.5, 2*HALF-0.5}); on each line the form is a 1-D parabola, minimised over
// each of the two half ranges by clamping its vertex: 16 three-instruction evaluations per square.
template <int HALF>
__device__ __forceinline__ unsigned subsquare_mask(const CullPrep &p, float x0, float y0) {
    constexpr float kL1 = (float)HALF - 0.5f, kL2 = (float)HALF + 0.5f, kL3 = 2.f * (float)HALF - 0.5f;
    float dx[4], dy[4];  // centre minus line coordinate (decreasing in k)
    dx[0] = p.mx - (x0 + 0.5f); dx[1] = p.mx - (x0 + kL1);
    dx[2] = p.mx - (x0 + kL2); dx[3] = p.mx - (x0 + kL3);
    dy[0] = p.my - (y0 + 0.5f); dy[1] = p.my - (y0 + kL1);
    dy[2] = p.my - (y0 + kL2); dy[3] = p.my - (y0 + kL3);
    float qv[4][2], qh[4][2];  // [line][half of the other axis]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float u = p.b * dx[k], t = -u * p.inv_c, A = p.ha * dx[k] * dx[k];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float d = fminf(fmaxf(t, dy[2 * h + 1]), dy[2 * h]);
            qv[k][h] = fmaf(d, fmaf(p.hc, d, u), A);
        }
        const float v = p.b * dy[k], s = -v * p.inv_a, Cc = p.hc * dy[k] * dy[k];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float d = fminf(fmaxf(s, dx[2 * h + 1]), dx[2 * h]);
            qh[k][h] = fmaf(d, fmaf(p.ha, d, v), Cc);
        }
    }
    const bool in_x[2] = {dx[1] <= 0.f && dx[0] >= 0.f, dx[3] <= 0.f && dx[2] >= 0.f};
    const bool in_y[2] = {dy[1] <= 0.f && dy[0] >= 0.f, dy[3] <= 0.f && dy[2] >= 0.f};
    unsigned m = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int qx = q & 1, qy = q >> 1;
        float best = fminf(fminf(qv[2 * qx][qy], qv[2 * qx + 1][qy]), fminf(qh[2 * qy][qx], qh[2 * qy + 1][qx]));
        if (in_x[qx] && in_y[qy]) best = 0.f;
        if (best <= p.tau) m |= 1u << q;
    }
    return m;
}

__device__ __forceinline__ unsigned quadrant_mask(const CullPrep &p, float tile_x0, float tile_y0) {
    return subsquare_mask<8>(p, tile_x0, tile_y0);
}

// the 4x4 pixel blocks of the 8x8 quadrant at (quad_x0, quad_y0): bit = by*2 + bx
__device__ __forceinline__ unsigned block_mask(const CullPrep &p, float quad_x0, float quad_y0) {
    return subsquare_mask<4>(p, quad_x0, quad_y0);
}

__device__ __forceinline__ unsigned quadrant_mask(float mx, float my, float opac, float a, float b, float c,
                                                  float tile_x0, float tile_y0) {
    return quadrant_mask(cull_prepare(mx, my, opac, a, b, c), tile_x0, tile_y0);
}

}  // namespace fsgs
