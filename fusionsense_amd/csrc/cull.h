// Exact reach test of a projected Gaussian against the 8x8 pixel quadrants of a 16x16 tile
// (shared by live.hip and the live emission kernels of isect.hip).
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

// 4-bit mask: bit q = qy*2+qx set if the Gaussian can reach quadrant q of the 16x16 tile at
// (tile_x0, tile_y0).  Conservative: a set bit never hides a contribution.
__device__ __forceinline__ unsigned quadrant_mask(float mx, float my, float opac, float a, float b, float c,
                                                  float tile_x0, float tile_y0) {
    if (!(opac >= kAlphaMin * 0.999f)) return 0u;                   // alpha <= opac < 1/255 everywhere
    if (!(a > 0.f && c > 0.f && a * c - b * b > 0.f)) return 0xFu;  // not positive definite: no bound
    const float tau = __logf(255.f * opac) + kCullMargin;
    const float inv_a = 1.f / a, inv_c = 1.f / c;
    unsigned m = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = tile_x0 + (float)((q & 1) * 8) + 0.5f, y0 = tile_y0 + (float)((q >> 1) * 8) + 0.5f;
        if (min_sigma_rect(mx, my, a, b, c, inv_a, inv_c, x0, x0 + 7.f, y0, y0 + 7.f) <= tau) m |= 1u << q;
    }
    return m;
}

}  // namespace fsgs
