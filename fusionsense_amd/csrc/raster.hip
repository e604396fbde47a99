// E7 / E8: per-pixel front-to-back alpha compositing, forward and VJP.
// Replaces gsplat 1.0.0 `rasterize_to_pixels_{fwd,bwd}` (reached from
// /root/reference/dn_splatter/dn_model.py:570-591, 4 channels RGB+depth, absgrad) and the legacy
// `rasterize_forward/backward` behind gsplat.rasterize_gaussians (dn_model.py:644-653, 3 channels).
//
// Mapping to CDNA4: one workgroup per 16x16 tile = 256 threads = 4 wave64, wave w covering the
// 8x8 pixel quadrant w.  The tile's depth-sorted list is walked in batches of 256:
//   * STAGE  each thread gathers one list entry's (xy, opacity, conic, colour) record from HBM
//            into registers *before* the current batch is composited (latency hides under
//            compute) and decides, exactly, which of the four wave quadrants the Gaussian can reach:
//            it minimises the conic's quadratic form over each strip's rectangle of pixel centres
//            in closed form and compares with ln(255*opacity) (the alpha >= 1/255 level set) —
//            the square 3-sigma bbox the lists were built from is far looser than that ellipse.
//            Entries that reach no strip are dropped; the rest are compacted per staging wave
//            with a 64-bit ballot into that wave's LDS segment (order preserved), together with
//            the 4-bit strip mask and the entry's list index.
//   * WALK   every wave runs over the 4 segments; all 64 lanes read the same LDS address
//            (broadcast).  A wave skips entries whose mask excludes its strip before touching
//            the exponential.  One barrier per batch (double-buffered LDS); early termination is
//            a per-wave ballot folded into that barrier.
// Dropping an entry is only done when NO pixel of the strip could pass `alpha >= 1/255` (with a
// safety margin on the bound), so images, last_ids and gradients are those of the plain walk.
// The forward is atomics-free and deterministic.
#include "common.h"

namespace fsgs {

constexpr int kMaxBlock = 256;
constexpr int kMaxWaves = kMaxBlock / 64;
constexpr float kCullMargin = 0.02f;  // slack on ln(255*opacity) against fp32 rounding of sigma / exp

template <int D>
struct Staged {
    int32_t id;       // Gaussian (flatten) id
    int32_t isect;    // index into the sorted list
    float x, y, opac;
    float ca, cb, cc;
    float col[D];
    unsigned mask;    // bit w set: wave strip w can be reached
};

template <int D, bool WITH_ID>
struct TileLds {
    float4 xyo[2][kMaxWaves][64];   // x, y, opacity, conic.a
    float2 cbc[2][kMaxWaves][64];   // conic.b, conic.c
    float col[2][kMaxWaves][64][D];
    int2 im[2][kMaxWaves][64];      // list index, strip mask
    int32_t id[WITH_ID ? 2 : 1][WITH_ID ? kMaxWaves : 1][WITH_ID ? 64 : 1];
    int cnt[2][kMaxWaves];
    int wave_done[2][kMaxWaves];
};

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

// Thread -> pixel mapping inside a tile.  For the 16x16 tile each wave owns an 8x8 quadrant
// (compact footprints touch fewer waves than with 16x4 strips); other tile sizes use row strips.
struct PixRect { int x0, x1, y0, y1; };  // inclusive pixel offsets inside the tile

__device__ __forceinline__ void thread_pixel(int tr, int tile_size, int &tx, int &ty) {
    if (tile_size == 16) {
        const int w = tr >> 6, l = tr & 63;
        tx = ((w & 1) << 3) + (l & 7);
        ty = ((w >> 1) << 3) + (l >> 3);
    } else {
        ty = tr / tile_size;
        tx = tr - ty * tile_size;
    }
}

__device__ __forceinline__ bool wave_rect(int w, int tile_size, PixRect &r) {
    if (tile_size == 16) {
        r.x0 = (w & 1) << 3; r.x1 = r.x0 + 7;
        r.y0 = (w >> 1) << 3; r.y1 = r.y0 + 7;
        return true;
    }
    r.x0 = 0; r.x1 = tile_size - 1;
    r.y0 = (64 * w) / tile_size;
    r.y1 = min((64 * w + 63) / tile_size, tile_size - 1);
    return r.y0 < tile_size;
}

// Which wave regions can this Gaussian reach?  Conservative: a set bit never hides a contribution.
__device__ __forceinline__ unsigned strip_mask(float mx, float my, float opac, float a, float b, float c,
                                               float tile_x0, float tile_y0, int tile_size, int n_waves) {
    const unsigned all = (1u << n_waves) - 1u;
    if (!(opac >= kAlphaMin * 0.999f)) return 0u;      // alpha <= opac < 1/255 everywhere (NaN -> 0 too)
    if (!(a > 0.f && c > 0.f && a * c - b * b > 0.f)) return all;  // not positive definite: no bound
    const float tau = __logf(255.f * opac) + kCullMargin;
    const float inv_a = 1.f / a, inv_c = 1.f / c;
    unsigned m = 0u;
    for (int w = 0; w < n_waves; ++w) {
        PixRect r;
        if (!wave_rect(w, tile_size, r)) break;
        const float smin = min_sigma_rect(mx, my, a, b, c, inv_a, inv_c, tile_x0 + (float)r.x0 + 0.5f,
                                          tile_x0 + (float)r.x1 + 0.5f, tile_y0 + (float)r.y0 + 0.5f,
                                          tile_y0 + (float)r.y1 + 0.5f);
        if (smin <= tau) m |= (1u << w);
    }
    return m;
}

template <int D>
__device__ __forceinline__ void gather(Staged<D> &s, int64_t isect, bool ok,
                                       const int32_t *__restrict__ flatten_ids,
                                       const float *__restrict__ means2d, const float *__restrict__ conics,
                                       const float *__restrict__ colors, const float *__restrict__ opacities,
                                       float tile_x0, float tile_y0, int tile_size, int n_waves) {
    s.mask = 0u;
    if (ok) {
        const int32_t g = flatten_ids[isect];
        s.id = g;
        s.isect = (int32_t)isect;
        const float2 m = reinterpret_cast<const float2 *>(means2d)[g];
        s.x = m.x; s.y = m.y;
        s.opac = opacities[g];
        s.ca = conics[(int64_t)g * 3 + 0];
        s.cb = conics[(int64_t)g * 3 + 1];
        s.cc = conics[(int64_t)g * 3 + 2];
        if (D == 4) {
            const float4 c4 = reinterpret_cast<const float4 *>(colors)[g];
            s.col[0] = c4.x; s.col[1] = c4.y; s.col[2] = c4.z; s.col[D - 1] = c4.w;
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k) s.col[k] = colors[(int64_t)g * D + k];
        }
        s.mask = strip_mask(s.x, s.y, s.opac, s.ca, s.cb, s.cc, tile_x0, tile_y0, tile_size, n_waves);
    }
}

__device__ __forceinline__ unsigned long long lanemask_lt64(int lane) {
    return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}

// Compact this wave's live entries into its LDS segment; every lane of the wave must call.
template <int D, bool WITH_ID>
__device__ __forceinline__ void put(TileLds<D, WITH_ID> &L, int buf, int w, int lane, const Staged<D> &s) {
    const bool live = s.mask != 0u;
    const unsigned long long b = __ballot(live);
    if (live) {
        const int slot = __popcll(b & lanemask_lt64(lane));
        L.xyo[buf][w][slot] = make_float4(s.x, s.y, s.opac, s.ca);
        L.cbc[buf][w][slot] = make_float2(s.cb, s.cc);
#pragma unroll
        for (int k = 0; k < D; ++k) L.col[buf][w][slot][k] = s.col[k];
        L.im[buf][w][slot] = make_int2(s.isect, (int)s.mask);
        if (WITH_ID) L.id[buf][w][slot] = s.id;
    }
    if (lane == 0) L.cnt[buf][w] = __popcll(b);
}

template <int D>
__global__ void __launch_bounds__(kMaxBlock)
raster_fwd_kernel(int N, int64_t n_isects, const float *__restrict__ means2d,
                  const float *__restrict__ conics, const float *__restrict__ colors,
                  const float *__restrict__ opacities, const float *__restrict__ backgrounds,
                  int W, int H, int tile_size, int tw, int th,
                  const int32_t *__restrict__ tile_offsets, const int32_t *__restrict__ flatten_ids,
                  float *__restrict__ render, float *__restrict__ alphas,
                  int32_t *__restrict__ last_ids) {
    __shared__ TileLds<D, false> L;
    const int cam = blockIdx.z;
    const int tile_lin = (cam * th + blockIdx.y) * tw + blockIdx.x;
    const int n_tiles_total = gridDim.z * th * tw;
    // 1-D workgroup of round_up(tile_size^2, 64) threads: every wave is full, so ballots and
    // DPP reductions always see 64 live lanes; surplus threads only help with staging.
    const int tr = threadIdx.x;
    const int block_size = blockDim.x;
    int tx, ty;
    thread_pixel(tr, tile_size, tx, ty);
    const int i = blockIdx.y * tile_size + ty;
    const int j = blockIdx.x * tile_size + tx;
    const int w = tr >> 6, lane = tr & 63;
    const int n_waves = block_size >> 6;
    const unsigned my_bit = 1u << w;
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const float tile_x0 = (float)(blockIdx.x * tile_size), tile_y0 = (float)(blockIdx.y * tile_size);
    const bool inside = (ty < tile_size) && (i < H) && (j < W);
    bool done = !inside;

    const int64_t range_start = tile_offsets[tile_lin];
    const int64_t range_end =
        (tile_lin == n_tiles_total - 1) ? n_isects : (int64_t)tile_offsets[tile_lin + 1];
    const int num_batches = (int)((range_end - range_start + block_size - 1) / block_size);

    float T = 1.f;
    int32_t cur_idx = 0;
    float pix[D];
#pragma unroll
    for (int k = 0; k < D; ++k) pix[k] = 0.f;

    if (tr < 2 * kMaxWaves) (&L.wave_done[0][0])[tr] = 0;
    Staged<D> st;
    if (num_batches > 0) {
        gather<D>(st, range_start + tr, range_start + tr < range_end, flatten_ids, means2d, conics, colors,
                  opacities, tile_x0, tile_y0, tile_size, n_waves);
        put<D, false>(L, 0, w, lane, st);
    }

    for (int b = 0; b < num_batches; ++b) {
        __syncthreads();
        const int buf = b & 1;
        int all = 0;
        for (int k = 0; k < n_waves; ++k) all += L.wave_done[buf][k];
        if (all == n_waves) break;
        const bool more = (b + 1 < num_batches);
        const int64_t nxt = range_start + (int64_t)(b + 1) * block_size + tr;
        if (more)
            gather<D>(st, nxt, nxt < range_end, flatten_ids, means2d, conics, colors, opacities, tile_x0,
                      tile_y0, tile_size, n_waves);

        if (!__all(done)) {
            for (int seg = 0; seg < n_waves && !done; ++seg) {
                const int n = L.cnt[buf][seg];
                for (int t = 0; t < n && !done; ++t) {
                    const int2 im = L.im[buf][seg][t];
                    if (!((unsigned)im.y & my_bit)) continue;  // wave-uniform
                    const float4 xyo = L.xyo[buf][seg][t];
                    const float2 cbc = L.cbc[buf][seg][t];
                    const float dx = xyo.x - px, dy = xyo.y - py;
                    const float sigma = 0.5f * (xyo.w * dx * dx + cbc.y * dy * dy) + cbc.x * dx * dy;
                    const float alpha = fminf(kAlphaMax, xyo.z * __expf(-sigma));
                    if (sigma < 0.f || alpha < kAlphaMin) continue;
                    const float next_T = T * (1.f - alpha);
                    if (next_T <= kTMin) {
                        done = true;
                        break;
                    }
                    const float vis = alpha * T;
#pragma unroll
                    for (int k = 0; k < D; ++k) pix[k] += L.col[buf][seg][t][k] * vis;
                    cur_idx = im.x;
                    T = next_T;
                }
            }
        }
        const int wd = __all(done) ? 1 : 0;
        if (lane == 0) L.wave_done[buf ^ 1][w] = wd;
        if (more) put<D, false>(L, buf ^ 1, w, lane, st);
    }

    if (inside) {
        const int64_t pix_id = ((int64_t)cam * H + i) * W + j;
        if (D == 4) {
            float4 o = make_float4(pix[0], pix[1], pix[2], pix[D - 1]);
            if (backgrounds) {
                const float *bg = backgrounds + cam * D;
                o.x += T * bg[0]; o.y += T * bg[1]; o.z += T * bg[2]; o.w += T * bg[D - 1];
            }
            reinterpret_cast<float4 *>(render)[pix_id] = o;
        } else {
#pragma unroll
            for (int k = 0; k < D; ++k)
                render[pix_id * D + k] = pix[k] + (backgrounds ? T * backgrounds[cam * D + k] : 0.f);
        }
        alphas[pix_id] = 1.f - T;
        last_ids[pix_id] = cur_idx;
    }
}

__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}

// Backward: walk the same list back-to-front starting at last_ids.  For every live entry each
// lane forms its pixel's partial gradients; the 64 partials are summed with DPP row operations
// (no LDS round trips) and lane 63 issues one global fp32 atomic per component.
template <int D, bool ABS>
__global__ void __launch_bounds__(kMaxBlock)
raster_bwd_kernel(int N, int64_t n_isects, const float *__restrict__ means2d,
                  const float *__restrict__ conics, const float *__restrict__ colors,
                  const float *__restrict__ opacities, const float *__restrict__ backgrounds,
                  int W, int H, int tile_size, int tw, int th,
                  const int32_t *__restrict__ tile_offsets, const int32_t *__restrict__ flatten_ids,
                  const float *__restrict__ alphas, const int32_t *__restrict__ last_ids,
                  const float *__restrict__ v_render, const float *__restrict__ v_alphas,
                  float *__restrict__ v_means2d, float *__restrict__ v_means2d_abs,
                  float *__restrict__ v_conics, float *__restrict__ v_colors,
                  float *__restrict__ v_opacities) {
    __shared__ TileLds<D, true> L;
    const int cam = blockIdx.z;
    const int tile_lin = (cam * th + blockIdx.y) * tw + blockIdx.x;
    const int n_tiles_total = gridDim.z * th * tw;
    const int tr = threadIdx.x;
    const int block_size = blockDim.x;
    int tx, ty;
    thread_pixel(tr, tile_size, tx, ty);
    const int i = blockIdx.y * tile_size + ty;
    const int j = blockIdx.x * tile_size + tx;
    const int w = tr >> 6, lane = tr & 63;
    const int n_waves = block_size >> 6;
    const unsigned my_bit = 1u << w;
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
    const float tile_x0 = (float)(blockIdx.x * tile_size), tile_y0 = (float)(blockIdx.y * tile_size);
    const bool inside = (ty < tile_size) && (i < H) && (j < W);
    const int64_t pix_id = ((int64_t)cam * H + min(i, H - 1)) * W + min(j, W - 1);

    const int64_t range_start = tile_offsets[tile_lin];
    const int64_t range_end =
        (tile_lin == n_tiles_total - 1) ? n_isects : (int64_t)tile_offsets[tile_lin + 1];
    const int num_batches = (int)((range_end - range_start + block_size - 1) / block_size);
    if (num_batches == 0) return;

    const float T_final = 1.f - alphas[pix_id];
    float T = T_final;
    float buffer[D], v_out[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        buffer[k] = 0.f;
        v_out[k] = inside ? v_render[pix_id * D + k] : 0.f;
    }
    const float v_out_a = inside ? v_alphas[pix_id] : 0.f;
    float bg_dot = 0.f;
    if (backgrounds) {
#pragma unroll
        for (int k = 0; k < D; ++k) bg_dot += backgrounds[cam * D + k] * v_out[k];
    }
    const int bin_final = inside ? last_ids[pix_id] : -1;
    const int wave_bin_final = wave_max_i32(bin_final);

    Staged<D> st;
    {
        const int64_t idx = range_end - 1 - tr;
        gather<D>(st, idx, idx >= range_start, flatten_ids, means2d, conics, colors, opacities, tile_x0,
                  tile_y0, tile_size, n_waves);
        put<D, true>(L, 0, w, lane, st);
    }
    for (int b = 0; b < num_batches; ++b) {
        __syncthreads();
        const int buf = b & 1;
        const int64_t batch_end = range_end - 1 - (int64_t)block_size * b;
        const bool more = (b + 1 < num_batches);
        const int64_t nxt = batch_end - block_size - tr;
        if (more)
            gather<D>(st, nxt, nxt >= range_start, flatten_ids, means2d, conics, colors, opacities, tile_x0,
                      tile_y0, tile_size, n_waves);

        // entries are in descending list order; nothing above wave_bin_final was composited
        if (batch_end - (int64_t)(block_size - 1) <= (int64_t)wave_bin_final) {
            for (int seg = 0; seg < n_waves; ++seg) {
                const int n = L.cnt[buf][seg];
                for (int t = 0; t < n; ++t) {
                    const int2 im = L.im[buf][seg][t];
                    if (!((unsigned)im.y & my_bit) || im.x > wave_bin_final) continue;  // wave-uniform
                    bool valid = inside && (im.x <= bin_final);
                    const float4 xyo = L.xyo[buf][seg][t];
                    const float2 cbc = L.cbc[buf][seg][t];
                    const float dx = xyo.x - px, dy = xyo.y - py;
                    const float sigma = 0.5f * (xyo.w * dx * dx + cbc.y * dy * dy) + cbc.x * dx * dy;
                    const float vis = __expf(-sigma);
                    const float alpha = fminf(kAlphaMax, xyo.z * vis);
                    if (sigma < 0.f || alpha < kAlphaMin) valid = false;
                    if (!__any(valid)) continue;

                    float g_col[D];
                    float g_ca = 0.f, g_cb = 0.f, g_cc = 0.f, g_x = 0.f, g_y = 0.f, g_op = 0.f;
#pragma unroll
                    for (int k = 0; k < D; ++k) g_col[k] = 0.f;
                    if (valid) {
                        const float ra = 1.f / (1.f - alpha);
                        T *= ra;
                        const float fac = alpha * T;
                        float v_alpha = 0.f;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            const float c = L.col[buf][seg][t][k];
                            g_col[k] = fac * v_out[k];
                            v_alpha += (c * T - buffer[k] * ra) * v_out[k];
                            buffer[k] += c * fac;
                        }
                        v_alpha += T_final * ra * v_out_a;
                        if (backgrounds) v_alpha -= T_final * ra * bg_dot;
                        if (xyo.z * vis <= kAlphaMax) {
                            const float v_sigma = -xyo.z * vis * v_alpha;
                            g_ca = 0.5f * v_sigma * dx * dx;
                            g_cb = v_sigma * dx * dy;
                            g_cc = 0.5f * v_sigma * dy * dy;
                            g_x = v_sigma * (xyo.w * dx + cbc.x * dy);
                            g_y = v_sigma * (cbc.x * dx + cbc.y * dy);
                            g_op = vis * v_alpha;
                        }
                    }
                    float g_ax = 0.f, g_ay = 0.f;
                    if (ABS) {
                        g_ax = wave_sum_to_last_row(fabsf(g_x));
                        g_ay = wave_sum_to_last_row(fabsf(g_y));
                    }
#pragma unroll
                    for (int k = 0; k < D; ++k) g_col[k] = wave_sum_to_last_row(g_col[k]);
                    g_ca = wave_sum_to_last_row(g_ca);
                    g_cb = wave_sum_to_last_row(g_cb);
                    g_cc = wave_sum_to_last_row(g_cc);
                    g_x = wave_sum_to_last_row(g_x);
                    g_y = wave_sum_to_last_row(g_y);
                    g_op = wave_sum_to_last_row(g_op);
                    if (lane == 63) {
                        const int64_t g = L.id[buf][seg][t];
#pragma unroll
                        for (int k = 0; k < D; ++k) unsafeAtomicAdd(&v_colors[g * D + k], g_col[k]);
                        unsafeAtomicAdd(&v_conics[g * 3 + 0], g_ca);
                        unsafeAtomicAdd(&v_conics[g * 3 + 1], g_cb);
                        unsafeAtomicAdd(&v_conics[g * 3 + 2], g_cc);
                        unsafeAtomicAdd(&v_means2d[g * 2 + 0], g_x);
                        unsafeAtomicAdd(&v_means2d[g * 2 + 1], g_y);
                        if (ABS) {
                            unsafeAtomicAdd(&v_means2d_abs[g * 2 + 0], g_ax);
                            unsafeAtomicAdd(&v_means2d_abs[g * 2 + 1], g_ay);
                        }
                        unsafeAtomicAdd(&v_opacities[g], g_op);
                    }
                }
            }
        }
        if (more) put<D, true>(L, buf ^ 1, w, lane, st);
    }
}

}  // namespace fsgs

using namespace fsgs;

static inline unsigned raster_block(int tile_size) { return ((tile_size * tile_size + 63) / 64) * 64; }

template <int D>
static int launch_fwd(int C, int N, const float *means2d, const float *conics, const float *colors,
                      const float *opacities, const float *backgrounds, int width, int height,
                      int tile_size, int tw, int th, const int32_t *offsets, const int32_t *flatten_ids,
                      int64_t n_isects, float *render, float *alphas, int32_t *last_ids, hipStream_t s) {
    hipLaunchKernelGGL((raster_fwd_kernel<D>), dim3(tw, th, C), dim3(raster_block(tile_size)), 0, s, N,
                       n_isects, means2d, conics, colors, opacities, backgrounds, width, height,
                       tile_size, tw, th, offsets, flatten_ids, render, alphas, last_ids);
    return check_launch();
}

extern "C" int fsgs_raster_fwd(int C, int N, int D, const float *means2d, const float *conics,
                               const float *colors, const float *opacities, const float *backgrounds,
                               int width, int height, int tile_size, int tile_width, int tile_height,
                               const int32_t *isect_offsets, const int32_t *flatten_ids,
                               int64_t n_isects, float *render, float *alphas, int32_t *last_ids,
                               fsgs_stream_t stream) {
    if (C < 0 || N < 0 || width < 0 || height < 0 || tile_size < 2 || tile_size > 16 || n_isects < 0)
        return FSGS_EINVAL;
    if (C == 0 || width == 0 || height == 0) return FSGS_OK;
    if (!isect_offsets || !render || !alphas || !last_ids) return FSGS_EINVAL;
    if (n_isects > 0 && (!means2d || !conics || !colors || !opacities || !flatten_ids)) return FSGS_EINVAL;
    hipStream_t s = as_stream(stream);
    switch (D) {
        case 1: return launch_fwd<1>(C, N, means2d, conics, colors, opacities, backgrounds, width, height, tile_size, tile_width, tile_height, isect_offsets, flatten_ids, n_isects, render, alphas, last_ids, s);
        case 3: return launch_fwd<3>(C, N, means2d, conics, colors, opacities, backgrounds, width, height, tile_size, tile_width, tile_height, isect_offsets, flatten_ids, n_isects, render, alphas, last_ids, s);
        case 4: return launch_fwd<4>(C, N, means2d, conics, colors, opacities, backgrounds, width, height, tile_size, tile_width, tile_height, isect_offsets, flatten_ids, n_isects, render, alphas, last_ids, s);
        default: return FSGS_EINVAL;
    }
}

template <int D>
static int launch_bwd(int C, int N, const float *means2d, const float *conics, const float *colors,
                      const float *opacities, const float *backgrounds, int width, int height,
                      int tile_size, int tw, int th, const int32_t *offsets, const int32_t *flatten_ids,
                      int64_t n_isects, const float *alphas, const int32_t *last_ids,
                      const float *v_render, const float *v_alphas, float *v_means2d,
                      float *v_means2d_abs, float *v_conics, float *v_colors, float *v_opacities,
                      hipStream_t s) {
    if (v_means2d_abs) {
        hipLaunchKernelGGL((raster_bwd_kernel<D, true>), dim3(tw, th, C), dim3(raster_block(tile_size)), 0, s,
                           N, n_isects, means2d, conics, colors, opacities, backgrounds, width, height,
                           tile_size, tw, th, offsets, flatten_ids, alphas, last_ids, v_render, v_alphas,
                           v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities);
    } else {
        hipLaunchKernelGGL((raster_bwd_kernel<D, false>), dim3(tw, th, C), dim3(raster_block(tile_size)), 0, s,
                           N, n_isects, means2d, conics, colors, opacities, backgrounds, width, height,
                           tile_size, tw, th, offsets, flatten_ids, alphas, last_ids, v_render, v_alphas,
                           v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities);
    }
    return check_launch();
}

extern "C" int fsgs_raster_bwd(int C, int N, int D, const float *means2d, const float *conics,
                               const float *colors, const float *opacities, const float *backgrounds,
                               int width, int height, int tile_size, int tile_width, int tile_height,
                               const int32_t *isect_offsets, const int32_t *flatten_ids,
                               int64_t n_isects, const float *alphas, const int32_t *last_ids,
                               const float *v_render, const float *v_alphas, float *v_means2d,
                               float *v_means2d_abs, float *v_conics, float *v_colors,
                               float *v_opacities, fsgs_stream_t stream) {
    if (C < 0 || N < 0 || width < 0 || height < 0 || tile_size < 2 || tile_size > 16 || n_isects < 0)
        return FSGS_EINVAL;
    if (C == 0 || width == 0 || height == 0 || n_isects == 0) return FSGS_OK;
    if (!means2d || !conics || !colors || !opacities || !isect_offsets || !flatten_ids || !alphas ||
        !last_ids || !v_render || !v_alphas || !v_means2d || !v_conics || !v_colors || !v_opacities)
        return FSGS_EINVAL;
    hipStream_t s = as_stream(stream);
#define FSGS_BWD_ARGS C, N, means2d, conics, colors, opacities, backgrounds, width, height, tile_size, \
    tile_width, tile_height, isect_offsets, flatten_ids, n_isects, alphas, last_ids, v_render, v_alphas, \
    v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities, s
    switch (D) {
        case 1: return launch_bwd<1>(FSGS_BWD_ARGS);
        case 3: return launch_bwd<3>(FSGS_BWD_ARGS);
        case 4: return launch_bwd<4>(FSGS_BWD_ARGS);
        default: return FSGS_EINVAL;
    }
#undef FSGS_BWD_ARGS
}
