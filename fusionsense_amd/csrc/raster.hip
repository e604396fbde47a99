// E7 / E8: per-pixel front-to-back alpha compositing, forward and VJP.
// Replaces gsplat 1.0.0 `rasterize_to_pixels_{fwd,bwd}` (reached from
// /root/reference/dn_splatter/dn_model.py:570-591, 4 channels RGB+depth, absgrad) and the legacy
// `rasterize_forward/backward` behind gsplat.rasterize_gaussians (dn_model.py:644-653, 3 channels).
//
// Mapping to CDNA4: one workgroup per 16x16 tile = 256 threads = 4 wave64, wave w covering
// pixel rows 4w..4w+3.  The tile's depth-sorted Gaussian list is walked in batches of 256:
// each thread gathers one Gaussian's (xy, opacity, conic, colour) record from HBM (coalesced
// index read, 12/16-byte gathers) into registers *before* the current batch is composited and
// drops it into the other half of a double-buffered LDS array afterwards, so gather latency
// hides under compute and one barrier per batch is enough.  Inside a batch all 64 lanes read
// the same LDS address (broadcast, conflict-free).  Early termination is decided per wave with
// a 64-bit ballot and folded into the per-batch barrier (no separate __syncthreads_count).
// The forward is atomics-free and deterministic.
#include "common.h"

namespace fsgs {

constexpr int kMaxBlock = 256;

template <int D>
struct Staged {
    int32_t id;
    float x, y, opac;
    float ca, cb, cc;
    float col[D];
};

template <int D>
struct TileLds {
    int32_t id[2][kMaxBlock];
    float4 xyo[2][kMaxBlock];   // x, y, opacity, conic.a
    float2 cbc[2][kMaxBlock];   // conic.b, conic.c
    float col[2][kMaxBlock][D];
    int wave_done[2][4];
};

template <int D>
__device__ __forceinline__ void gather(Staged<D> &s, int64_t isect, bool ok,
                                       const int32_t *__restrict__ flatten_ids,
                                       const float *__restrict__ means2d,
                                       const float *__restrict__ conics,
                                       const float *__restrict__ colors,
                                       const float *__restrict__ opacities) {
    if (ok) {
        const int32_t g = flatten_ids[isect];
        s.id = g;
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
    }
}

template <int D>
__device__ __forceinline__ void put(TileLds<D> &L, int buf, int slot, const Staged<D> &s) {
    L.id[buf][slot] = s.id;
    L.xyo[buf][slot] = make_float4(s.x, s.y, s.opac, s.ca);
    L.cbc[buf][slot] = make_float2(s.cb, s.cc);
#pragma unroll
    for (int k = 0; k < D; ++k) L.col[buf][slot][k] = s.col[k];
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
    __shared__ TileLds<D> L;
    const int cam = blockIdx.z;
    const int tile_lin = (cam * th + blockIdx.y) * tw + blockIdx.x;
    const int n_tiles_total = gridDim.z * th * tw;
    // 1-D workgroup of round_up(tile_size^2, 64) threads: every wave is full, so ballots and
    // DPP reductions always see 64 live lanes; surplus threads only help with staging.
    const int tr = threadIdx.x;
    const int block_size = blockDim.x;
    const int ty = tr / tile_size, tx = tr - ty * tile_size;
    const int i = blockIdx.y * tile_size + ty;
    const int j = blockIdx.x * tile_size + tx;
    const int w = tr >> 6, lane = tr & 63;
    const int n_waves = block_size >> 6;
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
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

    if (tr < 8) (&L.wave_done[0][0])[tr] = 0;  // block_size >= 64
    Staged<D> st;
    gather<D>(st, range_start + tr, range_start + tr < range_end, flatten_ids, means2d, conics, colors,
              opacities);
    if (num_batches > 0) put<D>(L, 0, tr, st);

    for (int b = 0; b < num_batches; ++b) {
        __syncthreads();
        const int buf = b & 1;
        int all = 0;
        for (int k = 0; k < n_waves; ++k) all += L.wave_done[buf][k];
        if (all == n_waves) break;
        const int64_t batch_start = range_start + (int64_t)b * block_size;
        const int batch_size = (int)min((int64_t)block_size, range_end - batch_start);
        const bool more = (b + 1 < num_batches);
        const int64_t nxt = batch_start + block_size + tr;
        if (more) gather<D>(st, nxt, nxt < range_end, flatten_ids, means2d, conics, colors, opacities);

        if (!__all(done)) {
            for (int t = 0; t < batch_size && !done; ++t) {
                const float4 xyo = L.xyo[buf][t];
                const float2 cbc = L.cbc[buf][t];
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
                for (int k = 0; k < D; ++k) pix[k] += L.col[buf][t][k] * vis;
                cur_idx = (int32_t)(batch_start + t);
                T = next_T;
            }
        }
        const int wd = __all(done) ? 1 : 0;
        if (lane == 0) L.wave_done[buf ^ 1][w] = wd;
        if (more) put<D>(L, buf ^ 1, tr, st);
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

// Backward: walk the same list back-to-front starting at last_ids.  For every Gaussian of a
// batch each lane forms its pixel's partial gradients; the 64 partials are summed with DPP
// row operations (no LDS round trips) and lane 63 issues one global fp32 atomic per component.
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
    __shared__ TileLds<D> L;
    const int cam = blockIdx.z;
    const int tile_lin = (cam * th + blockIdx.y) * tw + blockIdx.x;
    const int n_tiles_total = gridDim.z * th * tw;
    const int tr = threadIdx.x;
    const int block_size = blockDim.x;
    const int ty = tr / tile_size, tx = tr - ty * tile_size;
    const int i = blockIdx.y * tile_size + ty;
    const int j = blockIdx.x * tile_size + tx;
    const int lane = tr & 63;
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;
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
    const int bin_final = inside ? last_ids[pix_id] : 0;
    const int wave_bin_final = wave_max_i32(bin_final);

    Staged<D> st;
    {
        const int64_t idx = range_end - 1 - tr;
        gather<D>(st, idx, idx >= range_start, flatten_ids, means2d, conics, colors, opacities);
        put<D>(L, 0, tr, st);
    }
    for (int b = 0; b < num_batches; ++b) {
        __syncthreads();
        const int buf = b & 1;
        const int64_t batch_end = range_end - 1 - (int64_t)block_size * b;
        const int batch_size = (int)min((int64_t)block_size, batch_end + 1 - range_start);
        const bool more = (b + 1 < num_batches);
        const int64_t nxt = batch_end - block_size - tr;
        if (more) gather<D>(st, nxt, nxt >= range_start, flatten_ids, means2d, conics, colors, opacities);

        const int t0 = (int)max((int64_t)0, batch_end - (int64_t)wave_bin_final);
        for (int t = t0; t < batch_size; ++t) {
            bool valid = inside && (batch_end - t <= (int64_t)bin_final);
            const float4 xyo = L.xyo[buf][t];
            const float2 cbc = L.cbc[buf][t];
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
                    const float c = L.col[buf][t][k];
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
                const int64_t g = L.id[buf][t];
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
        if (more) put<D>(L, buf ^ 1, tr, st);
    }
}

}  // namespace fsgs

using namespace fsgs;

template <int D>
static int launch_fwd(int C, int N, const float *means2d, const float *conics, const float *colors,
                      const float *opacities, const float *backgrounds, int width, int height,
                      int tile_size, int tw, int th, const int32_t *offsets, const int32_t *flatten_ids,
                      int64_t n_isects, float *render, float *alphas, int32_t *last_ids, hipStream_t s) {
    hipLaunchKernelGGL((raster_fwd_kernel<D>), dim3(tw, th, C), dim3(((tile_size * tile_size + 63) / 64) * 64), 0, s, N,
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
        hipLaunchKernelGGL((raster_bwd_kernel<D, true>), dim3(tw, th, C), dim3(((tile_size * tile_size + 63) / 64) * 64), 0, s,
                           N, n_isects, means2d, conics, colors, opacities, backgrounds, width, height,
                           tile_size, tw, th, offsets, flatten_ids, alphas, last_ids, v_render, v_alphas,
                           v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities);
    } else {
        hipLaunchKernelGGL((raster_bwd_kernel<D, false>), dim3(tw, th, C), dim3(((tile_size * tile_size + 63) / 64) * 64), 0, s,
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
