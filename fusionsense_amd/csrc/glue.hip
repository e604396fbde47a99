// Caller-side glue of DNSplatterModel.get_outputs fused into a handful of kernels (SURVEY.md §8f
// row N2 and the "fuse" notes of §8a-11 / §7): the reference performs these steps as ~60 separate
// torch launches per iteration around its two rasterizer calls
// (/root/reference/dn_splatter/dn_model.py:572-575 activations, :602-613 rgb / depth, :655-656 normals).
//   activate_fwd/bwd : exp(scales), sigmoid(opacities) and their VJPs (+ the sum of the two
//                      quaternion gradient streams: projection and normals)
//   epilogue_fwd/bwd : rgb = clamp(render + (1-alpha) * bg, 0, 1); depth = alpha > 0 ? ED : max(ED);
//                      normal = (n / |n| + 1) / 2
// Elementwise, HBM-bound, coalesced; used by fusionsense_amd/fused.py only.
#include <initializer_list>

#include "common.h"
#include "adam_body.h"

namespace fsgs {

__global__ void __launch_bounds__(256)
activate_fwd_kernel(int N, const float *__restrict__ log_scales, const float *__restrict__ opac_logit,
                    float *__restrict__ scales, float *__restrict__ opac) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * 3) scales[i] = expf(log_scales[i]);
    if (i < N) opac[i] = 1.f / (1.f + expf(-opac_logit[i]));
}

__global__ void __launch_bounds__(256)
activate_bwd_kernel(int N, const float *__restrict__ scales, const float *__restrict__ opac,
                    const float *__restrict__ v_scales, const float *__restrict__ v_opac,
                    const float *__restrict__ v_quats_a, const float *__restrict__ v_quats_b,
                    float *__restrict__ v_log_scales, float *__restrict__ v_opac_logit,
                    float *__restrict__ v_quats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * 3) v_log_scales[i] = v_scales[i] * scales[i];
    if (i < N) {
        const float s = opac[i];
        v_opac_logit[i] = v_opac[i] * s * (1.f - s);
    }
    if (i < N * 4) v_quats[i] = v_quats_a[i] + (v_quats_b ? v_quats_b[i] : 0.f);
}

// one workgroup: max over the per-quadrant partial maxima
__global__ void __launch_bounds__(1024)
max_reduce_kernel(int n, const float *__restrict__ partial, float *__restrict__ out) {
    __shared__ float red[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, partial[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float r = red[0];
        for (int k = 1; k < 16; ++k) r = fmaxf(r, red[k]);
        out[0] = r;
    }
}

// LOSS: the depth / normal L1 terms of the training loss (fsgs_aux_l1_fwd_bwd of loss.hip) are evaluated on the
// pixel this thread has just produced: partial sums per workgroup and the two gradient images, no second pass over
// the depth and normal images and one launch less (the trainer's tape-free step, which knows targets and loss seed).
struct EpilogueLoss {
    const float *depth_gt, *normal_gt, *v_loss;
    float *partial, *v_depth, *v_normal;
    float g_depth, g_normal;
};

// Rider of the epilogue launch (round 5): the dispatch order of the compositing BACKWARD.  A backward workgroup's time is its
// quadrant's own share of records, and the launch is ~6 rounds of resident workgroups: started longest first, its last
// round is the short ones (config #3's trained scene: 0.191 -> 0.174 ms; config #2: 0.1675 -> 0.165).  The streams' lengths
// are known once the forward has finished — i.e. here: extra blocks of this launch (one wave per 256 quadrants, dispatched first) file every
// quadrant under one of kOrderClasses classes of record counts (16 records wide, the last one open-ended) — a counting
// sort: per-block class counts in LDS, ONE returning atomic per (block, class) on the class's counter reserves the block's
// run in the class's list, order[class][slot] = the quadrant's position in the backward's grid.  (Filing from the
// forward's finishing waves — one returning atomic per quadrant on a handful of hot counters — took the forward from 78 to
// 130 us.)  One camera.
constexpr int kOrderClasses = 32;
constexpr int kOrderPerLane = 4;  // a rider wave files 256 quadrants (16 per lane in ~12 blocks left a tail behind the bulk)
struct BwdOrderRider {
    const int32_t *n_rec, *seg_split;  // [4, T]
    int32_t *counters;                 // [kOrderClasses], zeroed by the caller (behind the backward's queue)
    int32_t *order;                    // [kOrderClasses, 4 T]
    int tw, n_tiles, n_rider;          // rider blocks = blockIdx.x < n_rider: dispatched FIRST, beside the launch's bulk
    int shift;                         // class = min(own records >> shift, kOrderClasses - 1)
};

__device__ __forceinline__ void bwd_order_rider_block(const BwdOrderRider &r, int block) {
    __shared__ int cnt[kOrderClasses], base[kOrderClasses];
    const int lane = threadIdx.x;  // 64 threads
    const int nq = 4 * r.n_tiles;
    if (lane < kOrderClasses) cnt[lane] = 0;
    __syncthreads();
    int cls[kOrderPerLane], rank[kOrderPerLane];
#pragma unroll
    for (int k = 0; k < kOrderPerLane; ++k) {
        const int i = (block * kOrderPerLane + k) * 64 + lane;  // index into [4, T]: quadrant q = i / T of tile i % T
        cls[k] = -1;
        if (i < nq) {
            const int own = r.n_rec[i] - 64 * (r.seg_split ? r.seg_split[i] : 0);
            cls[k] = min(max(own, 0) >> r.shift, kOrderClasses - 1);
            rank[k] = atomicAdd(&cnt[cls[k]], 1);
        }
    }
    __syncthreads();
    if (lane < kOrderClasses) base[lane] = cnt[lane] > 0 ? atomicAdd(r.counters + lane, cnt[lane]) : 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kOrderPerLane; ++k) {
        if (cls[k] < 0) continue;
        const int i = (block * kOrderPerLane + k) * 64 + lane;
        const int q = i / r.n_tiles, t = i - q * r.n_tiles;
        const int ty = t / r.tw, tx = t - ty * r.tw;
        r.order[(int64_t)cls[k] * nq + base[cls[k]] + rank[k]] = (2 * ty + (q >> 1)) * (2 * r.tw) + 2 * tx + (q & 1);
    }
}

// The image epilogues read and write four pixels per lane with 16-byte accesses: every image pointer must be 16-byte
// aligned (fsgs.h says so; a view that starts in the middle of an allocation is refused instead of relying on the GPU's
// unaligned-access mode).
static inline bool aligned16(std::initializer_list<const void *> ps) {
    for (const void *q : ps)
        if (q && (reinterpret_cast<uintptr_t>(q) & 15u)) return false;
    return true;
}

// One WAVE per 256 pixels, four consecutive pixels per lane (round 5): every array is read and written with 16-byte
// accesses — the 3-channel images as three float4 per lane (12 floats = 4 pixels) instead of twelve 4-byte accesses at a
// stride of 12 bytes, which cost three partial-line transactions each (17.4 -> see DESIGN.md 5.4).  The per-256-pixel
// partial sums keep their layout (one row per workgroup, as fsgs_aux_l1_fwd leaves them).
template <bool LOSS>
__global__ void __launch_bounds__(64)
epilogue_fwd_kernel(int64_t P, const float4 *__restrict__ render, const float *__restrict__ alphas,
                    const float *__restrict__ render_extra, const float *__restrict__ bg,
                    const float *__restrict__ max_last, int n_cells, float *__restrict__ rgb,
                    float *__restrict__ depth, float *__restrict__ normal, EpilogueLoss L, BwdOrderRider rider) {
    if ((int)blockIdx.x < rider.n_rider) {
        bwd_order_rider_block(rider, (int)blockIdx.x);
        return;
    }
    const int lane = threadIdx.x;
    const int blk = (int)blockIdx.x - rider.n_rider;
    const int64_t p0 = ((int64_t)blk * 64 + lane) * 4;  // this lane's first pixel
    const bool full = p0 + 3 < P;                              // all four pixels exist (else: per-pixel accesses)
    float a[4];
    float4 r[4];
    if (full) {
        const float4 a4 = *reinterpret_cast<const float4 *>(alphas + p0);
        a[0] = a4.x; a[1] = a4.y; a[2] = a4.z; a[3] = a4.w;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = render[p0 + k];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = (p0 + k < P) ? alphas[p0 + k] : 1.f;
            r[k] = (p0 + k < P) ? render[p0 + k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // the image maximum arrives as n_cells partial maxima; only waves with an empty pixel need it
    float fill = 0.f;
    const bool empty = !(a[0] > 0.f) || !(a[1] > 0.f) || !(a[2] > 0.f) || !(a[3] > 0.f);
    if (__any(empty)) {
        for (int k = lane; k < n_cells; k += 64) fill = fmaxf(fill, max_last[k]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) fill = fmaxf(fill, __shfl_xor(fill, d, 64));
    }
    const float up = LOSS ? L.v_loss[0] : 0.f;
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    float out_rgb[12], out_d[4], out_vd[4], out_n[12], out_vn[12];
    float sd = 0.f, sn = 0.f;
    float dgt[4] = {0.f, 0.f, 0.f, 0.f}, ex[12], ngt[12];
    const bool have_ngt = LOSS && L.normal_gt != nullptr;
    if (full) {
        if (LOSS) {
            const float4 g4 = *reinterpret_cast<const float4 *>(L.depth_gt + p0);
            dgt[0] = g4.x; dgt[1] = g4.y; dgt[2] = g4.z; dgt[3] = g4.w;
        }
        if (normal) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 e4 = *reinterpret_cast<const float4 *>(render_extra + p0 * 3 + 4 * k);
                ex[4 * k + 0] = e4.x; ex[4 * k + 1] = e4.y; ex[4 * k + 2] = e4.z; ex[4 * k + 3] = e4.w;
                if (have_ngt) {
                    const float4 n4 = *reinterpret_cast<const float4 *>(L.normal_gt + p0 * 3 + 4 * k);
                    ngt[4 * k + 0] = n4.x; ngt[4 * k + 1] = n4.y; ngt[4 * k + 2] = n4.z; ngt[4 * k + 3] = n4.w;
                }
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in = p0 + k < P;
            if (LOSS) dgt[k] = in ? L.depth_gt[p0 + k] : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                ex[3 * k + c] = (in && normal) ? render_extra[(p0 + k) * 3 + c] : 1.f;
                ngt[3 * k + c] = (in && have_ngt) ? L.normal_gt[(p0 + k) * 3 + c] : 0.f;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool in = p0 + k < P;
        const float t = 1.f - a[k];
        out_rgb[3 * k + 0] = fminf(fmaxf(r[k].x + t * b0, 0.f), 1.f);
        out_rgb[3 * k + 1] = fminf(fmaxf(r[k].y + t * b1, 0.f), 1.f);
        out_rgb[3 * k + 2] = fminf(fmaxf(r[k].z + t * b2, 0.f), 1.f);
        const float dep = (a[k] > 0.f) ? r[k].w : fill;
        out_d[k] = dep;
        if (LOSS) {
            const float d = dep - dgt[k];
            sd += in ? fabsf(d) : 0.f;
            out_vd[k] = up * L.g_depth * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
        }
        if (normal) {
            const float nx = ex[3 * k + 0], ny = ex[3 * k + 1], nz = ex[3 * k + 2];
            const float inv = 1.f / sqrtf(nx * nx + ny * ny + nz * nz);
            const float nn[3] = {(nx * inv + 1.f) * 0.5f, (ny * inv + 1.f) * 0.5f, (nz * inv + 1.f) * 0.5f};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                out_n[3 * k + c] = nn[c];
                if (have_ngt) {
                    const float e = nn[c] - ngt[3 * k + c];
                    sn += in ? fabsf(e) : 0.f;
                    out_vn[3 * k + c] = up * L.g_normal * ((e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f));
                }
            }
        }
    }
    auto st4 = [](float *dst, const float *v) { *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]); };
    if (full) {
#pragma unroll
        for (int k = 0; k < 3; ++k) st4(rgb + p0 * 3 + 4 * k, out_rgb + 4 * k);
        st4(depth + p0, out_d);
        if (LOSS) st4(L.v_depth + p0, out_vd);
        if (normal) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                st4(normal + p0 * 3 + 4 * k, out_n + 4 * k);
                if (have_ngt) st4(L.v_normal + p0 * 3 + 4 * k, out_vn + 4 * k);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (p0 + k < P) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    rgb[(p0 + k) * 3 + c] = out_rgb[3 * k + c];
                    if (normal) normal[(p0 + k) * 3 + c] = out_n[3 * k + c];
                    if (normal && have_ngt) L.v_normal[(p0 + k) * 3 + c] = out_vn[3 * k + c];
                }
                depth[p0 + k] = out_d[k];
                if (LOSS) L.v_depth[p0 + k] = out_vd[k];
            }
        }
    }
    if (LOSS) {  // per-workgroup partial sums, same layout as fsgs_aux_l1_fwd (256 pixels per row)
        sd = wave_sum_to_last_row(sd);
        sn = wave_sum_to_last_row(sn);
        if (lane == 63) {
            L.partial[2 * blk + 0] = sd;
            L.partial[2 * blk + 1] = sn;
        }
    }
}

__global__ void __launch_bounds__(256)
epilogue_bwd_kernel(int64_t P, const float4 *__restrict__ render, const float *__restrict__ alphas,
                    const float *__restrict__ render_extra, const float *__restrict__ bg,
                    const float *__restrict__ v_rgb, const float *__restrict__ v_depth,
                    const float *__restrict__ v_normal, const float *__restrict__ v_alpha_in,
                    float4 *__restrict__ v_render, float *__restrict__ v_alphas,
                    float *__restrict__ v_render_extra) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float4 r = render[p];
    const float a = alphas[p];
    const float t = 1.f - a;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float va = v_alpha_in ? v_alpha_in[p] : 0.f;
    if (v_rgb) {
        const float pre[3] = {r.x + t * bg[0], r.y + t * bg[1], r.z + t * bg[2]};
        float gg[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gg[k] = (pre[k] >= 0.f && pre[k] <= 1.f) ? v_rgb[p * 3 + k] : 0.f;
            va -= bg[k] * gg[k];
        }
        g.x = gg[0]; g.y = gg[1]; g.z = gg[2];
    }
    if (v_depth) g.w = (a > 0.f) ? v_depth[p] : 0.f;
    v_render[p] = g;
    v_alphas[p] = va;
    if (v_render_extra) {
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (v_normal) {
            const float nx = render_extra[p * 3 + 0], ny = render_extra[p * 3 + 1], nz = render_extra[p * 3 + 2];
            const float inv = 1.f / sqrtf(nx * nx + ny * ny + nz * nz);
            const float ux = nx * inv, uy = ny * inv, uz = nz * inv;
            const float hx = 0.5f * v_normal[p * 3 + 0], hy = 0.5f * v_normal[p * 3 + 1], hz = 0.5f * v_normal[p * 3 + 2];
            const float d = hx * ux + hy * uy + hz * uz;
            gx = (hx - d * ux) * inv; gy = (hy - d * uy) * inv; gz = (hz - d * uz) * inv;
        }
        v_render_extra[p * 3 + 0] = gx; v_render_extra[p * 3 + 1] = gy; v_render_extra[p * 3 + 2] = gz;
    }
}

}  // namespace fsgs

using namespace fsgs;

extern "C" int fsgs_activate_fwd(int N, const float *log_scales, const float *opac_logit, float *scales,
                                 float *opac, fsgs_stream_t stream) {
    if (N < 0) return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!log_scales || !opac_logit || !scales || !opac) return FSGS_EINVAL;
    hipLaunchKernelGGL(activate_fwd_kernel, dim3(ceil_div((int64_t)N * 3, 256)), dim3(256), 0, as_stream(stream),
                       N, log_scales, opac_logit, scales, opac);
    return check_launch();
}

extern "C" int fsgs_activate_bwd(int N, const float *scales, const float *opac, const float *v_scales,
                                 const float *v_opac, const float *v_quats_a, const float *v_quats_b,
                                 float *v_log_scales, float *v_opac_logit, float *v_quats,
                                 fsgs_stream_t stream) {
    if (N < 0) return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!scales || !opac || !v_scales || !v_opac || !v_quats_a || !v_log_scales || !v_opac_logit || !v_quats)
        return FSGS_EINVAL;
    hipLaunchKernelGGL(activate_bwd_kernel, dim3(ceil_div((int64_t)N * 4, 256)), dim3(256), 0, as_stream(stream),
                       N, scales, opac, v_scales, v_opac, v_quats_a, v_quats_b, v_log_scales, v_opac_logit,
                       v_quats);
    return check_launch();
}

extern "C" int fsgs_epilogue_fwd(int64_t n_pixels, const float *render, const float *alphas,
                                 const float *render_extra, const float *bg, const float *max_last_partial,
                                 int n_partial, float *max_last, float *rgb, float *depth, float *normal,
                                 fsgs_stream_t stream) {
    if (n_pixels < 0) return FSGS_EINVAL;
    if (n_pixels == 0) return FSGS_OK;
    if (!render || !alphas || !bg || (n_partial > 0 && !max_last_partial) || !max_last || !rgb || !depth)
        return FSGS_EINVAL;
    if (normal && !render_extra) return FSGS_EINVAL;
    if (!aligned16({render, alphas, render_extra, rgb, depth, normal})) return FSGS_EINVAL;
    hipStream_t s = as_stream(stream);
    if (n_partial > 0)  // n_partial <= 0: max_last already holds max(1, -n_partial) partial maxima (fsgs_raster_fwd_quad)
        hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(1024), 0, s, n_partial, max_last_partial, max_last);
    const int n_cells = n_partial < 0 ? -n_partial : 1;
    hipLaunchKernelGGL(epilogue_fwd_kernel<false>, dim3(ceil_div(n_pixels, 256)), dim3(64), 0, s, n_pixels,
                       reinterpret_cast<const float4 *>(render), alphas, render_extra, bg, max_last, n_cells, rgb, depth,
                       normal, EpilogueLoss{}, BwdOrderRider{});
    return check_launch();
}

// fsgs_epilogue_fwd (max_last = n_cells partial maxima as left by fsgs_raster_fwd_quad) + fsgs_aux_l1_fwd_bwd in one
// pass: the depth / normal L1 partial sums [ceil(n_pixels / 256), 2] and gradient images come out with the images.
extern "C" int fsgs_epilogue_loss_fwd(int64_t n_pixels, const float *render, const float *alphas,
                                      const float *render_extra, const float *bg, const float *max_last, int n_cells,
                                      float *rgb, float *depth, float *normal, const float *depth_gt,
                                      const float *normal_gt, const float *v_loss, float g_depth, float g_normal,
                                      float *partial, float *v_depth, float *v_normal, fsgs_stream_t stream) {
    if (!depth_gt) return FSGS_EINVAL;
    return fsgs_epilogue_fwd_order(n_pixels, render, alphas, render_extra, bg, max_last, n_cells, rgb, depth, normal,
                                   depth_gt, normal_gt, v_loss, g_depth, g_normal, partial, v_depth, v_normal, nullptr,
                                   nullptr, nullptr, nullptr, 0, 0, 0, stream);
}

// fsgs_epilogue_loss_fwd (depth_gt != NULL) or fsgs_epilogue_fwd over the forward's partial maxima (depth_gt == NULL: no
// loss terms) — and, riding in the launch, the dispatch order of the compositing backward (bwd_order != NULL: fsgs.h)
extern "C" int fsgs_epilogue_fwd_order(int64_t n_pixels, const float *render, const float *alphas,
                                       const float *render_extra, const float *bg, const float *max_last, int n_cells,
                                       float *rgb, float *depth, float *normal, const float *depth_gt,
                                       const float *normal_gt, const float *v_loss, float g_depth, float g_normal,
                                       float *partial, float *v_depth, float *v_normal, const int32_t *n_rec,
                                       const int32_t *seg_split, int32_t *order_counters, int32_t *bwd_order,
                                       int order_shift, int tile_width, int tile_height, fsgs_stream_t stream) {
    if (n_pixels < 0 || n_cells < 1) return FSGS_EINVAL;
    if (n_pixels == 0) return FSGS_OK;
    if (!render || !alphas || !bg || !max_last || !rgb || !depth) return FSGS_EINVAL;
    if (depth_gt && (!v_loss || !partial || !v_depth)) return FSGS_EINVAL;
    if ((normal && !render_extra) || (normal_gt && (!depth_gt || !normal || !v_normal))) return FSGS_EINVAL;
    if (bwd_order && (!n_rec || !order_counters || tile_width < 1 || tile_height < 1 || order_shift < 0 || order_shift > 16))
        return FSGS_EINVAL;
    if (!aligned16({render, alphas, render_extra, rgb, depth, normal, depth_gt, normal_gt, v_depth, v_normal})) return FSGS_EINVAL;
    const int n_blocks = (int)ceil_div(n_pixels, 256);
    BwdOrderRider rd{};
    int n_rider = 0;
    if (bwd_order) {
        n_rider = (int)ceil_div((int64_t)4 * tile_width * tile_height, 64 * kOrderPerLane);
        rd = BwdOrderRider{n_rec, seg_split, order_counters, bwd_order, tile_width, tile_width * tile_height, n_rider,
                           order_shift};
    }
    const float4 *r4 = reinterpret_cast<const float4 *>(render);
    if (depth_gt) {
        const EpilogueLoss L = {depth_gt, normal_gt, v_loss, partial, v_depth, v_normal, g_depth, g_normal};
        hipLaunchKernelGGL(epilogue_fwd_kernel<true>, dim3(n_blocks + n_rider), dim3(64), 0, as_stream(stream), n_pixels, r4,
                           alphas, render_extra, bg, max_last, n_cells, rgb, depth, normal, L, rd);
    } else {
        hipLaunchKernelGGL(epilogue_fwd_kernel<false>, dim3(n_blocks + n_rider), dim3(64), 0, as_stream(stream), n_pixels,
                           r4, alphas, render_extra, bg, max_last, n_cells, rgb, depth, normal, EpilogueLoss{}, rd);
    }
    return check_launch();
}

extern "C" int fsgs_epilogue_bwd(int64_t n_pixels, const float *render, const float *alphas,
                                 const float *render_extra, const float *bg, const float *v_rgb,
                                 const float *v_depth, const float *v_normal, const float *v_alpha_in,
                                 float *v_render, float *v_alphas, float *v_render_extra,
                                 fsgs_stream_t stream) {
    if (n_pixels < 0) return FSGS_EINVAL;
    if (n_pixels == 0) return FSGS_OK;
    if (!render || !alphas || !bg || !v_render || !v_alphas) return FSGS_EINVAL;
    if (v_render_extra && !render_extra) return FSGS_EINVAL;
    hipLaunchKernelGGL(epilogue_bwd_kernel, dim3(ceil_div(n_pixels, 256)), dim3(256), 0, as_stream(stream),
                       n_pixels, reinterpret_cast<const float4 *>(render), alphas, render_extra, bg, v_rgb, v_depth,
                       v_normal, v_alpha_in, reinterpret_cast<float4 *>(v_render), v_alphas, v_render_extra);
    return check_launch();
}

// ---- N1: Adam for all Gaussian parameter groups in ONE launch -----------------------------------
// The reference steps eight torch Adam optimizers per iteration (dn_splatter/dn_config.py:36-75;
// six hold live parameters).  Same update rule as torch.optim.Adam (amsgrad=False, weight_decay=0):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// HBM-bound by construction: 28 B per parameter float (1.65 KB per Gaussian per step).
namespace fsgs {
__global__ void __launch_bounds__(256) adam_kernel(AdamArgs a) { adam_body<1>(a, blockIdx.x, 256); }
}  // namespace fsgs

extern "C" int fsgs_adam_step(int n_groups, float *const *params, const float *const *grads, float *const *exp_avg,
                              float *const *exp_avg_sq, const int64_t *numel, const float *lr, int step, double beta1_d,
                              double beta2_d, float eps, fsgs_stream_t stream) {
    return fsgs_adam_step_h16(n_groups, params, grads, exp_avg, exp_avg_sq, nullptr, numel, lr, step, beta1_d, beta2_d,
                              eps, stream);
}

// The same update; half_mirrors (nullable array, nullable entries): an IEEE-half copy of the group's parameters that
// is rewritten together with the fp32 master (round to nearest even) — what the render kernels read in config #5.
extern "C" int fsgs_adam_step_h16(int n_groups, float *const *params, const float *const *grads, float *const *exp_avg,
                                  float *const *exp_avg_sq, void *const *half_mirrors, const int64_t *numel,
                                  const float *lr, int step, double beta1_d, double beta2_d, float eps,
                                  fsgs_stream_t stream) {
    fsgs::AdamArgs a;
    const int rc = fsgs::build_adam_args(a, 256, n_groups, params, grads, exp_avg, exp_avg_sq, half_mirrors, numel, lr,
                                         step, beta1_d, beta2_d, eps);
    if (rc != FSGS_OK) return rc;
    if (a.n_blocks == 0) return FSGS_OK;
    hipLaunchKernelGGL(fsgs::adam_kernel, dim3((unsigned)a.n_blocks), dim3(256), 0, fsgs::as_stream(stream), a);
    return fsgs::check_launch();
}
