// a-15: the terms DNSplatterModel.get_loss_dict adds to splatfacto's photometric loss
// (/root/reference/dn_splatter/dn_model.py:673-925), as one stencil pass over the rendered depth / normal images
// plus two tiny per-Gaussian kernels.  The reference evaluates them as ~60 elementwise torch launches with
// boolean-mask gathers; here a thread owns a pixel, reads its right / left / down / up neighbours through the
// cache, and writes the two gradient images in the same pass when the caller knows the loss seed.
//   EdgeAwareLogL1 (dn_splatter/losses.py:177-214), TVLoss (:269-285), mask handling (dn_model.py:702-714),
//   valid mask (:721), clamp of the edge image (:692), normal L1 + TV (:809-815), min-scale (:817-819),
//   touch-normal MSE (:893-902).
// HBM-bound: ~70 B/pixel algorithmic (depth 4, normal 12, image 12, sensor 4, normal_gt 12, mask 4; writes 16 + partials).
#include "common.h"

namespace fsgs {

constexpr int kFA = 8;  // columns of the partial-sum table

__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

struct FusionAuxW { float w[8]; };

// flags of fsgs_fusion_aux_loss (fsgs.h): the switches of get_loss_dict that are off in FusionSense's configuration
constexpr int kFaGtUnmasked = 1;   // normal_gt is used as it is (pseudo normals from depth, dn_model.py:774-795), not * mask
constexpr int kFaEdgeTv = 2;       // depth smoothness = EdgeAwareTV (losses.py:241-266) instead of TVLoss
constexpr int kFaTypeShift = 2;    // bits 2-3: depth-loss type 0 EdgeAwareLogL1, 1 L1, 2 LogL1, 3 MSE (dn_model.py:725-736)

// Riders of the fusion_aux launch (round 5): the min-scale term's and the touch-normal term's partial sums are two
// launches of a few workgroups each (5 us apiece with their boundaries) that depend on nothing the step has not long
// finished — their blocks take extra grid ROWS of this launch instead (blockIdx.y >= gy).
struct FusionAuxRiders {
    int gy;                       // first rider row (= the image's block rows)
    int N, nm;                    // min-scale: Gaussians, blocks
    const float *log_scales;
    float *ms_partial;
    int n_touch, nt;              // touch normals: anchors, blocks
    const int64_t *touch_idx;
    const float *normals_world, *touch_normals;
    float *t_partial;
};

__device__ __forceinline__ void min_scale_partial_block(int block, int N, const float *__restrict__ log_scales,
                                                        float *__restrict__ partial, float *red) {
    const int n = block * 256 + threadIdx.x;
    float v = 0.f;
    if (n < N) {
        const float s0 = log_scales[n * 3 + 0], s1 = log_scales[n * 3 + 1], s2 = log_scales[n * 3 + 2];
        v = expf(fminf(s0, fminf(s1, s2)));  // min exp = exp min (monotone)
    }
    v = wave_sum_to_last_row(v);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * block + 0] = red[0] + red[1] + red[2] + red[3];
        partial[2 * block + 1] = 0.f;
    }
}

__device__ __forceinline__ void touch_sqerr_partial_block(int block, int n, const int64_t *__restrict__ idx,
                                                          const float *__restrict__ normals_world,
                                                          const float *__restrict__ touch_normals,
                                                          float *__restrict__ partial, float *red) {
    const int i = block * 256 + threadIdx.x;
    float v = 0.f;
    if (i < n) {
        const int64_t r = idx[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float e = normals_world[r * 3 + k] - touch_normals[(int64_t)i * 3 + k];
            v += e * e;
        }
    }
    v = wave_sum_to_last_row(v);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * block + 0] = red[0] + red[1] + red[2] + red[3];
        partial[2 * block + 1] = 0.f;
    }
}

// 256 threads = 32 x 8 pixels
template <bool GRADS>
__global__ void __launch_bounds__(256)
fusion_aux_kernel(int H, int W, const float *__restrict__ depth, const float *__restrict__ normal,
                  const float *__restrict__ image, const float *__restrict__ sensor,
                  const float *__restrict__ normal_gt, const float *__restrict__ mask, float tol, FusionAuxW ww,
                  const float *__restrict__ v_loss, float *__restrict__ partial, float *__restrict__ v_depth,
                  float *__restrict__ v_normal, int flags, FusionAuxRiders rd) {
    __shared__ float red[4][kFA];
    if ((int)blockIdx.y >= rd.gy) {  // a rider block (uniform per workgroup)
        const int r = ((int)blockIdx.y - rd.gy) * (int)gridDim.x + (int)blockIdx.x;
        if (r < rd.nm) min_scale_partial_block(r, rd.N, rd.log_scales, rd.ms_partial, &red[0][0]);
        else if (r - rd.nm < rd.nt)
            touch_sqerr_partial_block(r - rd.nm, rd.n_touch, rd.touch_idx, rd.normals_world, rd.touch_normals, rd.t_partial,
                                      &red[0][0]);
        return;
    }
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    const bool in = x < W && y < H;
    float acc[kFA];
#pragma unroll
    for (int k = 0; k < kFA; ++k) acc[k] = 0.f;
    if (in) {
        const int64_t p = (int64_t)y * W + x;
        const bool hr = x + 1 < W, hl = x > 0, hd = y + 1 < H, hu = y > 0;
        const int64_t pr = hr ? p + 1 : p, pl = hl ? p - 1 : p, pd = hd ? p + W : p, pu = hu ? p - W : p;
        const float m = mask ? mask[p] : 1.f;
        const float mr = mask ? mask[pr] : 1.f, ml = mask ? mask[pl] : 1.f, md = mask ? mask[pd] : 1.f,
                    mu = mask ? mask[pu] : 1.f;
        const float up = GRADS ? v_loss[0] : 0.f;
        // ---- depth: EdgeAwareLogL1 against the sensor depth + TV ----
        const float d = depth[p] * m;
        const float dr = depth[pr] * mr, dl = depth[pl] * ml, dd = depth[pd] * md, du = depth[pu] * mu;
        const float g = sensor[p] * m;
        const bool valid = g > tol;
        const int dtype = (flags >> kFaTypeShift) & 3;
        const bool edge_tv = (flags & kFaEdgeTv) != 0;
        float gd = 0.f;
        // edge weights exp(-mean_c |dI|) of the clamped ground-truth image towards the right / lower / left / upper pixel
        auto lam_to = [&](int64_t q) {
            float s3 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                s3 += fabsf(fmaxf(image[p * 3 + k], 10.f / 255.f) - fmaxf(image[q * 3 + k], 10.f / 255.f));
            return expf(-s3 * (1.f / 3.f));
        };
        if (valid && dtype == 0) {
            const float e = d - g;
            const float ll = logf(1.f + fabsf(e));
            float wsum = 0.f;
            if (hr) {
                const float lam = lam_to(pr);
                acc[0] = lam * ll;
                wsum += lam * ww.w[0];
            }
            if (hd) {
                const float lam = lam_to(pd);
                acc[1] = lam * ll;
                wsum += lam * ww.w[1];
            }
            gd = wsum * sgnf(e) / (1.f + fabsf(e));
        } else if (valid) {
            // L1 / LogL1 / MSE over the valid pixels (losses.py:147-174, torch.nn.MSELoss): column 0, weight lambda / #valid
            const float e = d - g, ae = fabsf(e);
            acc[0] = dtype == 1 ? ae : (dtype == 2 ? logf(1.f + ae) : e * e);
            gd = ww.w[0] * (dtype == 1 ? sgnf(e) : (dtype == 2 ? sgnf(e) / (1.f + ae) : 2.f * e));
        }
        if (!edge_tv) {
            if (hr) acc[2] = fabsf(d - dr);
            if (hd) acc[3] = fabsf(d - dd);
            if (GRADS)
                gd += ww.w[2] * ((hr ? sgnf(d - dr) : 0.f) - (hl ? sgnf(dl - d) : 0.f)) +
                      ww.w[3] * ((hd ? sgnf(d - dd) : 0.f) - (hu ? sgnf(du - d) : 0.f));
        } else {
            // EdgeAwareTV: every difference weighted by the image edge between the same two pixels
            const float lr_ = hr ? lam_to(pr) : 0.f, ld_ = hd ? lam_to(pd) : 0.f;
            if (hr) acc[2] = lr_ * fabsf(d - dr);
            if (hd) acc[3] = ld_ * fabsf(d - dd);
            if (GRADS) {
                const float ll_ = hl ? lam_to(pl) : 0.f, lu_ = hu ? lam_to(pu) : 0.f;
                gd += ww.w[2] * (lr_ * (hr ? sgnf(d - dr) : 0.f) - ll_ * (hl ? sgnf(dl - d) : 0.f)) +
                      ww.w[3] * (ld_ * (hd ? sgnf(d - dd) : 0.f) - lu_ * (hu ? sgnf(du - d) : 0.f));
            }
        }
        if (GRADS) v_depth[p] = up * m * gd;
        // ---- normals: L1 against the monocular normals + TV ----
        if (normal) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float n = normal[p * 3 + k] * m;
                const float nr = normal[pr * 3 + k] * mr, nl = normal[pl * 3 + k] * ml, nd = normal[pd * 3 + k] * md,
                            nu = normal[pu * 3 + k] * mu;
                float gn = 0.f;
                if (normal_gt) {
                    const float e = n - normal_gt[p * 3 + k] * ((flags & kFaGtUnmasked) ? 1.f : m);
                    acc[4] += fabsf(e);
                    gn = ww.w[4] * sgnf(e);
                }
                if (hr) acc[5] += fabsf(n - nr);
                if (hd) acc[6] += fabsf(n - nd);
                if (GRADS) {
                    gn += ww.w[5] * ((hr ? sgnf(n - nr) : 0.f) - (hl ? sgnf(nl - n) : 0.f)) +
                          ww.w[6] * ((hd ? sgnf(n - nd) : 0.f) - (hu ? sgnf(nu - n) : 0.f));
                    v_normal[p * 3 + k] = up * m * gn;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kFA - 1; ++k) {
        const float s = wave_sum_to_last_row(acc[k]);
        if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < kFA) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        const int k = threadIdx.x;
        partial[(int64_t)blk * kFA + k] = (k < kFA - 1) ? red[0][k] + red[1][k] + red[2][k] + red[3][k] : 0.f;
    }
}

// Pseudo normals of the (masked) rendered depth, as get_loss_dict's normal_supervision == "depth" forms its target
// (dn_model.py:774-795 over utils/normal_utils.py:8-46 and utils/camera_utils.py:92-144 with c2w = identity): every pixel
// centre (x + 0.5, y + 0.5) is back-projected with its z-depth d * mask; an interior pixel's normal is
// normalize(cross(right - left, top - bottom)) (F.normalize: / max(|v|, 1e-12)), the one-pixel border is zero; then the
// y and z components are negated and the result is mapped to [0, 1]: out = (1 + n * (1, -1, -1)) / 2.
__global__ void __launch_bounds__(256)
normals_from_depth_kernel(int H, int W, const float *__restrict__ depth, const float *__restrict__ mask, float fx,
                          float fy, float cx, float cy, float *__restrict__ out) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const int64_t p = (int64_t)y * W + x;
    float n0 = 0.f, n1 = 0.f, n2 = 0.f;
    if (x > 0 && y > 0 && x + 1 < W && y + 1 < H) {
        auto point = [&](int xx, int yy, float (&q)[3]) {
            const int64_t i = (int64_t)yy * W + xx;
            const float d = depth[i] * (mask ? mask[i] : 1.f);
            q[0] = ((float)xx + 0.5f - cx) * d / fx;
            q[1] = ((float)yy + 0.5f - cy) * d / fy;
            q[2] = d;
        };
        float r[3], l[3], t[3], b[3];
        point(x + 1, y, r); point(x - 1, y, l); point(x, y - 1, t); point(x, y + 1, b);
        const float a0 = r[0] - l[0], a1 = r[1] - l[1], a2 = r[2] - l[2];  // left -> right
        const float b0 = t[0] - b[0], b1 = t[1] - b[1], b2 = t[2] - b[2];  // bottom -> top
        const float c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
        const float inv = 1.f / fmaxf(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), 1e-12f);
        n0 = c0 * inv; n1 = c1 * inv; n2 = c2 * inv;
    }
    out[p * 3 + 0] = (1.f + n0) * 0.5f;
    out[p * 3 + 1] = (1.f - n1) * 0.5f;
    out[p * 3 + 2] = (1.f - n2) * 0.5f;
}

// The cosine term of get_loss_dict AS THE REFERENCE CALLS IT (dn_model.py:807-813): metrics.mean_angular_error
// (metrics.py:58-73, written for [B,3,H,W]) receives [3,H,W] tensors, so its sum(dim=1) runs over the image ROWS: one
// "dot product" per (channel, column), s = sum_y ((g - 1)/2)((n - 1)/2) with n = normal * mask, g = the target (mono: *
// mask as well), and the loss is mean over the 3 W columns of acos(clamp(s, -1, 1)).  One thread per (column, channel).
// partial[ceil(3W/256), 2] col 0 = sum of the acos values; v_loss != NULL: v_normal += v_loss[0] * w * d/dn (the clamp
// passes gradients on [-1, 1] inclusive, as torch.clamp does; acos' own derivative is -1/sqrt(1 - s^2)).
template <bool GRADS>
__global__ void __launch_bounds__(256)
normal_cosine_kernel(int H, int W, const float *__restrict__ normal, const float *__restrict__ normal_gt,
                     const float *__restrict__ mask, int gt_unmasked, float w, const float *__restrict__ v_loss,
                     float *__restrict__ partial, float *__restrict__ v_normal) {
    __shared__ float red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;  // = x * 3 + c
    float val = 0.f;
    if (i < 3 * W) {
        const int x = i / 3;
        float s = 0.f;
        for (int y = 0; y < H; ++y) {
            const int64_t q = ((int64_t)y * W + x);
            const float m = mask ? mask[q] : 1.f;
            const float n = normal[q * 3 + (i - 3 * x)] * m;
            const float g = normal_gt[q * 3 + (i - 3 * x)] * (gt_unmasked ? 1.f : m);
            s += ((g - 1.f) * 0.5f) * ((n - 1.f) * 0.5f);
        }
        const float sc = fminf(1.f, fmaxf(-1.f, s));
        val = acosf(sc);
        if (GRADS) {
            const float dacos = (s >= -1.f && s <= 1.f) ? -1.f / sqrtf(1.f - sc * sc) : 0.f;
            const float up = v_loss[0] * w * dacos;
            if (up != 0.f) {
                for (int y = 0; y < H; ++y) {
                    const int64_t q = ((int64_t)y * W + x);
                    const float m = mask ? mask[q] : 1.f;
                    const float g = normal_gt[q * 3 + (i - 3 * x)] * (gt_unmasked ? 1.f : m);
                    v_normal[q * 3 + (i - 3 * x)] += up * ((g - 1.f) * 0.5f) * 0.5f * m;
                }
            }
        }
    }
    val = wave_sum_to_last_row(val);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = val;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x + 0] = red[0] + red[1] + red[2] + red[3];
        partial[2 * blockIdx.x + 1] = 0.f;
    }
}

__global__ void __launch_bounds__(256)
depth_valid_counts_kernel(int H, int W, const float *__restrict__ sensor, const float *__restrict__ mask, float tol,
                          unsigned long long *__restrict__ counts) {
    const int64_t P = (int64_t)H * W;
    unsigned cx = 0, cy = 0, ca = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        const float g = sensor[p] * (mask ? mask[p] : 1.f);
        if (g > tol) {
            const int x = (int)(p % W), y = (int)(p / W);
            cx += (x + 1 < W);
            cy += (y + 1 < H);
            ca += 1;
        }
    }
    // integer wave sums (exact), one atomic triple per wave
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        cx += __shfl_xor(cx, s, 64);
        cy += __shfl_xor(cy, s, 64);
        ca += __shfl_xor(ca, s, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[0], (unsigned long long)cx);
        atomicAdd(&counts[1], (unsigned long long)cy);
        atomicAdd(&counts[2], (unsigned long long)ca);
    }
}

template <bool GRADS>
__global__ void __launch_bounds__(256)
min_scale_kernel(int N, const float *__restrict__ log_scales, float g, const float *__restrict__ v_loss,
                 float *__restrict__ partial, float *__restrict__ g_log_scales) {
    __shared__ float red[4];
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.f;
    if (n < N) {
        const float s0 = log_scales[n * 3 + 0], s1 = log_scales[n * 3 + 1], s2 = log_scales[n * 3 + 2];
        int k = 0;
        float sm = s0;
        if (s1 < sm) { sm = s1; k = 1; }
        if (s2 < sm) { sm = s2; k = 2; }
        v = expf(sm);  // min exp = exp min (monotone)
        // (an explicit fused multiply-add: fsgs_gauss_sh_bwd_adam folds this term into its launch and must agree to the bit)
        if (GRADS) g_log_scales[n * 3 + k] = __builtin_fmaf(v_loss[0] * g, v, g_log_scales[n * 3 + k]);
    }
    v = wave_sum_to_last_row(v);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x + 0] = red[0] + red[1] + red[2] + red[3];
        partial[2 * blockIdx.x + 1] = 0.f;
    }
}

__global__ void __launch_bounds__(256)
touch_normal_sqerr_kernel(int n, const int64_t *__restrict__ idx, const float *__restrict__ normals_world,
                          const float *__restrict__ touch_normals, float *__restrict__ partial) {
    __shared__ float red[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = 0.f;
    if (i < n) {
        const int64_t r = idx[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float e = normals_world[r * 3 + k] - touch_normals[(int64_t)i * 3 + k];
            v += e * e;
        }
    }
    v = wave_sum_to_last_row(v);
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x + 0] = red[0] + red[1] + red[2] + red[3];
        partial[2 * blockIdx.x + 1] = 0.f;
    }
}

}  // namespace fsgs

using namespace fsgs;

extern "C" int64_t fsgs_fusion_aux_num_partials(int H, int W) {
    return (int64_t)ceil_div(W, 32) * ceil_div(H, 8);
}

extern "C" int fsgs_fusion_aux_loss(int H, int W, const float *depth, const float *normal, const float *image,
                                    const float *sensor_depth, const float *normal_gt, const float *mask,
                                    float depth_tol, const float *w, const float *v_loss, float *partial,
                                    float *v_depth, float *v_normal, int flags, fsgs_stream_t stream) {
    return fsgs_fusion_aux_loss_riders(H, W, depth, normal, image, sensor_depth, normal_gt, mask, depth_tol, w, v_loss,
                                       partial, v_depth, v_normal, flags, 0, nullptr, nullptr, 0, nullptr, nullptr,
                                       nullptr, nullptr, stream);
}

extern "C" int fsgs_fusion_aux_loss_riders(int H, int W, const float *depth, const float *normal, const float *image,
                                           const float *sensor_depth, const float *normal_gt, const float *mask,
                                           float depth_tol, const float *w, const float *v_loss, float *partial,
                                           float *v_depth, float *v_normal, int flags, int N, const float *log_scales,
                                           float *min_scale_partial, int n_touch, const int64_t *touch_idx,
                                           const float *normals_world, const float *touch_normals,
                                           float *touch_partial, fsgs_stream_t stream) {
    if (H < 0 || W < 0 || flags < 0 || flags > 15 || N < 0 || n_touch < 0) return FSGS_EINVAL;
    if (H == 0 || W == 0) return (N > 0 || n_touch > 0) ? FSGS_EINVAL : FSGS_OK;  // (riders need an image's launch)
    if (!depth || !image || !sensor_depth || !w || !partial) return FSGS_EINVAL;
    if (normal_gt && !normal) return FSGS_EINVAL;
    if (v_loss && (!v_depth || (normal && !v_normal))) return FSGS_EINVAL;
    if (N > 0 && (!log_scales || !min_scale_partial)) return FSGS_EINVAL;
    if (n_touch > 0 && (!touch_idx || !normals_world || !touch_normals || !touch_partial)) return FSGS_EINVAL;
    FusionAuxW ww;
    for (int k = 0; k < 7; ++k) ww.w[k] = w[k];
    ww.w[7] = 0.f;
    const int gx = (int)ceil_div(W, 32), gy = (int)ceil_div(H, 8);
    FusionAuxRiders rd = {gy, N, (int)ceil_div(N, 256), log_scales, min_scale_partial, n_touch, (int)ceil_div(n_touch, 256),
                          touch_idx, normals_world, touch_normals, touch_partial};
    const dim3 grid(gx, gy + (int)ceil_div(rd.nm + rd.nt, gx));
    if (v_loss)
        hipLaunchKernelGGL(fusion_aux_kernel<true>, grid, dim3(256), 0, as_stream(stream), H, W, depth, normal, image,
                           sensor_depth, normal_gt, mask, depth_tol, ww, v_loss, partial, v_depth, v_normal, flags, rd);
    else
        hipLaunchKernelGGL(fusion_aux_kernel<false>, grid, dim3(256), 0, as_stream(stream), H, W, depth, normal, image,
                           sensor_depth, normal_gt, mask, depth_tol, ww, nullptr, partial, nullptr, nullptr, flags, rd);
    return check_launch();
}

extern "C" int fsgs_normals_from_depth(int H, int W, const float *depth, const float *mask, float fx, float fy, float cx,
                                       float cy, float *normal_out, fsgs_stream_t stream) {
    if (H < 0 || W < 0) return FSGS_EINVAL;
    if (H == 0 || W == 0) return FSGS_OK;
    if (!depth || !normal_out || fx == 0.f || fy == 0.f) return FSGS_EINVAL;
    hipLaunchKernelGGL(normals_from_depth_kernel, dim3(ceil_div(W, 32), ceil_div(H, 8)), dim3(256), 0, as_stream(stream),
                       H, W, depth, mask, fx, fy, cx, cy, normal_out);
    return check_launch();
}

extern "C" int64_t fsgs_normal_cosine_num_partials(int W) { return ceil_div(3 * W, 256); }

extern "C" int fsgs_normal_cosine_loss(int H, int W, const float *normal, const float *normal_gt, const float *mask,
                                       int gt_unmasked, float w, const float *v_loss, float *partial, float *v_normal,
                                       fsgs_stream_t stream) {
    if (H < 0 || W < 0) return FSGS_EINVAL;
    if (H == 0 || W == 0) return FSGS_OK;
    if (!normal || !normal_gt || !partial || (v_loss && !v_normal)) return FSGS_EINVAL;
    const dim3 grid(ceil_div(3 * W, 256));
    if (v_loss)
        hipLaunchKernelGGL(normal_cosine_kernel<true>, grid, dim3(256), 0, as_stream(stream), H, W, normal, normal_gt, mask,
                           gt_unmasked, w, v_loss, partial, v_normal);
    else
        hipLaunchKernelGGL(normal_cosine_kernel<false>, grid, dim3(256), 0, as_stream(stream), H, W, normal, normal_gt,
                           mask, gt_unmasked, w, nullptr, partial, nullptr);
    return check_launch();
}

extern "C" int fsgs_depth_valid_counts(int H, int W, const float *sensor_depth, const float *mask, float depth_tol,
                                       int64_t *counts, fsgs_stream_t stream) {
    if (H < 0 || W < 0 || !counts) return FSGS_EINVAL;
    hipStream_t s = as_stream(stream);
    hipError_t e = hipMemsetAsync(counts, 0, 3 * sizeof(int64_t), s);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return FSGS_ELAUNCH; }
    if (H == 0 || W == 0) return FSGS_OK;
    if (!sensor_depth) return FSGS_EINVAL;
    const int64_t P = (int64_t)H * W;
    const int blocks = (int)((P + 255) / 256 < 1024 ? (P + 255) / 256 : 1024);
    hipLaunchKernelGGL(depth_valid_counts_kernel, dim3(blocks), dim3(256), 0, s, H, W, sensor_depth, mask, depth_tol,
                       reinterpret_cast<unsigned long long *>(counts));
    return check_launch();
}

extern "C" int fsgs_min_scale_loss(int N, const float *log_scales, float g, const float *v_loss, float *partial,
                                   float *g_log_scales, fsgs_stream_t stream) {
    if (N < 0) return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!log_scales || !partial || (v_loss && !g_log_scales)) return FSGS_EINVAL;
    if (v_loss)
        hipLaunchKernelGGL(min_scale_kernel<true>, dim3(ceil_div(N, 256)), dim3(256), 0, as_stream(stream), N,
                           log_scales, g, v_loss, partial, g_log_scales);
    else
        hipLaunchKernelGGL(min_scale_kernel<false>, dim3(ceil_div(N, 256)), dim3(256), 0, as_stream(stream), N,
                           log_scales, g, nullptr, partial, nullptr);
    return check_launch();
}

extern "C" int fsgs_touch_normal_sqerr(int n, const int64_t *idx, const float *normals_world,
                                       const float *touch_normals, float *partial, fsgs_stream_t stream) {
    if (n < 0) return FSGS_EINVAL;
    if (n == 0) return FSGS_OK;
    if (!idx || !normals_world || !touch_normals || !partial) return FSGS_EINVAL;
    hipLaunchKernelGGL(touch_normal_sqerr_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), n, idx,
                       normals_world, touch_normals, partial);
    return check_launch();
}
