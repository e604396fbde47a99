// a-11, a-12, a-13: the per-Gaussian kernels around the rasterizer:
//   * per-Gaussian normals as /root/reference/dn_splatter/dn_model.py:618-636 computes them
//     (~8 torch kernels there, one here) and their VJP into the quaternions;
//   * SplatfactoModel.after_train statistics (registered at dn_model.py:1385-1389);
//   * the row movement behind refinement_after (dn_model.py:326-451): order-preserving
//     compaction of parameter / Adam-moment rows, and the split sampler of split_gaussians.
// All HBM-bound, one thread per Gaussian (or per float4 of a row), coalesced SoA access.
#include "common.h"
#include "scan.h"

namespace fsgs {

struct NormalCtx {
    float n[3];      // world normal after flip, unit length
    float sign;      // +1 / -1 flip
    int axis;
    float inv_qn;    // 1/|q|
    float qw, qx, qy, qz;  // unit quaternion
};

__device__ __forceinline__ void normal_forward(const float4 q, const float *ls, const float *mean,
                                               const float *c2w, NormalCtx &o) {
    const float qn = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    o.inv_qn = 1.f / qn;
    o.qw = q.x * o.inv_qn; o.qx = q.y * o.inv_qn; o.qy = q.z * o.inv_qn; o.qz = q.w * o.inv_qn;
    // torch.argmin returns the first minimal index
    int a = 0;
    if (ls[1] < ls[a]) a = 1;
    if (ls[2] < ls[a]) a = 2;
    o.axis = a;
    const float w = o.qw, x = o.qx, y = o.qy, z = o.qz;
    float c0, c1, c2;  // column `a` of R(q)
    if (a == 0) { c0 = 1.f - 2.f * (y * y + z * z); c1 = 2.f * (x * y + w * z); c2 = 2.f * (x * z - w * y); }
    else if (a == 1) { c0 = 2.f * (x * y - w * z); c1 = 1.f - 2.f * (x * x + z * z); c2 = 2.f * (y * z + w * x); }
    else { c0 = 2.f * (x * z + w * y); c1 = 2.f * (y * z - w * x); c2 = 1.f - 2.f * (x * x + y * y); }
    const float inv = 1.f / fmaxf(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), 1e-12f);
    c0 *= inv; c1 *= inv; c2 *= inv;
    // view direction = campos - mean (its normalisation does not change the sign of the dot)
    const float vx = c2w[3] - mean[0], vy = c2w[7] - mean[1], vz = c2w[11] - mean[2];
    o.sign = (c0 * vx + c1 * vy + c2 * vz < 0.f) ? -1.f : 1.f;
    o.n[0] = o.sign * c0; o.n[1] = o.sign * c1; o.n[2] = o.sign * c2;
}

__global__ void __launch_bounds__(256)
normals_fwd_kernel(int N, const float *__restrict__ quats, const float *__restrict__ log_scales,
                   const float *__restrict__ means, const float *__restrict__ c2w,
                   float *__restrict__ normals_world, float *__restrict__ normals_cam) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float4 q = reinterpret_cast<const float4 *>(quats)[n];
    const float ls[3] = {log_scales[n * 3], log_scales[n * 3 + 1], log_scales[n * 3 + 2]};
    const float mean[3] = {means[n * 3], means[n * 3 + 1], means[n * 3 + 2]};
    NormalCtx o;
    normal_forward(q, ls, mean, c2w, o);
#pragma unroll
    for (int k = 0; k < 3; ++k) normals_world[n * 3 + k] = o.n[k];
    // n_cam = n @ c2w[:3,:3]  ->  n_cam[j] = sum_i n[i] * R[i][j]
#pragma unroll
    for (int j = 0; j < 3; ++j)
        normals_cam[n * 3 + j] = o.n[0] * c2w[0 * 4 + j] + o.n[1] * c2w[1 * 4 + j] + o.n[2] * c2w[2 * 4 + j];
}

__global__ void __launch_bounds__(256)
normals_bwd_kernel(int N, const float *__restrict__ quats, const float *__restrict__ log_scales,
                   const float *__restrict__ means, const float *__restrict__ c2w,
                   const float *__restrict__ v_normals_cam, float *__restrict__ v_quats) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float4 q = reinterpret_cast<const float4 *>(quats)[n];
    const float ls[3] = {log_scales[n * 3], log_scales[n * 3 + 1], log_scales[n * 3 + 2]};
    const float mean[3] = {means[n * 3], means[n * 3 + 1], means[n * 3 + 2]};
    NormalCtx o;
    normal_forward(q, ls, mean, c2w, o);
    // v_n(world) = R_c2w v_ncam ; undo flip
    float vn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        vn[i] = o.sign * (c2w[i * 4 + 0] * v_normals_cam[n * 3 + 0] + c2w[i * 4 + 1] * v_normals_cam[n * 3 + 1] +
                          c2w[i * 4 + 2] * v_normals_cam[n * 3 + 2]);
    // through F.normalize of the column (column of a rotation: |c| = 1 up to rounding)
    const float cs[3] = {o.sign * o.n[0], o.sign * o.n[1], o.sign * o.n[2]};
    const float dp = vn[0] * cs[0] + vn[1] * cs[1] + vn[2] * cs[2];
    const float vc[3] = {vn[0] - dp * cs[0], vn[1] - dp * cs[1], vn[2] - dp * cs[2]};
    // column `a` of R(q_hat) -> q_hat
    const float w = o.qw, x = o.qx, y = o.qy, z = o.qz;
    float gw, gx, gy, gz;
    if (o.axis == 0) {
        gw = 2.f * (z * vc[1] - y * vc[2]);
        gx = 2.f * (y * vc[1] + z * vc[2]);
        gy = 2.f * (-2.f * y * vc[0] + x * vc[1] - w * vc[2]);
        gz = 2.f * (-2.f * z * vc[0] + w * vc[1] + x * vc[2]);
    } else if (o.axis == 1) {
        gw = 2.f * (-z * vc[0] + x * vc[2]);
        gx = 2.f * (y * vc[0] - 2.f * x * vc[1] + w * vc[2]);
        gy = 2.f * (x * vc[0] + z * vc[2]);
        gz = 2.f * (-w * vc[0] - 2.f * z * vc[1] + y * vc[2]);
    } else {
        gw = 2.f * (y * vc[0] - x * vc[1]);
        gx = 2.f * (z * vc[0] - w * vc[1] - 2.f * x * vc[2]);
        gy = 2.f * (w * vc[0] + z * vc[1] - 2.f * y * vc[2]);
        gz = 2.f * (x * vc[0] + y * vc[1]);
    }
    const float d = gw * w + gx * x + gy * y + gz * z;
    reinterpret_cast<float4 *>(v_quats)[n] = make_float4((gw - d * w) * o.inv_qn, (gx - d * x) * o.inv_qn,
                                                         (gy - d * y) * o.inv_qn, (gz - d * z) * o.inv_qn);
}

__global__ void __launch_bounds__(256)
densify_stats_kernel(int N, const int32_t *__restrict__ radii, const float *__restrict__ absgrad,
                     float inv_max_hw, float *__restrict__ xys_grad_norm,
                     float *__restrict__ vis_counts, float *__restrict__ max_2Dsize) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int r = radii[n];
    if (r <= 0) return;
    const float2 g = reinterpret_cast<const float2 *>(absgrad)[n];
    xys_grad_norm[n] += sqrtf(g.x * g.x + g.y * g.y);
    vis_counts[n] += 1.f;
    max_2Dsize[n] = fmaxf(max_2Dsize[n], (float)r * inv_max_hw);
}

// One thread per float of the destination row set: thread -> (row, col); rows move as units.
__global__ void __launch_bounds__(256)
compact_rows_kernel(int64_t n_rows, int row_floats, const uint8_t *__restrict__ keep,
                    const int64_t *__restrict__ positions, const float *__restrict__ src,
                    float *__restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = e / row_floats;
    if (row >= n_rows) return;
    if (!keep[row]) return;
    const int col = (int)(e - row * row_floats);
    dst[positions[row] * row_floats + col] = src[e];
}

__global__ void __launch_bounds__(256)
split_samples_kernel(int64_t S, int n_samples, const int64_t *__restrict__ ids,
                     const float *__restrict__ means, const float *__restrict__ quats,
                     const float *__restrict__ log_scales, const float *__restrict__ randn,
                     float *__restrict__ new_means, float *__restrict__ new_log_scales) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= S * n_samples) return;
    const int64_t i = e % S;  // sample-major output: row = s*S + i
    const int64_t p = ids[i];
    const float4 q = reinterpret_cast<const float4 *>(quats)[p];
    const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    const float w = q.x * inv, x = q.y * inv, y = q.z * inv, z = q.w * inv;
    const float ls[3] = {log_scales[p * 3], log_scales[p * 3 + 1], log_scales[p * 3 + 2]};
    const float v[3] = {expf(ls[0]) * randn[e * 3], expf(ls[1]) * randn[e * 3 + 1], expf(ls[2]) * randn[e * 3 + 2]};
    const float R[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - w * z), 2.f * (x * z + w * y)},
                           {2.f * (x * y + w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - w * x)},
                           {2.f * (x * z - w * y), 2.f * (y * z + w * x), 1.f - 2.f * (x * x + y * y)}};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        new_means[e * 3 + r] = R[r][0] * v[0] + R[r][1] * v[1] + R[r][2] * v[2] + means[p * 3 + r];
        new_log_scales[e * 3 + r] = logf(expf(ls[r]) / 1.6f);
    }
}

}  // namespace fsgs

using namespace fsgs;

extern "C" int fsgs_normals_fwd(int N, const float *quats, const float *log_scales, const float *means,
                                const float *c2w, float *normals_world, float *normals_cam,
                                fsgs_stream_t stream) {
    if (N < 0) return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!quats || !log_scales || !means || !c2w || !normals_world || !normals_cam) return FSGS_EINVAL;
    hipLaunchKernelGGL(normals_fwd_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, as_stream(stream), N,
                       quats, log_scales, means, c2w, normals_world, normals_cam);
    return check_launch();
}

extern "C" int fsgs_normals_bwd(int N, const float *quats, const float *log_scales, const float *means,
                                const float *c2w, const float *v_normals_cam, float *v_quats,
                                fsgs_stream_t stream) {
    if (N < 0) return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!quats || !log_scales || !means || !c2w || !v_normals_cam || !v_quats) return FSGS_EINVAL;
    hipLaunchKernelGGL(normals_bwd_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, as_stream(stream), N,
                       quats, log_scales, means, c2w, v_normals_cam, v_quats);
    return check_launch();
}

extern "C" int fsgs_densify_stats(int N, const int32_t *radii, const float *absgrad, float inv_max_hw,
                                  float *xys_grad_norm, float *vis_counts, float *max_2Dsize,
                                  fsgs_stream_t stream) {
    if (N < 0) return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!radii || !absgrad || !xys_grad_norm || !vis_counts || !max_2Dsize) return FSGS_EINVAL;
    hipLaunchKernelGGL(densify_stats_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, as_stream(stream), N,
                       radii, absgrad, inv_max_hw, xys_grad_norm, vis_counts, max_2Dsize);
    return check_launch();
}

extern "C" int fsgs_mask_scan(int64_t n_rows, const uint8_t *keep, int64_t *positions, void *scratch,
                              size_t scratch_bytes, fsgs_stream_t stream) {
    if (n_rows < 0) return FSGS_EINVAL;
    if (n_rows == 0) return FSGS_OK;
    if (!keep || !positions) return FSGS_EINVAL;
    return device_scan<uint8_t, false>(n_rows, keep, positions, scratch, scratch_bytes, as_stream(stream));
}

extern "C" int fsgs_compact_rows(int64_t n_rows, int row_floats, const uint8_t *keep,
                                 const int64_t *positions, const float *src, float *dst,
                                 fsgs_stream_t stream) {
    if (n_rows < 0 || row_floats < 1) return FSGS_EINVAL;
    if (n_rows == 0) return FSGS_OK;
    if (!keep || !positions || !src || !dst) return FSGS_EINVAL;
    const int64_t total = n_rows * row_floats;
    hipLaunchKernelGGL(compact_rows_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream),
                       n_rows, row_floats, keep, positions, src, dst);
    return check_launch();
}

extern "C" int fsgs_split_samples(int64_t S, int n_samples, const int64_t *ids, const float *means,
                                  const float *quats, const float *log_scales, const float *randn,
                                  float *new_means, float *new_log_scales, fsgs_stream_t stream) {
    if (S < 0 || n_samples < 1) return FSGS_EINVAL;
    if (S == 0) return FSGS_OK;
    if (!ids || !means || !quats || !log_scales || !randn || !new_means || !new_log_scales) return FSGS_EINVAL;
    hipLaunchKernelGGL(split_samples_kernel, dim3(ceil_div(S * n_samples, 256)), dim3(256), 0,
                       as_stream(stream), S, n_samples, ids, means, quats, log_scales, randn, new_means,
                       new_log_scales);
    return check_launch();
}
