// a-11, a-12, a-13: the per-Gaussian kernels around the rasterizer:
//   * per-Gaussian normals as /root/reference/dn_splatter/dn_model.py:618-636 computes them
//     (~8 torch kernels there, one here) and their VJP into the quaternions;
//   * SplatfactoModel.after_train statistics (registered at dn_model.py:1385-1389);
//   * the row movement behind refinement_after (dn_model.py:326-451): order-preserving
//     compaction of parameter / Adam-moment rows, and the split sampler of split_gaussians.
// All HBM-bound, one thread per Gaussian (or per float4 of a row), coalesced SoA access.
#include "common.h"
#include "normal_math.h"
#include "scan.h"

namespace fsgs {

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
    float ex[3];
    normal_to_camera(o.n, c2w, ex);
#pragma unroll
    for (int j = 0; j < 3; ++j) normals_cam[n * 3 + j] = ex[j];
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
    const float vn[3] = {v_normals_cam[n * 3 + 0], v_normals_cam[n * 3 + 1], v_normals_cam[n * 3 + 2]};
    reinterpret_cast<float4 *>(v_quats)[n] = normal_backward(q, ls, mean, c2w, vn);
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

// The same for up to kCompactMaxJobs tensors in ONE launch (a refinement compacts six parameters, their twelve Adam
// moments and the six blocks of new rows: 24 launches with their host overhead were a third of a rebuild).  Every job has
// its own mask / positions / row width; consecutive workgroups take consecutive 256-float pieces of job 0, then job 1, ...
constexpr int kCompactMaxJobs = 32;
struct CompactJobs {
    const float *src[kCompactMaxJobs];
    float *dst[kCompactMaxJobs];
    const uint8_t *keep[kCompactMaxJobs];
    const int64_t *positions[kCompactMaxJobs];
    int64_t n_rows[kCompactMaxJobs];
    int64_t first_block[kCompactMaxJobs + 1];
    int row_floats[kCompactMaxJobs];
    int n_jobs;
};

__global__ void __launch_bounds__(256) compact_rows_multi_kernel(CompactJobs J) {
    int j = 0;
    while (j + 1 < J.n_jobs && (int64_t)blockIdx.x >= J.first_block[j + 1]) ++j;  // (wave-uniform: scalar loads)
    const int64_t e = ((int64_t)blockIdx.x - J.first_block[j]) * 256 + threadIdx.x;
    const int rf = J.row_floats[j];
    const int64_t row = e / rf;
    if (row >= J.n_rows[j] || !J.keep[j][row]) return;
    const int col = (int)(e - row * rf);
    J.dst[j][J.positions[j][row] * rf + col] = J.src[j][e];
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

// a-14: the union of oriented-box tests of touch_pruning / add_touch_patch (points_in_non_aabb per patch,
// /root/reference/dn_splatter/dn_model.py:1173-1184, 1284-1294, 1996-2034) in ONE launch: a box is 18 floats — centre,
// three unit axes (rows), lower and upper bounds of the vertices' coordinates along them (fusionsense_amd/touch.py:
// box_frames, computed once per run) —, a point is inside when lo <= (p - centre) . axis <= hi on all three, bounds
// inclusive.  (The reference and rounds 1-4 ran ~15 small torch operators per patch: 1.2 ms of host time per refinement.)
__global__ void __launch_bounds__(256)
points_in_boxes_kernel(int64_t n, const float *__restrict__ pts, int n_boxes, const float *__restrict__ boxes,
                       uint8_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pts[i * 3 + 0], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    bool any = false;
    for (int b = 0; b < n_boxes; ++b) {
        const float *B = boxes + 18 * b;
        const float dx = x - B[0], dy = y - B[1], dz = z - B[2];
        bool in = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float r = fmaf(dz, B[3 + 3 * k + 2], fmaf(dy, B[3 + 3 * k + 1], dx * B[3 + 3 * k + 0]));
            in = in && (r >= B[12 + k]) && (r <= B[15 + k]);
        }
        any = any || in;
    }
    out[i] = any ? 1 : 0;
}

// a-14: nearest point of a set for every query (hull_pruning's cdist(...).min and add_touch_patch's 1-NN colour
// lookup, /root/reference/dn_splatter/dn_model.py:1181-1182, 1258-1264), brute force: a thread keeps its query in
// registers, a slice of the point set streams through LDS in tiles of 1024 (broadcast reads).  Exact fp32 differences
// (no |a|^2 + |b|^2 - 2ab expansion), ties -> lowest index.
// The point set is SLICED over gridDim.y (round 5: with one workgroup per 256 queries the 2 000 points of a touch patch
// against 60 000 Gaussians ran on 8 of 256 CUs for 0.72 ms): every (query block, slice) workgroup folds its best
// (squared distance, index) into the query's 64-bit word of ``best`` — distance bits above index bits, so that the
// unsigned minimum IS the nearest point with the lowest index on ties (squared distances are >= +0: their bit patterns
// order like the values) —, and nearest_point_unpack_kernel turns the words into the two outputs.
// WANT_IDX = false (hull_pruning needs distances only): the running minimum is one v_min instead of a compare and two
// selects.  ``active`` (nullable): queries whose byte is 0 are skipped — a wave without an active query leaves at once
// (hull_pruning only asks for the rows near the object; their distance reads +inf, their index 0).
template <bool WANT_IDX>
__global__ void __launch_bounds__(256)
nearest_point_kernel(int nq, const float *__restrict__ q, int np, const float *__restrict__ p, int tiles_per_slice,
                     const uint8_t *__restrict__ active, float4 close, unsigned long long *__restrict__ best_out) {
    __shared__ float4 sp[1024];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool on = i < nq && (!active || active[i]);
    const float qx = on ? q[i * 3 + 0] : 0.f, qy = on ? q[i * 3 + 1] : 0.f, qz = on ? q[i * 3 + 2] : 0.f;
    if (close.w >= 0.f) {  // (fsgs_nearest_point_words: only the queries within close.w of close.xyz are searched for)
        const float dx = qx - close.x, dy = qy - close.y, dz = qz - close.z;
        on = on && sqrtf(dx * dx + dy * dy + dz * dz) <= close.w;
    }
    if (!__syncthreads_or(on)) return;
    float best = INFINITY;
    int best_j = 0;
    const int lo = (int)blockIdx.y * tiles_per_slice * 1024;
    const int hi = min(np, lo + tiles_per_slice * 1024);
    for (int base = lo; base < hi; base += 1024) {
        const int n = min(1024, hi - base);
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += 256) {
            const float *s = p + (int64_t)(base + k) * 3;
            sp[k] = make_float4(s[0], s[1], s[2], 0.f);
        }
        __syncthreads();
        if (!__any(on)) continue;
        for (int j = 0; j < n; ++j) {
            const float4 c = sp[j];
            const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
            const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));  // (spelled out: the same bits in both instantiations)
            if (WANT_IDX) {
                if (d2 < best) { best = d2; best_j = base + j; }
            } else {
                best = fminf(best, d2);  // (fminf drops a NaN operand, like the comparison above never takes one)
            }
        }
    }
    if (on && best < INFINITY)
        atomicMin(best_out + i, ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)best_j);
}

__global__ void __launch_bounds__(256)
nearest_point_unpack_kernel(int nq, int64_t *__restrict__ best_idx, float *__restrict__ out_dist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const unsigned long long w = reinterpret_cast<const unsigned long long *>(best_idx)[i];
    // (no point compared below +inf — NaN coordinates, or all of them infinitely far: distance inf, index 0)
    const bool none = w == ~0ull;
    if (out_dist) out_dist[i] = none ? INFINITY : sqrtf(__uint_as_float((unsigned)(w >> 32)));
    best_idx[i] = none ? 0 : (int64_t)(w & 0xFFFFFFFFull);
}

// The k nearest points of a set for every query (compute_level_surface_points' knn_sk, dn_model.py:1762-1764, and
// sklearn's kneighbors behind it): brute force like nearest_point_kernel — the point set streams through LDS in tiles of
// 1024 (broadcast reads), a thread keeps its query and a sorted list of its KK best (distance, index) pairs in
// registers; a candidate is first tested against the list's worst entry (one compare: after the first few hundred points
// almost every candidate fails it) and only then sifted in through an unrolled compare-exchange chain.  Exact fp32
// differences, every product and sum rounded on its own (no contraction: the order of torch's elementwise formulation);
// ties -> lowest index.  out_idx [nq, KK - skip]: the neighbours ranked skip .. KK - 1 (skip = 1 is knn_sk's
// "drop the first column").  640 k queries x 300 k points (one 800 x 800 frame against config #2): O(100 ms) instead of
// the ~10 s of a KD-tree on the host.
template <int KK>
__global__ void __launch_bounds__(256)
knn_points_kernel(int64_t nq, const float *__restrict__ q, int np, const float *__restrict__ p, int kk, int skip,
                  int64_t *__restrict__ out_idx) {
    __shared__ float sp[1024 * 3];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = i < nq;
    const float qx = on ? q[i * 3 + 0] : 0.f, qy = on ? q[i * 3 + 1] : 0.f, qz = on ? q[i * 3 + 2] : 0.f;
    float bd[KK];
    int bi[KK];
#pragma unroll
    for (int s = 0; s < KK; ++s) { bd[s] = INFINITY; bi[s] = 0; }
    // (lists shorter than KK: the entries beyond kk are never reported; the worst REPORTED entry is slot kk - 1)
    for (int base = 0; base < np; base += 1024) {
        const int n = min(1024, np - base);
        __syncthreads();
        for (int k = threadIdx.x; k < n * 3; k += 256) sp[k] = p[(int64_t)base * 3 + k];
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const float dx = qx - sp[j * 3 + 0], dy = qy - sp[j * 3 + 1], dz = qz - sp[j * 3 + 2];
            const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            if (d2 < bd[KK - 1]) {
                float cd = d2;
                int ci = base + j;
                bool inserted = false;
#pragma unroll
                for (int s = 0; s < KK; ++s) {  // sift in: strict <, so equal distances keep the earlier index in front;
                    const bool before = inserted || cd < bd[s];  // once inserted, the rest of the list shifts down
                    inserted = before;
                    const float td = before ? bd[s] : cd;
                    const int ti = before ? bi[s] : ci;
                    bd[s] = before ? cd : bd[s];
                    bi[s] = before ? ci : bi[s];
                    cd = td; ci = ti;
                }
            }
        }
    }
    if (on) {
#pragma unroll
        for (int s = 0; s < KK; ++s)
            if (s >= skip && s < kk) out_idx[i * (kk - skip) + (s - skip)] = bi[s];
    }
}

}  // namespace fsgs

using namespace fsgs;

extern "C" int fsgs_knn_points(int64_t nq, const float *queries, int np, const float *points, int k, int skip,
                               int64_t *out_idx, fsgs_stream_t stream) {
    // k = neighbours searched per query (<= 33, <= np), skip = leading ones not reported (0 or 1): out_idx [nq, k - skip]
    if (nq < 0 || np < 1 || k < 1 || k > 33 || k > np || skip < 0 || skip >= k) return FSGS_EINVAL;
    if (nq == 0) return FSGS_OK;
    if (!queries || !points || !out_idx) return FSGS_EINVAL;
    const dim3 grid((unsigned)((nq + 255) / 256));
    hipStream_t s = as_stream(stream);
    // (the list is kept at its compiled length; with k below it the worst kept entry is still slot KK - 1, so the list
    // holds the KK best and the first k of them are the k best)
    if (k <= 17) hipLaunchKernelGGL(knn_points_kernel<17>, grid, dim3(256), 0, s, nq, queries, np, points, k, skip, out_idx);
    else hipLaunchKernelGGL(knn_points_kernel<33>, grid, dim3(256), 0, s, nq, queries, np, points, k, skip, out_idx);
    return check_launch();
}

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

extern "C" int fsgs_compact_rows_multi(int n_jobs, const int64_t *n_rows, const int *row_floats,
                                       const uint8_t *const *keep, const int64_t *const *positions,
                                       const float *const *src, float *const *dst, fsgs_stream_t stream) {
    if (n_jobs < 0 || n_jobs > kCompactMaxJobs) return FSGS_EINVAL;
    if (n_jobs == 0) return FSGS_OK;
    if (!n_rows || !row_floats || !keep || !positions || !src || !dst) return FSGS_EINVAL;
    CompactJobs J;
    J.n_jobs = 0;
    int64_t blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (n_rows[j] < 0 || row_floats[j] < 1) return FSGS_EINVAL;
        if (n_rows[j] == 0) continue;
        if (!keep[j] || !positions[j] || !src[j] || !dst[j]) return FSGS_EINVAL;
        const int k = J.n_jobs++;
        J.src[k] = src[j]; J.dst[k] = dst[j]; J.keep[k] = keep[j]; J.positions[k] = positions[j];
        J.n_rows[k] = n_rows[j]; J.row_floats[k] = row_floats[j];
        J.first_block[k] = blocks;
        blocks += ceil_div(n_rows[j] * row_floats[j], 256);
    }
    if (J.n_jobs == 0) return FSGS_OK;
    J.first_block[J.n_jobs] = blocks;
    if (blocks > 0x7FFFFFFFll) return FSGS_EINVAL;
    hipLaunchKernelGGL(compact_rows_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), J);
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

extern "C" int fsgs_points_in_boxes(int64_t n, const float *points, int n_boxes, const float *boxes, uint8_t *out_mask,
                                    fsgs_stream_t stream) {
    if (n < 0 || n_boxes < 0) return FSGS_EINVAL;
    if (n == 0) return FSGS_OK;
    if (!points || !out_mask || (n_boxes > 0 && !boxes)) return FSGS_EINVAL;
    hipLaunchKernelGGL(points_in_boxes_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), n, points, n_boxes,
                       boxes, out_mask);
    return check_launch();
}

extern "C" int fsgs_nearest_point(int nq, const float *queries, int np, const float *points, const uint8_t *active,
                                  float *out_dist, int64_t *out_idx, int want_idx, fsgs_stream_t stream) {
    if (nq < 0 || np < 1) return FSGS_EINVAL;
    if (nq == 0) return FSGS_OK;
    if (!queries || !points || !out_idx) return FSGS_EINVAL;  // (out_idx doubles as the slices' meeting place)
    const int q_blocks = (int)ceil_div(nq, 256), tiles = (int)ceil_div(np, 1024);
    // slices: as many as keep the launch at a few thousand workgroups (one tile of 1024 points costs a thread ~4 us)
    const int tiles_per_slice = (int)std::max<int64_t>(1, ceil_div((int64_t)q_blocks * tiles, 4096));
    const int slices = (int)ceil_div(tiles, tiles_per_slice);
    if (hipMemsetAsync(out_idx, 0xFF, (size_t)nq * sizeof(int64_t), as_stream(stream)) != hipSuccess) return FSGS_ELAUNCH;
    auto *best = reinterpret_cast<unsigned long long *>(out_idx);
    const float4 everywhere = make_float4(0.f, 0.f, 0.f, -1.f);
    if (want_idx)
        hipLaunchKernelGGL(nearest_point_kernel<true>, dim3(q_blocks, slices), dim3(256), 0, as_stream(stream), nq, queries,
                           np, points, tiles_per_slice, active, everywhere, best);
    else
        hipLaunchKernelGGL(nearest_point_kernel<false>, dim3(q_blocks, slices), dim3(256), 0, as_stream(stream), nq, queries,
                           np, points, tiles_per_slice, active, everywhere, best);
    hipLaunchKernelGGL(nearest_point_unpack_kernel, dim3(q_blocks), dim3(256), 0, as_stream(stream), nq, out_idx, out_dist);
    return check_launch();
}

extern "C" int fsgs_nearest_point_words(int nq, const float *queries, int np, const float *points, const float *center,
                                        float close_radius, uint64_t *out_words, fsgs_stream_t stream) {
    if (nq < 0 || np < 1) return FSGS_EINVAL;
    if (nq == 0) return FSGS_OK;
    if (!queries || !points || !out_words || (close_radius >= 0.f && !center)) return FSGS_EINVAL;
    const int q_blocks = (int)ceil_div(nq, 256), tiles = (int)ceil_div(np, 1024);
    const int tiles_per_slice = (int)std::max<int64_t>(1, ceil_div((int64_t)q_blocks * tiles, 4096));
    const int slices = (int)ceil_div(tiles, tiles_per_slice);
    if (hipMemsetAsync(out_words, 0xFF, (size_t)nq * sizeof(uint64_t), as_stream(stream)) != hipSuccess) return FSGS_ELAUNCH;
    const float4 close = close_radius >= 0.f ? make_float4(center[0], center[1], center[2], close_radius)
                                              : make_float4(0.f, 0.f, 0.f, -1.f);
    hipLaunchKernelGGL(nearest_point_kernel<false>, dim3(q_blocks, slices), dim3(256), 0, as_stream(stream), nq, queries, np,
                       points, tiles_per_slice, nullptr, close, reinterpret_cast<unsigned long long *>(out_words));
    return check_launch();
}
