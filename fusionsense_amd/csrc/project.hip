// E2: fused 3D->2D Gaussian projection, forward and VJP.
// Replaces gsplat 1.0.0 `fully_fused_projection_{fwd,bwd}` as called from
// gsplat.rendering.rasterization at /root/reference/dn_splatter/dn_model.py:570-591.
//
// One thread per (camera, Gaussian).  Per-Gaussian attributes are SoA tensors of 3/4 floats,
// so consecutive lanes read consecutive 12/16-byte records: fully coalesced, no LDS needed.
// HBM-bound: 40 B read + 28 B written per Gaussian (forward).
#include "common.h"
#include "normal_math.h"
#include "project_math.h"
#include "gauss_bwd_body.h"

namespace fsgs {

// ACT: `scales` holds log-scales and the activations of dn_model.py:573-574 are applied here
// (scales_out = exp(scales), opac_out = sigmoid(opac_logit)): one launch less and the projection
// never re-reads the activated scales from memory.
template <bool ACT>
__global__ void __launch_bounds__(256)
project_fwd_kernel(int C, int N, const float *__restrict__ means, const void *__restrict__ quats,
                   const void *__restrict__ scales, const float *__restrict__ viewmats,
                   const float *__restrict__ Ks, int width, int height, float eps2d, float near_plane,
                   float far_plane, float radius_clip, int32_t *__restrict__ radii,
                   float *__restrict__ means2d, float *__restrict__ depths,
                   float *__restrict__ conics, float *__restrict__ compensations,
                   float *__restrict__ opac_logit, float *__restrict__ scales_out,
                   float *__restrict__ opac_out, int binarise, float binary_threshold,
                   __half *__restrict__ opac_logit_h, int attr_half) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)C * N) return;
    const int c = (int)(idx / N);
    const int n = (int)(idx - (int64_t)c * N);
    float s_act[3] = {0.f, 0.f, 0.f};
    float s_in[3];
    ld_attr3(scales, n, attr_half, s_in);
    if (ACT) {
        s_act[0] = expf(s_in[0]); s_act[1] = expf(s_in[1]); s_act[2] = expf(s_in[2]);
        if (c == 0) {
            scales_out[n * 3 + 0] = s_act[0]; scales_out[n * 3 + 1] = s_act[1]; scales_out[n * 3 + 2] = s_act[2];
            float ol = attr_half ? __half2float(opac_logit_h[n]) : opac_logit[n];
            if (binarise) {  // the binary-opacity write of get_outputs (dn_model.py:492-503), on the parameter itself
                ol = (ol >= binary_threshold) ? 1.f : 0.f;
                if (opac_logit) opac_logit[n] = ol;
                if (attr_half) opac_logit_h[n] = __float2half(ol);
            }
            opac_out[n] = 1.f / (1.f + expf(-ol));
        }
    }

    const float mx = means[n * 3 + 0], my = means[n * 3 + 1], mz = means[n * 3 + 2];
    const float4 q = ld_attr4(quats, n, attr_half);
    const float s[3] = {ACT ? s_act[0] : s_in[0], ACT ? s_act[1] : s_in[1], ACT ? s_act[2] : s_in[2]};
    const ProjOut po = project_one(mx, my, mz, q, s, viewmats + c * 16, Ks + c * 9, width, height, eps2d, near_plane,
                                   far_plane, radius_clip);
    const int32_t radius_out = po.radius;
    const float o_m2[2] = {po.u, po.v}, o_depth = po.depth, o_conic[3] = {po.ca, po.cb, po.cc}, o_comp = po.comp;
    radii[idx] = radius_out;
    reinterpret_cast<float2 *>(means2d)[idx] = make_float2(o_m2[0], o_m2[1]);
    depths[idx] = o_depth;
    conics[idx * 3 + 0] = o_conic[0];
    conics[idx * 3 + 1] = o_conic[1];
    conics[idx * 3 + 2] = o_conic[2];
    if (compensations) compensations[idx] = o_comp;
}

// VJP.  One thread per Gaussian, looping over cameras (C is 1 on the training path), so the
// per-Gaussian gradients are written once, without atomics, deterministically.
// FUSED (C == 1, the training path of fusionsense_amd/fused.py): the whole per-Gaussian tail of the
// backward in this one kernel — the image-space gradients come straight from the rasterizer's packed
// accumulator (which is cleared for the next frame), and the VJPs of exp(scales) / sigmoid(opacity),
// of the normal pass (quaternions), the absgrad output and SplatfactoModel.after_train's statistics
// are applied in place.  Replaces five launches (unpack, project_bwd, normals_bwd, activate_bwd,
// densify_stats) that each moved the same 300 k records.
template <bool FUSED>
__global__ void __launch_bounds__(256)
project_bwd_kernel(int C, int N, const float *__restrict__ means, const void *__restrict__ quats,
                   const float *__restrict__ scales, const float *__restrict__ viewmats,
                   const float *__restrict__ Ks, int width, int height, float eps2d,
                   const int32_t *__restrict__ radii, const float *__restrict__ conics,
                   const float *__restrict__ compensations, const float *__restrict__ v_means2d,
                   const float *__restrict__ v_depths, const float *__restrict__ v_conics,
                   const float *__restrict__ v_compensations, float *__restrict__ v_means,
                   float *__restrict__ v_quats, float *__restrict__ v_scales,
                   float *__restrict__ v_viewmats, GaussBwdFused fz) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa, pc = pa, pd = pa;
    if (FUSED) gauss_bwd_load_line(n, fz, radii, conics, pa, pb, pc, pd);
    gaussian_bwd_one<FUSED>(n, C, N, means, quats, scales, viewmats, Ks, width, height, eps2d, radii, conics,
                            compensations, v_means2d, v_depths, v_conics, v_compensations, v_means, v_quats, v_scales,
                            v_viewmats, fz, pa, pb, pc, pd, nullptr);
}

}  // namespace fsgs

using namespace fsgs;

extern "C" int fsgs_project_fwd(int C, int N, const float *means, const float *quats,
                                const float *scales, const float *viewmats, const float *Ks,
                                int width, int height, float eps2d, float near_plane,
                                float far_plane, float radius_clip, int32_t *radii, float *means2d,
                                float *depths, float *conics, float *compensations,
                                fsgs_stream_t stream) {
    if (C < 0 || N < 0) return FSGS_EINVAL;
    if ((int64_t)C * N == 0) return FSGS_OK;
    if (!means || !quats || !scales || !viewmats || !Ks || !radii || !means2d || !depths || !conics)
        return FSGS_EINVAL;
    const int64_t total = (int64_t)C * N;
    hipLaunchKernelGGL((project_fwd_kernel<false>), dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream),
                       C, N, means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane,
                       far_plane, radius_clip, radii, means2d, depths, conics, compensations, nullptr, nullptr,
                       nullptr, 0, 0.f, nullptr, 0);
    return check_launch();
}

extern "C" int fsgs_project_fwd_act(int C, int N, const float *means, const float *quats,
                                    const float *log_scales, float *opac_logit, int binarise,
                                    float binary_threshold, const float *viewmats,
                                    const float *Ks, int width, int height, float eps2d, float near_plane,
                                    float far_plane, float radius_clip, float *scales_out, float *opac_out,
                                    int32_t *radii, float *means2d, float *depths, float *conics,
                                    fsgs_stream_t stream) {
    if (C < 0 || N < 0) return FSGS_EINVAL;
    if ((int64_t)C * N == 0) return FSGS_OK;
    if (!means || !quats || !log_scales || !opac_logit || !viewmats || !Ks || !scales_out || !opac_out || !radii ||
        !means2d || !depths || !conics)
        return FSGS_EINVAL;
    const int64_t total = (int64_t)C * N;
    hipLaunchKernelGGL((project_fwd_kernel<true>), dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream),
                       C, N, means, quats, log_scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                       radius_clip, radii, means2d, depths, conics, nullptr, opac_logit, scales_out, opac_out,
                       binarise, binary_threshold, nullptr, 0);
    return check_launch();
}

// Half attribute storage (BASELINE config #5): quats / log_scales / opac_logit_h are IEEE-half mirrors of the fp32
// masters; a binary-opacity write goes to the mirror and to the master (nullable).
extern "C" int fsgs_project_fwd_act_h16(int C, int N, const float *means, const void *quats_h,
                                        const void *log_scales_h, void *opac_logit_h, float *opac_logit_master,
                                        int binarise, float binary_threshold, const float *viewmats, const float *Ks,
                                        int width, int height, float eps2d, float near_plane, float far_plane,
                                        float radius_clip, float *scales_out, float *opac_out, int32_t *radii,
                                        float *means2d, float *depths, float *conics, fsgs_stream_t stream) {
    if (C < 0 || N < 0) return FSGS_EINVAL;
    if ((int64_t)C * N == 0) return FSGS_OK;
    if (!means || !quats_h || !log_scales_h || !opac_logit_h || !viewmats || !Ks || !scales_out || !opac_out || !radii ||
        !means2d || !depths || !conics)
        return FSGS_EINVAL;
    const int64_t total = (int64_t)C * N;
    hipLaunchKernelGGL((project_fwd_kernel<true>), dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream),
                       C, N, means, quats_h, log_scales_h, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
                       radius_clip, radii, means2d, depths, conics, nullptr, opac_logit_master, scales_out, opac_out,
                       binarise, binary_threshold, reinterpret_cast<__half *>(opac_logit_h), 1);
    return check_launch();
}

extern "C" int fsgs_project_bwd(int C, int N, const float *means, const float *quats,
                                const float *scales, const float *viewmats, const float *Ks,
                                int width, int height, float eps2d, const int32_t *radii,
                                const float *conics, const float *compensations,
                                const float *v_means2d, const float *v_depths, const float *v_conics,
                                const float *v_compensations, float *v_means, float *v_quats,
                                float *v_scales, float *v_viewmats, fsgs_stream_t stream) {
    if (C < 0 || N < 0) return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!means || !quats || !scales || !viewmats || !Ks || !radii || !conics || !v_means2d ||
        !v_depths || !v_conics || !v_means || !v_quats || !v_scales)
        return FSGS_EINVAL;
    hipLaunchKernelGGL((project_bwd_kernel<false>), dim3(ceil_div(N, 256)), dim3(256), 0, as_stream(stream), C,
                       N, means, quats, scales, viewmats, Ks, width, height, eps2d, radii, conics,
                       compensations, v_means2d, v_depths, v_conics, v_compensations, v_means,
                       v_quats, v_scales, v_viewmats, GaussBwdFused{});
    return check_launch();
}

static int gaussian_bwd_impl(int N, const float *means, const void *quats, const void *log_scales, int attr_half,
                             int64_t replica_rows,
                                 const float *scales, const float *opac, const float *viewmat, const float *K,
                                 const float *c2w, int width, int height, float eps2d, const int32_t *radii,
                                 const float *conics, float *v_packed, int accumulate_means, float *v_means,
                                 float *v_quats, float *v_log_scales, float *v_opac_logit, float *absgrad,
                                 float *xys_grad_norm, float *vis_counts, float *max_2Dsize, float inv_max_hw,
                                 const uint8_t *frozen, fsgs_stream_t stream) {
    if (N < 0) return FSGS_EINVAL;
    if (N == 0) return FSGS_OK;
    if (!means || !quats || !log_scales || !scales || !opac || !viewmat || !K || !c2w || !radii || !conics ||
        !v_packed || !v_means || !v_quats || !v_log_scales || !v_opac_logit || !absgrad)
        return FSGS_EINVAL;
    if ((xys_grad_norm != nullptr) != (vis_counts != nullptr) || (xys_grad_norm != nullptr) != (max_2Dsize != nullptr))
        return FSGS_EINVAL;
    GaussBwdFused fz;
    fz.v_packed = reinterpret_cast<float4 *>(v_packed);
    fz.log_scales = log_scales; fz.opac = opac; fz.c2w = c2w; fz.absgrad = absgrad; fz.v_opac_logit = v_opac_logit;
    fz.accumulate_means = accumulate_means;
    fz.frozen = frozen;
    fz.attr_half = attr_half;
    if (replica_rows < 0) return FSGS_EINVAL;
    fz.replica_rows = replica_rows;
    fz.xys_grad_norm = xys_grad_norm; fz.vis_counts = vis_counts; fz.max_2Dsize = max_2Dsize; fz.inv_max_hw = inv_max_hw;
    hipLaunchKernelGGL((project_bwd_kernel<true>), dim3(ceil_div(N, 256)), dim3(256), 0, as_stream(stream), 1, N,
                       means, quats, scales, viewmat, K, width, height, eps2d, radii, conics, nullptr, nullptr,
                       nullptr, nullptr, nullptr, v_means, v_quats, v_log_scales, nullptr, fz);
    return check_launch();
}

extern "C" int fsgs_gaussian_bwd(int N, const float *means, const float *quats, const float *log_scales,
                                 const float *scales, const float *opac, const float *viewmat, const float *K,
                                 const float *c2w, int width, int height, float eps2d, const int32_t *radii,
                                 const float *conics, float *v_packed, int accumulate_means, float *v_means,
                                 float *v_quats, float *v_log_scales, float *v_opac_logit, float *absgrad,
                                 float *xys_grad_norm, float *vis_counts, float *max_2Dsize, float inv_max_hw,
                                 const uint8_t *frozen, int64_t replica_rows, fsgs_stream_t stream) {
    return gaussian_bwd_impl(N, means, quats, log_scales, 0, replica_rows, scales, opac, viewmat, K, c2w, width, height, eps2d, radii,
                             conics, v_packed, accumulate_means, v_means, v_quats, v_log_scales, v_opac_logit, absgrad,
                             xys_grad_norm, vis_counts, max_2Dsize, inv_max_hw, frozen, stream);
}

// quats_h / log_scales_h: IEEE-half mirrors (BASELINE config #5); gradients are fp32 as ever
extern "C" int fsgs_gaussian_bwd_h16(int N, const float *means, const void *quats_h, const void *log_scales_h,
                                     const float *scales, const float *opac, const float *viewmat, const float *K,
                                     const float *c2w, int width, int height, float eps2d, const int32_t *radii,
                                     const float *conics, float *v_packed, int accumulate_means, float *v_means,
                                     float *v_quats, float *v_log_scales, float *v_opac_logit, float *absgrad,
                                     float *xys_grad_norm, float *vis_counts, float *max_2Dsize, float inv_max_hw,
                                     const uint8_t *frozen, int64_t replica_rows, fsgs_stream_t stream) {
    return gaussian_bwd_impl(N, means, quats_h, log_scales_h, 1, replica_rows, scales, opac, viewmat, K, c2w, width, height, eps2d,
                             radii, conics, v_packed, accumulate_means, v_means, v_quats, v_log_scales, v_opac_logit,
                             absgrad, xys_grad_norm, vis_counts, max_2Dsize, inv_max_hw, frozen, stream);
}
