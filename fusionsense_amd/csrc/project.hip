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
struct GaussBwdFused {
    float4 *v_packed;            // [N,4] float4: [0..3] v_colors [4..6] v_conics [7..8] v_means2d [9..10] abs [11] v_opac [12..14] v_ncam
    const void *log_scales;      // raw scales (argmin axis of the normal)
    const float *opac;           // sigmoid(opacities)
    const float *c2w;            // [3,4] camera-to-world of the normal pass
    float *absgrad;              // [N,2] out
    float *v_opac_logit;         // [N] out
    int accumulate_means;        // v_means += (the SH backward wrote its share first) instead of =
    const uint8_t *frozen;       // nullable [N]: touch anchors — no gradient for means / scales / opacity (dn_model.py:535-541)
    int attr_half;               // quats / log_scales are IEEE-half mirrors (BASELINE config #5)
    int64_t replica_rows;        // > 0: large Gaussians own kGradReplicas gradient lines, this many rows apart (common.h)
    float *xys_grad_norm, *vis_counts, *max_2Dsize;  // after_train statistics (nullable together)
    float inv_max_hw;
};

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
    float4 pa, pb, pc, pd;  // FUSED: this Gaussian's packed gradient record
    if (FUSED) {
        pa = fz.v_packed[n * 4 + 0]; pb = fz.v_packed[n * 4 + 1]; pc = fz.v_packed[n * 4 + 2]; pd = fz.v_packed[n * 4 + 3];
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        fz.v_packed[n * 4 + 0] = zero4; fz.v_packed[n * 4 + 1] = zero4; fz.v_packed[n * 4 + 2] = zero4; fz.v_packed[n * 4 + 3] = zero4;
        if (fz.replica_rows > 0 && radii[n] > 0 &&
            grad_spread(conics[n * 3 + 0], conics[n * 3 + 1], conics[n * 3 + 2])) {
            // a large Gaussian: fold (and clear) the replicas the compositing backward spread its atomics over
            // (the colour float4 of each replica has been read by the SH backward before; this is the last reader)
            auto add4 = [](float4 &a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
#pragma unroll
            for (int r = 1; r < kGradReplicas; ++r) {
                float4 *line = fz.v_packed + ((int64_t)r * fz.replica_rows + n) * 4;
                add4(pa, line[0]); add4(pb, line[1]); add4(pc, line[2]); add4(pd, line[3]);
                line[0] = zero4; line[1] = zero4; line[2] = zero4; line[3] = zero4;
            }
        }
        reinterpret_cast<float2 *>(fz.absgrad)[n] = make_float2(pc.y, pc.z);
    }
    const float mean[3] = {means[n * 3 + 0], means[n * 3 + 1], means[n * 3 + 2]};
    const float4 q = ld_attr4(quats, n, FUSED ? fz.attr_half : 0);
    const float qn = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
    const float inv = 1.f / qn;
    const float qw = q.x * inv, qx = q.y * inv, qy = q.z * inv, qz = q.w * inv;
    Mat3 Rq;
    quat_to_rot(qw, qx, qy, qz, Rq);
    const float s[3] = {scales[n * 3 + 0], scales[n * 3 + 1], scales[n * 3 + 2]};

    float g_mean[3] = {0.f, 0.f, 0.f};
    float g_Sigma[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // world-space covariance

    for (int c = 0; c < C; ++c) {
        const int64_t idx = (int64_t)c * N + n;
        if (radii[idx] <= 0) continue;
        const float *vm = viewmats + c * 16;
        float W[3][3] = {{vm[0], vm[1], vm[2]}, {vm[4], vm[5], vm[6]}, {vm[8], vm[9], vm[10]}};
        const float x = W[0][0] * mean[0] + W[0][1] * mean[1] + W[0][2] * mean[2] + vm[3];
        const float y = W[1][0] * mean[0] + W[1][1] * mean[1] + W[1][2] * mean[2] + vm[7];
        const float z = W[2][0] * mean[0] + W[2][1] * mean[1] + W[2][2] * mean[2] + vm[11];
        float Sc[3][3];
        covar_cam(Rq, s, W, Sc);
        const float *K = Ks + c * 9;
        const float fx = K[0], fy = K[4];
        const float lim_x = 1.3f * (0.5f * width / fx), lim_y = 1.3f * (0.5f * height / fy);
        const float rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz;
        const float xr = x * rz, yr = y * rz;
        const bool free_x = (xr <= lim_x) && (xr >= -lim_x);
        const bool free_y = (yr <= lim_y) && (yr >= -lim_y);
        const float tx = z * fminf(lim_x, fmaxf(-lim_x, xr));
        const float ty = z * fminf(lim_y, fmaxf(-lim_y, yr));
        const float J00 = fx * rz, J02 = -fx * tx * rz2, J11 = fy * rz, J12 = -fy * ty * rz2;

        // conic = inverse(Sigma2 + eps I):  G = -X V X, X = conic matrix, V = sym(v_conic)
        const float ca = conics[idx * 3 + 0], cb = conics[idx * 3 + 1], cc = conics[idx * 3 + 2];
        const float va = FUSED ? pb.x : v_conics[idx * 3 + 0];
        const float vb = 0.5f * (FUSED ? pb.y : v_conics[idx * 3 + 1]);
        const float vc = FUSED ? pb.z : v_conics[idx * 3 + 2];
        // T = X V
        const float t00 = ca * va + cb * vb, t01 = ca * vb + cb * vc;
        const float t10 = cb * va + cc * vb, t11 = cb * vb + cc * vc;
        float G00 = -(t00 * ca + t01 * cb);
        float G01 = -(t00 * cb + t01 * cc);
        float G11 = -(t10 * cb + t11 * cc);
        if (v_compensations != nullptr && compensations != nullptr) {
            // comp = sqrt(det0/det); Sigma2' = Sigma2 + eps I has inverse X, det = 1/det(X)
            const float comp = compensations[idx];
            const float vcomp = v_compensations[idx];
            if (comp > 0.f && vcomp != 0.f) {
                const float detX = ca * cc - cb * cb;  // 1/det
                const float det = 1.f / detX;
                // Sigma2' entries from X: a = cc*det, b0 = -cb*det, c = ca*det
                const float a = cc * det, b0 = -cb * det, c2 = ca * det;
                const float a0 = a - eps2d, c0 = c2 - eps2d;
                const float det0 = a0 * c0 - b0 * b0;
                const float k = vcomp * 0.5f / comp;  // d comp / d (det0/det)
                // d(det0/det) = d det0 / det - det0/det^2 d det
                const float k0 = k / det, k1 = -k * det0 / (det * det);
                G00 += k0 * c0 + k1 * c2;
                G11 += k0 * a0 + k1 * a;
                G01 += -(k0 + k1) * b0;  // per off-diagonal entry (single-b gradient is -2b(k0+k1))
            }
        }
        // Sigma2 = J Sc J^T :  v_Sc = J^T G J ;  v_J = 2 G J Sc
        const float GJ0[3] = {G00 * J00, G01 * J11, G00 * J02 + G01 * J12};  // row 0 of G J
        const float GJ1[3] = {G01 * J00, G11 * J11, G01 * J02 + G11 * J12};  // row 1 of G J
        float vSc[3][3];
        const float Jc0[3] = {J00, 0.f, J02}, Jc1[3] = {0.f, J11, J12};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) vSc[i][j] = Jc0[i] * GJ0[j] + Jc1[i] * GJ1[j];
        float vJ0[3], vJ1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            vJ0[j] = 2.f * (GJ0[0] * Sc[0][j] + GJ0[1] * Sc[1][j] + GJ0[2] * Sc[2][j]);
            vJ1[j] = 2.f * (GJ1[0] * Sc[0][j] + GJ1[1] * Sc[1][j] + GJ1[2] * Sc[2][j]);
        }
        // camera-space mean gradient
        const float vu = FUSED ? pb.w : v_means2d[idx * 2 + 0], vv = FUSED ? pc.x : v_means2d[idx * 2 + 1];
        float gx = fx * rz * vu;
        float gy = fy * rz * vv;
        float gz = -(fx * x * vu + fy * y * vv) * rz2 + (FUSED ? pa.w : v_depths[idx]);
        gz += -fx * rz2 * vJ0[0] - fy * rz2 * vJ1[1];
        if (free_x) {
            gx += -fx * rz2 * vJ0[2];
            gz += 2.f * fx * tx * rz3 * vJ0[2];
        } else {
            gz += fx * tx * rz3 * vJ0[2];
        }
        if (free_y) {
            gy += -fy * rz2 * vJ1[2];
            gz += 2.f * fy * ty * rz3 * vJ1[2];
        } else {
            gz += fy * ty * rz3 * vJ1[2];
        }
        // back to world: v_mean += W^T g ; v_Sigma += W^T vSc W
#pragma unroll
        for (int j = 0; j < 3; ++j) g_mean[j] += W[0][j] * gx + W[1][j] * gy + W[2][j] * gz;
        float Tm[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                Tm[i][j] = vSc[i][0] * W[0][j] + vSc[i][1] * W[1][j] + vSc[i][2] * W[2][j];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                g_Sigma[i][j] += W[0][i] * Tm[0][j] + W[1][i] * Tm[1][j] + W[2][i] * Tm[2][j];

        if (v_viewmats != nullptr) {
            // p = W mean + t ; Sc = W Sigma W^T  ->  v_W = g mean^T + 2 vSc W Sigma ; v_t = g
            float Sig[3][3];
            {
                float A[3][3];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) A[i][j] = Rq.m[i][j] * s[j];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        Sig[i][j] = A[i][0] * A[j][0] + A[i][1] * A[j][1] + A[i][2] * A[j][2];
            }
            const float g[3] = {gx, gy, gz};
            float *vvm = v_viewmats + c * 16;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    // (vSc_sym W Sigma)_{ij}, vSc is symmetric here
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc += Tm[i][k] * Sig[k][j];
                    atomicAdd(&vvm[i * 4 + j], g[i] * mean[j] + 2.f * acc);
                }
                atomicAdd(&vvm[i * 4 + 3], g[i]);
            }
        }
    }

    // Sigma = M M^T, M = Rq S :  v_M = 2 v_Sigma M
    float vM[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            vM[i][j] = 2.f * (g_Sigma[i][0] * Rq.m[0][j] + g_Sigma[i][1] * Rq.m[1][j] +
                              g_Sigma[i][2] * Rq.m[2][j]) * s[j];
    float vs[3], vR[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        vs[j] = Rq.m[0][j] * vM[0][j] + Rq.m[1][j] * vM[1][j] + Rq.m[2][j] * vM[2][j];
#pragma unroll
        for (int i = 0; i < 3; ++i) vR[i][j] = vM[i][j] * s[j];
    }
    const float gw = 2.f * (qx * (vR[2][1] - vR[1][2]) + qy * (vR[0][2] - vR[2][0]) + qz * (vR[1][0] - vR[0][1]));
    const float gqx = 2.f * (-2.f * qx * (vR[1][1] + vR[2][2]) + qy * (vR[0][1] + vR[1][0]) +
                             qz * (vR[0][2] + vR[2][0]) + qw * (vR[2][1] - vR[1][2]));
    const float gqy = 2.f * (qx * (vR[0][1] + vR[1][0]) - 2.f * qy * (vR[0][0] + vR[2][2]) +
                             qz * (vR[1][2] + vR[2][1]) + qw * (vR[0][2] - vR[2][0]));
    const float gqz = 2.f * (qx * (vR[0][2] + vR[2][0]) + qy * (vR[1][2] + vR[2][1]) -
                             2.f * qz * (vR[0][0] + vR[1][1]) + qw * (vR[1][0] - vR[0][1]));
    // through q_hat = q/|q|
    const float dotp = gw * qw + gqx * qx + gqy * qy + gqz * qz;
    float4 vq = make_float4((gw - dotp * qw) * inv, (gqx - dotp * qx) * inv, (gqy - dotp * qy) * inv,
                            (gqz - dotp * qz) * inv);
    if (!FUSED) {
        v_means[n * 3 + 0] = g_mean[0];
        v_means[n * 3 + 1] = g_mean[1];
        v_means[n * 3 + 2] = g_mean[2];
        reinterpret_cast<float4 *>(v_quats)[n] = vq;
        v_scales[n * 3 + 0] = vs[0];
        v_scales[n * 3 + 1] = vs[1];
        v_scales[n * 3 + 2] = vs[2];
        return;
    }
    const bool frozen = fz.frozen != nullptr && fz.frozen[n] != 0;
    if (frozen) {
        v_means[n * 3 + 0] = 0.f; v_means[n * 3 + 1] = 0.f; v_means[n * 3 + 2] = 0.f;
    } else if (fz.accumulate_means) {
        v_means[n * 3 + 0] += g_mean[0]; v_means[n * 3 + 1] += g_mean[1]; v_means[n * 3 + 2] += g_mean[2];
    } else {
        v_means[n * 3 + 0] = g_mean[0]; v_means[n * 3 + 1] = g_mean[1]; v_means[n * 3 + 2] = g_mean[2];
    }
    // `scales` holds exp(log_scales): d/d log_scale = v * exp(log_scale)
    v_scales[n * 3 + 0] = frozen ? 0.f : vs[0] * s[0];
    v_scales[n * 3 + 1] = frozen ? 0.f : vs[1] * s[1];
    v_scales[n * 3 + 2] = frozen ? 0.f : vs[2] * s[2];
    const float o = fz.opac[n];
    fz.v_opac_logit[n] = frozen ? 0.f : pc.w * o * (1.f - o);
    {   // the normal pass reaches the quaternions only (dn_model.py:618-656)
        float ls[3];
        ld_attr3(fz.log_scales, n, fz.attr_half, ls);
        const float vn[3] = {pd.x, pd.y, pd.z};
        const float4 vqb = normal_backward(q, ls, mean, fz.c2w, vn);
        vq.x += vqb.x; vq.y += vqb.y; vq.z += vqb.z; vq.w += vqb.w;
    }
    reinterpret_cast<float4 *>(v_quats)[n] = vq;
    if (fz.xys_grad_norm) {  // SplatfactoModel.after_train (dn_model.py:1385-1389 registration)
        const int r = radii[n];
        if (r > 0) {
            fz.xys_grad_norm[n] += sqrtf(pc.y * pc.y + pc.z * pc.z);
            fz.vis_counts[n] += 1.f;
            fz.max_2Dsize[n] = fmaxf(fz.max_2Dsize[n], (float)r * fz.inv_max_hw);
        }
    }
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
