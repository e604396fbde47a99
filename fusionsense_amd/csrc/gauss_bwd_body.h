// The per-Gaussian VJP of the projection (+ the fused tail of the training backward) as a device function, shared by
// project_bwd_kernel (project.hip) and the kernel that runs the SH backward in the same thread (sh.hip).
#pragma once
#include <hip/hip_fp16.h>
#include "common.h"
#include "normal_math.h"
#include "project_math.h"

namespace fsgs {

// Adam inside the per-Gaussian backward (fsgs_gauss_sh_bwd_adam): the gradients are applied where they are formed and
// never written.  Groups in the trainer's order: 0 means, 1 log-scales, 2 quats, 3 features_dc, 4 features_rest,
// 5 opacity logits; h = the IEEE-half mirror the render kernels read (nullable per group), rewritten with the update.
struct GaussAdam {
    int on;
    float *p[6], *m[6], *v[6];
    __half *h[6];
    float ss[6];
    float b1, b2, omb1, omb2, isb2, eps;
    // the two_d_gaussians term of get_loss_dict (dn_model.py:817-819) reaches the log-scales directly: its gradient,
    // min_scale_g * exp(min_k log_scale_k) on the arg-min axis (what fsgs_min_scale_loss adds to the gradient slab after
    // the backward), is added here when the slab does not exist; 0 = no such term
    float min_scale_g;
    // 1: an element with zero gradient on zero moments is left alone (8 instead of 24 B: its moments are read first,
    // the parameter only if something changes — a dependent load, for scenes most of whose Gaussians no view has
    // reached); 0: parameter and moments are read together (one round trip) and every element is stepped.  Same
    // parameters either way: the skipped update is exactly "nothing changes".
    int skip_idle;
};

// The eleven geometry elements of Gaussian n (means 3, log-scales 3, quaternion 4, opacity logit 1) take the gradients
// gr[0..10] — same arithmetic as adam_kernel (common.h adam_one).  All loads of the thread are issued BEFORE its first
// store (the compiler cannot move a load of element k + 1 above the stores of element k: the pointers may alias as far
// as it knows, and eleven dependent load -> store round trips per thread were what bounded this launch in round 3):
// moments of all elements, then — with skip_idle — the parameters of the elements that change, else everything at once.
__device__ __forceinline__ void gauss_adam_geometry(const GaussAdam &a, int64_t n, const float (&gr)[11]) {
    constexpr int G[11] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 5};
    constexpr int W[11] = {3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 1};
    constexpr int K[11] = {0, 1, 2, 0, 1, 2, 0, 1, 2, 3, 0};
    float m[11], v[11], p[11];
    bool act[11];
#pragma unroll
    for (int e = 0; e < 11; ++e) {
        const int64_t i = n * W[e] + K[e];
        m[e] = a.m[G[e]][i]; v[e] = a.v[G[e]][i];
    }
    if (a.skip_idle) {
        // (zero gradient on zero moments — a Gaussian no view has reached yet: the update is exactly "nothing changes",
        // so the parameter is not even read and nothing is written: 8 instead of 24 B for this float)
#pragma unroll
        for (int e = 0; e < 11; ++e) {
            act[e] = !(gr[e] == 0.f && m[e] == 0.f && v[e] == 0.f);
            p[e] = act[e] ? a.p[G[e]][n * W[e] + K[e]] : 0.f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 11; ++e) {
            act[e] = true;
            p[e] = a.p[G[e]][n * W[e] + K[e]];
        }
    }
#pragma unroll
    for (int e = 0; e < 11; ++e)
        if (act[e]) adam_one(p[e], gr[e], m[e], v[e], a.b1, a.b2, a.omb1, a.omb2, a.ss[G[e]], a.isb2, a.eps);
#pragma unroll
    for (int e = 0; e < 11; ++e) {
        if (!act[e]) continue;
        const int64_t i = n * W[e] + K[e];
        a.p[G[e]][i] = p[e]; a.m[G[e]][i] = m[e]; a.v[G[e]][i] = v[e];
        if (a.h[G[e]]) a.h[G[e]][i] = __float2half(p[e]);
    }
}

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
    GaussAdam adam;              // .on: apply instead of writing v_means / v_quats / v_scales / v_opac_logit
};

// FUSED: this Gaussian's packed gradient record — read, replicas folded in (common.h: grad_spread), everything cleared
// for the next frame; writes absgrad.
__device__ __forceinline__ void gauss_bwd_load_line(int n, const GaussBwdFused &fz, const int32_t *__restrict__ radii,
                                                    const float *__restrict__ conics, float4 &pa, float4 &pb,
                                                    float4 &pc, float4 &pd) {
    pa = fz.v_packed[n * 4 + 0]; pb = fz.v_packed[n * 4 + 1]; pc = fz.v_packed[n * 4 + 2]; pd = fz.v_packed[n * 4 + 3];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // (a line nothing was added to is already clear: in dense scenes that is most of them, 64 B of stores less each)
    auto any4 = [](const float4 v) { return (__float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z) | __float_as_uint(v.w)) != 0u; };
    if (any4(pa) || any4(pb) || any4(pc) || any4(pd)) {
        fz.v_packed[n * 4 + 0] = zero4; fz.v_packed[n * 4 + 1] = zero4; fz.v_packed[n * 4 + 2] = zero4; fz.v_packed[n * 4 + 3] = zero4;
    }
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

template <bool FUSED, bool ADAM = false>
__device__ __forceinline__ void gaussian_bwd_one(
    int n, int C, int N, const float *__restrict__ means, const void *__restrict__ quats,
    const float *__restrict__ scales, const float *__restrict__ viewmats, const float *__restrict__ Ks, int width,
    int height, float eps2d, const int32_t *__restrict__ radii, const float *__restrict__ conics,
    const float *__restrict__ compensations, const float *__restrict__ v_means2d, const float *__restrict__ v_depths,
    const float *__restrict__ v_conics, const float *__restrict__ v_compensations, float *__restrict__ v_means,
    float *__restrict__ v_quats, float *__restrict__ v_scales, float *__restrict__ v_viewmats, const GaussBwdFused &fz,
    const float4 pa, const float4 pb, const float4 pc, const float4 pd, const float *mean_share) {
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
    float gm[3];
    bool gm_known = true;
    if (frozen) {
        gm[0] = 0.f; gm[1] = 0.f; gm[2] = 0.f;
    } else if (mean_share) {  // (the SH backward ran in this thread: its view-direction share arrives in registers)
        gm[0] = mean_share[0] + g_mean[0]; gm[1] = mean_share[1] + g_mean[1]; gm[2] = mean_share[2] + g_mean[2];
    } else if (fz.accumulate_means) {
        gm_known = false;
        v_means[n * 3 + 0] += g_mean[0]; v_means[n * 3 + 1] += g_mean[1]; v_means[n * 3 + 2] += g_mean[2];
    } else {
        gm[0] = g_mean[0]; gm[1] = g_mean[1]; gm[2] = g_mean[2];
    }
    // `scales` holds exp(log_scales): d/d log_scale = v * exp(log_scale)
    float gs[3] = {frozen ? 0.f : vs[0] * s[0], frozen ? 0.f : vs[1] * s[1], frozen ? 0.f : vs[2] * s[2]};
    if (ADAM && fz.adam.min_scale_g != 0.f) {  // (fp32 masters, first minimum on ties: as min_scale_kernel)
        const float *lsm = fz.adam.p[1] + (int64_t)n * 3;
        int k = 0;
        float sm = lsm[0];
        if (lsm[1] < sm) { sm = lsm[1]; k = 1; }
        if (lsm[2] < sm) { sm = lsm[2]; k = 2; }
        const float e = expf(sm);  // (one fused multiply-add, as min_scale_kernel accumulates it)
#pragma unroll
        for (int j = 0; j < 3; ++j) gs[j] = (k == j) ? __builtin_fmaf(fz.adam.min_scale_g, e, gs[j]) : gs[j];
    }
    const float o = fz.opac[n];
    const float go = frozen ? 0.f : pc.w * o * (1.f - o);
    {   // the normal pass reaches the quaternions only (dn_model.py:618-656)
        float ls[3];
        ld_attr3(fz.log_scales, n, fz.attr_half, ls);
        const float vn[3] = {pd.x, pd.y, pd.z};
        const float4 vqb = normal_backward(q, ls, mean, fz.c2w, vn);
        vq.x += vqb.x; vq.y += vqb.y; vq.z += vqb.z; vq.w += vqb.w;
    }
    if (ADAM) {  // (every input of this Gaussian has been read above: its parameters may move now)
        const float gr[11] = {gm[0], gm[1], gm[2], gs[0], gs[1], gs[2], vq.x, vq.y, vq.z, vq.w, go};
        gauss_adam_geometry(fz.adam, (int64_t)n, gr);
    } else {
        if (gm_known) { v_means[n * 3 + 0] = gm[0]; v_means[n * 3 + 1] = gm[1]; v_means[n * 3 + 2] = gm[2]; }
        v_scales[n * 3 + 0] = gs[0]; v_scales[n * 3 + 1] = gs[1]; v_scales[n * 3 + 2] = gs[2];
        fz.v_opac_logit[n] = go;
        reinterpret_cast<float4 *>(v_quats)[n] = vq;
    }
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
