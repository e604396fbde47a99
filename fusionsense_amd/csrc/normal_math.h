// Gaussian normals (shortest-axis column of R(q), flipped towards the camera) and their VJP, shared by
// densify.hip (stand-alone kernels) and project.hip (the fused per-Gaussian backward).  Mirrors
// /root/reference/dn_splatter/dn_model.py:617-656 (normals from quats/scales for the normal pass).
#pragma once
#include "common.h"

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
    // (no FMA contraction: the kernels that inline this — packing forward in its storage variants, the per-Gaussian
    // backward — must produce the same normal to the bit)
#pragma clang fp contract(off)
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

// world -> camera rotation of a normal (rows of c2w^T), same fixed rounding sequence
__device__ __forceinline__ void normal_to_camera(const float *n, const float *c2w, float *ex) {
#pragma clang fp contract(off)
#pragma unroll
    for (int j = 0; j < 3; ++j) ex[j] = n[0] * c2w[0 * 4 + j] + n[1] * c2w[1 * 4 + j] + n[2] * c2w[2 * 4 + j];
}

// v_quats (raw, un-normalised quaternion) from the gradient of the CAMERA-space normal
__device__ __forceinline__ float4 normal_backward(const float4 q, const float *ls, const float *mean,
                                                  const float *c2w, const float *v_ncam) {
    NormalCtx o;
    normal_forward(q, ls, mean, c2w, o);
    // v_n(world) = R_c2w v_ncam ; undo flip
    float vn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        vn[i] = o.sign * (c2w[i * 4 + 0] * v_ncam[0] + c2w[i * 4 + 1] * v_ncam[1] +
                          c2w[i * 4 + 2] * v_ncam[2]);
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
    return make_float4((gw - d * w) * o.inv_qn, (gx - d * x) * o.inv_qn, (gy - d * y) * o.inv_qn,
                       (gz - d * z) * o.inv_qn);
}


}  // namespace fsgs
