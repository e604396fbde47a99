// Projection math shared by project.hip and the direct-binning count pass of isect.hip (E2):
// one Gaussian, one camera.  Same statements in both callers; the compiler schedules / contracts them differently in
// the two kernels, so the conics agree to rounding (~1e-6 relative, both within 1e-6 of an f64 evaluation), not to the bit.
#pragma once
#include "common.h"

namespace fsgs {

struct Mat3 {
    float m[3][3];
};

__device__ __forceinline__ void quat_to_rot(float w, float x, float y, float z, Mat3 &R) {
    R.m[0][0] = 1.f - 2.f * (y * y + z * z);
    R.m[0][1] = 2.f * (x * y - w * z);
    R.m[0][2] = 2.f * (x * z + w * y);
    R.m[1][0] = 2.f * (x * y + w * z);
    R.m[1][1] = 1.f - 2.f * (x * x + z * z);
    R.m[1][2] = 2.f * (y * z - w * x);
    R.m[2][0] = 2.f * (x * z - w * y);
    R.m[2][1] = 2.f * (y * z + w * x);
    R.m[2][2] = 1.f - 2.f * (x * x + y * y);
}

// Sigma_c = W (Rq S)(Rq S)^T W^T, returned as a full symmetric 3x3.
__device__ __forceinline__ void covar_cam(const Mat3 &Rq, const float s[3], const float W[3][3],
                                           float Sc[3][3]) {
    float A[3][3];  // A = W * Rq * S
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            A[i][j] = (W[i][0] * Rq.m[0][j] + W[i][1] * Rq.m[1][j] + W[i][2] * Rq.m[2][j]) * s[j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Sc[i][j] = A[i][0] * A[j][0] + A[i][1] * A[j][1] + A[i][2] * A[j][2];
}

struct ProjOut {
    int32_t radius;
    float u, v, depth, ca, cb, cc, comp;
};

// vm: world-to-camera 4x4 (row major), K: 3x3 intrinsics; s: activated scales
__device__ __forceinline__ ProjOut project_one(float mx, float my, float mz, const float4 q, const float s[3],
                                               const float *__restrict__ vm, const float *__restrict__ K, int width,
                                               int height, float eps2d, float near_plane, float far_plane,
                                               float radius_clip) {
    float W[3][3] = {{vm[0], vm[1], vm[2]}, {vm[4], vm[5], vm[6]}, {vm[8], vm[9], vm[10]}};
    const float t[3] = {vm[3], vm[7], vm[11]};
    const float x = W[0][0] * mx + W[0][1] * my + W[0][2] * mz + t[0];
    const float y = W[1][0] * mx + W[1][1] * my + W[1][2] * mz + t[1];
    const float z = W[2][0] * mx + W[2][1] * my + W[2][2] * mz + t[2];
    ProjOut o = {0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (z >= near_plane && z <= far_plane) {
        const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        Mat3 Rq;
        quat_to_rot(q.x * inv, q.y * inv, q.z * inv, q.w * inv, Rq);
        float Sc[3][3];
        covar_cam(Rq, s, W, Sc);
        const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
        const float lim_x = 1.3f * (0.5f * width / fx), lim_y = 1.3f * (0.5f * height / fy);
        const float rz = 1.f / z, rz2 = rz * rz;
        const float tx = z * fminf(lim_x, fmaxf(-lim_x, x * rz));
        const float ty = z * fminf(lim_y, fmaxf(-lim_y, y * rz));
        const float J00 = fx * rz, J02 = -fx * tx * rz2, J11 = fy * rz, J12 = -fy * ty * rz2;
        // Sigma2 = J Sigma_c J^T with J = [[J00,0,J02],[0,J11,J12]]
        const float r0[3] = {J00 * Sc[0][0] + J02 * Sc[2][0], J00 * Sc[0][1] + J02 * Sc[2][1],
                             J00 * Sc[0][2] + J02 * Sc[2][2]};
        const float r1[3] = {J11 * Sc[1][0] + J12 * Sc[2][0], J11 * Sc[1][1] + J12 * Sc[2][1],
                             J11 * Sc[1][2] + J12 * Sc[2][2]};
        const float a0 = r0[0] * J00 + r0[2] * J02;
        const float b0 = 0.5f * ((r0[1] * J11 + r0[2] * J12) + (r1[0] * J00 + r1[2] * J02));
        const float c0 = r1[1] * J11 + r1[2] * J12;
        const float det0 = a0 * c0 - b0 * b0;
        const float a = a0 + eps2d, cc = c0 + eps2d;
        const float det = a * cc - b0 * b0;
        if (det > 0.f) {
            const float bb = 0.5f * (a + cc);
            const float v1 = bb + sqrtf(fmaxf(0.01f, bb * bb - det));
            const float radius = ceilf(3.f * sqrtf(v1));
            const float u = fx * x * rz + cx, v = fy * y * rz + cy;
            const bool inside = (u + radius > 0.f) && (u - radius < (float)width) &&
                                (v + radius > 0.f) && (v - radius < (float)height);
            if (radius > radius_clip && inside) {
                const float idet = 1.f / det;
                o.radius = (int32_t)radius;
                o.u = u; o.v = v; o.depth = z;
                o.ca = cc * idet; o.cb = -b0 * idet; o.cc = a * idet;
                o.comp = sqrtf(fmaxf(0.f, det0 * idet));
            }
        }
    }
    return o;
}

}  // namespace fsgs
