// se3_device.h -- f64 SE(3) helpers for the bundle-adjustment kernels: unit quaternion + translation poses stored as
// 7 doubles (qx qy qz qw tx ty tz), restating g2o::SE3Quat (reference Thirdparty/g2o/g2o/types/se3quat.h) and the Eigen
// quaternion formulas it relies on.
#pragma once

#include <hip/hip_runtime.h>

namespace se3 {

__device__ __forceinline__ void quat_rotate(const double* q, const double* v, double* out)
{
    double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    ux += ux; uy += uy; uz += uz;
    out[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    out[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    out[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

__device__ __forceinline__ void pose_map(const double* T, const double* X, double* out)
{
    quat_rotate(T, X, out);
    out[0] += T[4]; out[1] += T[5]; out[2] += T[6];
}

__device__ __forceinline__ void quat_to_R(const double* q, double* R)
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// RECIP = one reciprocal and multiplications instead of a division per component (<= 1 ulp apart): for code where the
// call sits on a serial critical path (pose_solver.hip); the default divides exactly like Eigen does.
template <bool RECIP = false>
__device__ __forceinline__ void quat_normalize(double* q)       // SE3Quat::normalizeRotation
{
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (RECIP) { const double in = 1.0 / n; q[0] *= in; q[1] *= in; q[2] *= in; q[3] *= in; }
    else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}

template <int I>
__device__ inline void quat_from_R_case(const double* R, double* q)   // compile-time indices: no private-memory array
{
    constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
    double t = sqrt(R[I * 3 + I] - R[J * 3 + J] - R[K * 3 + K] + 1.0);
    q[I] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[K * 3 + J] - R[J * 3 + K]) * t;
    q[J] = (R[J * 3 + I] + R[I * 3 + J]) * t;
    q[K] = (R[K * 3 + I] + R[I * 3 + K]) * t;
}

__device__ inline void quat_from_R(const double* R, double* q)   // Eigen Quaternion(Matrix3d)
{
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else if (R[8] > (R[4] > R[0] ? R[4] : R[0])) {
        quat_from_R_case<2>(R, q);
    } else if (R[4] > R[0]) {
        quat_from_R_case<1>(R, q);
    } else {
        quat_from_R_case<0>(R, q);
    }
}

// VertexSE3Expmap::oplusImpl: est <- SE3Quat::exp(update) * est, update = (omega, upsilon)
template <bool RECIP = false>
__device__ inline void pose_oplus(const double* T, const double* u, double* out)
{
    const double om[3] = {u[0], u[1], u[2]};
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }   // se3quat.h:237-243 quirk
    } else {
        double sn, cs;
        sincos(theta, &sn, &cs);                // one argument reduction for both
        double a, b, c;
        if (RECIP) {
            const double it = 1.0 / theta, it2 = it * it;
            a = sn * it; b = (1 - cs) * it2; c = (theta - sn) * (it2 * it);
        } else {
            a = sn / theta;
            b = (1 - cs) / (theta * theta);
            c = (theta - sn) / (theta * theta * theta);
        }
        for (int i = 0; i < 9; i++) {
            const double I = (i % 4 == 0) ? 1.0 : 0.0;
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    double dq[4], dt[3];
    quat_from_R(R, dq);
    quat_normalize<RECIP>(dq);
    for (int i = 0; i < 3; i++) dt[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
    // result = exp * T
    double rt[3];
    quat_rotate(dq, T + 4, rt);
    out[4] = dt[0] + rt[0]; out[5] = dt[1] + rt[1]; out[6] = dt[2] + rt[2];
    out[3] = dq[3] * T[3] - dq[0] * T[0] - dq[1] * T[1] - dq[2] * T[2];
    out[0] = dq[3] * T[0] + dq[0] * T[3] + dq[1] * T[2] - dq[2] * T[1];
    out[1] = dq[3] * T[1] + dq[1] * T[3] + dq[2] * T[0] - dq[0] * T[2];
    out[2] = dq[3] * T[2] + dq[2] * T[3] + dq[0] * T[1] - dq[1] * T[0];
    quat_normalize<RECIP>(out);
}

}  // namespace se3
