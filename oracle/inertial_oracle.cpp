/*
 * inertial_oracle.cpp -- CPU ORACLE (test infrastructure, not the product) for Optimizer::LocalInertialBA,
 * SURVEY.md 8(f) rank 4.  GROUNDWORK: there is no HIP counterpart yet; this restatement and its
 * tests exist so that the kernels of the next round have something to be checked against.  PARITY UNPINNED.
 *
 * Restates, from the reference text:
 *   Optimizer::LocalInertialBA                      src/Optimizer.cc:2383-2958 (numerical core :2503-2860; the graph walk
 *                                                   :2383-2500 and the map write-back :2862-2957 are the host shim's)
 *   ImuCamPose::Update / Project / isDepthPositive  src/G2oTypes.cc:203-258
 *   EdgeMono / EdgeStereo computeError, linearize   include/G2oTypes.h:342-470, src/G2oTypes.cc:349-420
 *   EdgeInertial computeError / linearizeOplus      src/G2oTypes.cc:520-590
 *   EdgeGyroRW / EdgeAccRW                          include/G2oTypes.h:635-700
 *   ExpSO3 / LogSO3 / (Inverse)RightJacobianSO3     src/G2oTypes.cc:777-853
 *   Preintegrated::GetDeltaRotation/Velocity/Position(bias)   src/ImuTypes.cc:276-307 -- FLOAT expressions on a FLOAT bias
 *   g2o Levenberg on BlockSolverX with marginalised landmarks: same control flow as lba_oracle.cpp (levenberg.cpp:61-194)
 *
 * Inputs are what the shim reads off the reference's objects: per key frame Rwb, twb, velocity, biases; per inertial link
 * the pre-integrated dR, dV, dP, their bias Jacobians, dT, the linearisation bias and the three information matrices (the
 * 9 x 9 one already symmetrised and eigenvalue-clamped, G2oTypes.cc:510-518, and scaled by 1e-2 for the link to the fixed
 * key frame, Optimizer.cc:2651); IMU::Preintegrated::IntegrateNewMeasurement itself stays on the host.
 *
 * Deviations, both far inside the 1e-4 tolerance of the BA path: NormalizeRotation (JacobiSVD, U V^T) is evaluated as the
 * polar factor by Newton iteration; Sophus::SO3f::exp is evaluated in double and rounded to float.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "oracle.h"

namespace {


void mat_mul(const double* A, const double* B, double* C)            /* C = A B (3x3, row major) */
{
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    std::memcpy(C, t, sizeof(t));
}
void mat_tr(const double* A, double* T) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * j + i]; }
void mat_vec(const double* A, const double* v, double* o)
{
    const double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2], t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2], t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
bool inv3(const double* A, double* Ai)
{
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    if (det == 0.0 || !std::isfinite(det)) return false;
    const double id = 1.0 / det;
    Ai[0] = c00 * id; Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id; Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Ai[3] = c01 * id; Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id; Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Ai[6] = c02 * id; Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id; Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return true;
}
/* NormalizeRotation: U V^T of the SVD = the orthogonal polar factor (G2oTypes.h:67-71) */
void normalize_rotation(double* R)
{
    for (int it = 0; it < 6; it++) {
        double Ri[9], Rit[9];
        if (!inv3(R, Ri)) return;
        mat_tr(Ri, Rit);
        double delta = 0;
        for (int k = 0; k < 9; k++) { const double n = 0.5 * (R[k] + Rit[k]); delta = std::max(delta, std::fabs(n - R[k])); R[k] = n; }
        if (delta < 1e-16) break;
    }
}
void skew(const double* w, double* W) { W[0] = 0; W[1] = -w[2]; W[2] = w[1]; W[3] = w[2]; W[4] = 0; W[5] = -w[0]; W[6] = -w[1]; W[7] = w[0]; W[8] = 0; }
void exp_so3(const double* w, double* R)                               /* G2oTypes.cc:782-798 */
{
    const double d2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], d = std::sqrt(d2);
    double W[9], W2[9];
    skew(w, W);
    mat_mul(W, W, W2);
    const double a = d < 1e-5 ? 1.0 : std::sin(d) / d, b = d < 1e-5 ? 0.5 : (1.0 - std::cos(d)) / d2;
    for (int k = 0; k < 9; k++) R[k] = ((k % 4 == 0) ? 1.0 : 0.0) + W[k] * a + W2[k] * b;
    normalize_rotation(R);
}
void log_so3(const double* R, double* w)                               /* :800-814 */
{
    const double tr = R[0] + R[4] + R[8];
    w[0] = (R[7] - R[5]) / 2; w[1] = (R[2] - R[6]) / 2; w[2] = (R[3] - R[1]) / 2;
    const double costheta = (tr - 1.0) * 0.5f;
    if (costheta > 1 || costheta < -1) return;
    const double theta = std::acos(costheta), s = std::sin(theta);
    if (std::fabs(s) < 1e-5) return;
    for (int k = 0; k < 3; k++) w[k] = theta * w[k] / s;
}
void inv_right_jac(const double* v, double* J)                         /* :821-832 */
{
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = std::sqrt(d2);
    double W[9], W2[9];
    skew(v, W);
    mat_mul(W, W, W2);
    for (int k = 0; k < 9; k++) J[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (d < 1e-5) return;
    const double c = 1.0 / d2 - (1.0 + std::cos(d)) / (2.0 * d * std::sin(d));
    for (int k = 0; k < 9; k++) J[k] += W[k] / 2 + W2[k] * c;
}
void right_jac(const double* v, double* J)                             /* :839-854 */
{
    const double d2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2], d = std::sqrt(d2);
    double W[9], W2[9];
    skew(v, W);
    mat_mul(W, W, W2);
    for (int k = 0; k < 9; k++) J[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (d < 1e-5) return;
    const double a = (1.0 - std::cos(d)) / d2, b = (d - std::sin(d)) / (d2 * d);
    for (int k = 0; k < 9; k++) J[k] += -W[k] * a + W2[k] * b;
}

struct KF {
    double Rwb[9], twb[3], Rcw[9], tcw[3];
    int its;
    double v[3], bg[3], ba[3];
};

struct Problem {
    const OracleInertialProblem* p;
    std::vector<KF> kf, saved;
    std::vector<double> pts, pts_saved;
    std::vector<int> off_pose, off_v, off_g, off_a;       /* offsets in the reduced vector, -1 = fixed / absent */
    int np = 0;                                           /* reduced (non-landmark) unknowns */
    double Rcb[9], tcb[3], Rbc[9], tbc[3];
    /* linearisation */
    std::vector<double> Hpp, bp, Hll, bl, W;              /* W: 6x3 per visual edge */
    std::vector<double> verr, vrho1;                      /* visual edge errors (3 each) and Huber weights */
    std::vector<double> x;

    void camera_from_body(KF& k) const                    /* ImuCamPose::Update tail (:249-257) */
    {
        double Rbw[9], tbw[3];
        mat_tr(k.Rwb, Rbw);
        mat_vec(Rbw, k.twb, tbw);
        for (int i = 0; i < 3; i++) tbw[i] = -tbw[i];
        mat_mul(Rcb, Rbw, k.Rcw);
        mat_vec(Rcb, tbw, k.tcw);
        for (int i = 0; i < 3; i++) k.tcw[i] += tcb[i];
    }
    void update_pose(KF& k, const double* pu) const       /* ImuCamPose::Update (:230-258) */
    {
        double t[3], dR[9];
        mat_vec(k.Rwb, pu + 3, t);
        for (int i = 0; i < 3; i++) k.twb[i] += t[i];
        exp_so3(pu, dR);
        mat_mul(k.Rwb, dR, k.Rwb);
        if (++k.its >= 3) { normalize_rotation(k.Rwb); k.its = 0; }
        camera_from_body(k);
    }
    void project(const KF& k, const double* Xw, double* Xc) const
    {
        mat_vec(k.Rcw, Xw, Xc);
        for (int i = 0; i < 3; i++) Xc[i] += k.tcw[i];
    }

    /* Preintegrated::GetDelta*(b1): float expressions on a float bias (ImuTypes.cc:276-307) */
    void delta(const OracleInertialLink& L, const KF& k1, double* dR, double* dV, double* dP, double* dbg_out) const
    {
        const float bgf[3] = {(float)k1.bg[0], (float)k1.bg[1], (float)k1.bg[2]}, baf[3] = {(float)k1.ba[0], (float)k1.ba[1], (float)k1.ba[2]};
        float dbg[3], dba[3];
        for (int i = 0; i < 3; i++) { dbg[i] = bgf[i] - L.bias0[3 + i]; dba[i] = baf[i] - L.bias0[i]; }     /* bias0 = bax bay baz bwx bwy bwz */
        float w[3];
        for (int i = 0; i < 3; i++) w[i] = L.JRg[3 * i] * dbg[0] + L.JRg[3 * i + 1] * dbg[1] + L.JRg[3 * i + 2] * dbg[2];
        /* Sophus::SO3f::exp(w).matrix(): quaternion exponential; evaluated in double, rounded to float */
        const double wd[3] = {w[0], w[1], w[2]};
        const double th2 = wd[0] * wd[0] + wd[1] * wd[1] + wd[2] * wd[2], th = std::sqrt(th2);
        const double imag = th < 1e-5 ? 0.5 - th2 / 48.0 : std::sin(0.5 * th) / th, real = th < 1e-5 ? 1.0 - th2 / 8.0 : std::cos(0.5 * th);
        const double qx = imag * wd[0], qy = imag * wd[1], qz = imag * wd[2], qw = real;
        float E[9];
        E[0] = (float)(1 - 2 * (qy * qy + qz * qz)); E[1] = (float)(2 * (qx * qy - qz * qw)); E[2] = (float)(2 * (qx * qz + qy * qw));
        E[3] = (float)(2 * (qx * qy + qz * qw)); E[4] = (float)(1 - 2 * (qx * qx + qz * qz)); E[5] = (float)(2 * (qy * qz - qx * qw));
        E[6] = (float)(2 * (qx * qz - qy * qw)); E[7] = (float)(2 * (qy * qz + qx * qw)); E[8] = (float)(1 - 2 * (qx * qx + qy * qy));
        float Rf[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rf[3 * i + j] = L.dR[3 * i] * E[j] + L.dR[3 * i + 1] * E[3 + j] + L.dR[3 * i + 2] * E[6 + j];
        for (int k = 0; k < 9; k++) dR[k] = Rf[k];
        normalize_rotation(dR);
        for (int k = 0; k < 9; k++) dR[k] = (double)(float)dR[k];                /* the reference's result is a Matrix3f */
        for (int i = 0; i < 3; i++) {
            const float dv = L.dV[i] + (L.JVg[3 * i] * dbg[0] + L.JVg[3 * i + 1] * dbg[1] + L.JVg[3 * i + 2] * dbg[2]) +
                             (L.JVa[3 * i] * dba[0] + L.JVa[3 * i + 1] * dba[1] + L.JVa[3 * i + 2] * dba[2]);
            const float dp = L.dP[i] + (L.JPg[3 * i] * dbg[0] + L.JPg[3 * i + 1] * dbg[1] + L.JPg[3 * i + 2] * dbg[2]) +
                             (L.JPa[3 * i] * dba[0] + L.JPa[3 * i + 1] * dba[1] + L.JPa[3 * i + 2] * dba[2]);
            dV[i] = dv; dP[i] = dp;
        }
        for (int i = 0; i < 3; i++) dbg_out[i] = dbg[i];
    }

    /* EdgeInertial::computeError (G2oTypes.cc:520-540) */
    void inertial_error(const OracleInertialLink& L, double* e9) const
    {
        const KF &k1 = kf[L.kf1], &k2 = kf[L.kf2];
        double dR[9], dV[3], dP[3], dbg[3];
        delta(L, k1, dR, dV, dP, dbg);
        const double dt = L.dT, g[3] = {0, 0, -9.81};
        double Rbw1[9], dRt[9], eR[9], t[9];
        mat_tr(k1.Rwb, Rbw1); mat_tr(dR, dRt);
        mat_mul(dRt, Rbw1, t); mat_mul(t, k2.Rwb, eR);
        log_so3(eR, e9);
        double a[3], b[3];
        for (int i = 0; i < 3; i++) a[i] = k2.v[i] - k1.v[i] - g[i] * dt;
        mat_vec(Rbw1, a, e9 + 3);
        for (int i = 0; i < 3; i++) e9[3 + i] -= dV[i];
        for (int i = 0; i < 3; i++) b[i] = k2.twb[i] - k1.twb[i] - k1.v[i] * dt - g[i] * dt * dt / 2;
        mat_vec(Rbw1, b, e9 + 6);
        for (int i = 0; i < 3; i++) e9[6 + i] -= dP[i];
    }

    /* EdgeInertial::linearizeOplus (:542-590): J[v] is 9 x dim(v), v = P1 V1 G1 A1 P2 V2 */
    void inertial_jacobians(const OracleInertialLink& L, double J[6][54]) const
    {
        const KF &k1 = kf[L.kf1], &k2 = kf[L.kf2];
        double dR[9], dV[3], dP[3], dbg[3];
        delta(L, k1, dR, dV, dP, dbg);
        const double dt = L.dT, g[3] = {0, 0, -9.81};
        double Rbw1[9], dRt[9], eR[9], t[9], er[3], invJr[9];
        mat_tr(k1.Rwb, Rbw1); mat_tr(dR, dRt);
        mat_mul(dRt, Rbw1, t); mat_mul(t, k2.Rwb, eR);
        log_so3(eR, er);
        inv_right_jac(er, invJr);
        for (int v = 0; v < 6; v++) std::memset(J[v], 0, sizeof(J[v]));
        auto set = [](double* Jv, int cols, int r0, int c0, const double* B, double s) {
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Jv[(r0 + i) * cols + c0 + j] = s * B[3 * i + j];
        };
        double Rbw2[9], A[9], B[9], h[9], a[3], b[3], ra[3], rb[3];
        mat_tr(k2.Rwb, Rbw2);
        mat_mul(Rbw2, k1.Rwb, A); mat_mul(invJr, A, B);
        set(J[0], 6, 0, 0, B, -1.0);                                               /* -invJr Rwb2^T Rwb1 */
        for (int i = 0; i < 3; i++) a[i] = k2.v[i] - k1.v[i] - g[i] * dt;
        mat_vec(Rbw1, a, ra); skew(ra, h); set(J[0], 6, 3, 0, h, 1.0);
        for (int i = 0; i < 3; i++) b[i] = k2.twb[i] - k1.twb[i] - k1.v[i] * dt - 0.5 * g[i] * dt * dt;
        mat_vec(Rbw1, b, rb); skew(rb, h); set(J[0], 6, 6, 0, h, 1.0);
        const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        set(J[0], 6, 6, 3, I3, -1.0);
        set(J[1], 3, 3, 0, Rbw1, -1.0); set(J[1], 3, 6, 0, Rbw1, -dt);
        double JRg[9], JVg[9], JVa[9], JPg[9], JPa[9], w[3], rj[9], eRt[9], C[9], D[9];
        for (int k = 0; k < 9; k++) { JRg[k] = L.JRg[k]; JVg[k] = L.JVg[k]; JVa[k] = L.JVa[k]; JPg[k] = L.JPg[k]; JPa[k] = L.JPa[k]; }
        mat_vec(JRg, dbg, w); right_jac(w, rj); mat_tr(eR, eRt);
        mat_mul(invJr, eRt, C); mat_mul(C, rj, D); mat_mul(D, JRg, C);
        set(J[2], 3, 0, 0, C, -1.0); set(J[2], 3, 3, 0, JVg, -1.0); set(J[2], 3, 6, 0, JPg, -1.0);
        set(J[3], 3, 3, 0, JVa, -1.0); set(J[3], 3, 6, 0, JPa, -1.0);
        set(J[4], 6, 0, 0, invJr, 1.0);
        mat_mul(Rbw1, k2.Rwb, A); set(J[4], 6, 6, 3, A, 1.0);
        set(J[5], 3, 3, 0, Rbw1, 1.0);
    }

    /* computeActiveErrors + activeRobustChi2 */
    double errors()
    {
        const OracleInertialProblem& P = *p;
        double chi = 0;
        for (int l = 0; l < P.n_links; l++) {
            const OracleInertialLink& L = P.links[l];
            double e[9], c = 0;
            inertial_error(L, e);
            for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) c += e[i] * L.info9[9 * i + j] * e[j];
            chi += L.robust ? huber(c, P.huber_inertial) : c;
            double eg = 0, ea = 0, dg[3], da[3];
            for (int i = 0; i < 3; i++) { dg[i] = kf[L.kf2].bg[i] - kf[L.kf1].bg[i]; da[i] = kf[L.kf2].ba[i] - kf[L.kf1].ba[i]; }
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { eg += dg[i] * L.info_gyro[3 * i + j] * dg[j]; ea += da[i] * L.info_acc[3 * i + j] * da[j]; }
            chi += eg; chi += ea;
        }
        for (int e = 0; e < P.n_edges; e++) {
            double Xc[3];
            project(kf[P.edge_kf[e]], &pts[3 * P.edge_point[e]], Xc);
            const double u = P.fx * Xc[0] / Xc[2] + P.cx, v = P.fy * Xc[1] / Xc[2] + P.cy;
            double* r = &verr[3 * e];
            r[0] = P.edge_obs[3 * e] - u; r[1] = P.edge_obs[3 * e + 1] - v; r[2] = 0;
            if (P.edge_stereo[e]) r[2] = P.edge_obs[3 * e + 2] - (u - P.bf * (1 / Xc[2]));        /* ImuCamPose::ProjectStereo */
            const double c = P.edge_inv_sigma2[e] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            const double delta = P.edge_stereo[e] ? P.huber_stereo : P.huber_mono;
            chi += huber(c, delta);
            vrho1[e] = c <= delta * delta ? 1.0 : delta / std::sqrt(c);
        }
        return chi;
    }
    static double huber(double c, double delta) { const double d2 = delta * delta; return c <= d2 ? c : 2 * std::sqrt(c) * delta - d2; }

    void add_block(int oi, int di, int oj, int dj, const double* Ji, const double* Jj, const double* Om, int dim_e)
    {
        if (oi < 0 || oj < 0) return;                              /* H(oi.., oj..) += Ji^T Om Jj */
        for (int a = 0; a < di; a++)
            for (int b = 0; b < dj; b++) {
                double s = 0;
                for (int r = 0; r < dim_e; r++) {
                    double t = 0;
                    for (int c = 0; c < dim_e; c++) t += Om[r * dim_e + c] * Jj[c * dj + b];
                    s += Ji[r * di + a] * t;
                }
                Hpp[(size_t)(oi + a) * np + oj + b] += s;
            }
    }

    /* buildSystem on the current estimate (errors() must have run) */
    void linearize()
    {
        const OracleInertialProblem& P = *p;
        std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(bp.begin(), bp.end(), 0.0);
        std::fill(Hll.begin(), Hll.end(), 0.0); std::fill(bl.begin(), bl.end(), 0.0); std::fill(W.begin(), W.end(), 0.0);
        for (int l = 0; l < P.n_links; l++) {
            const OracleInertialLink& L = P.links[l];
            double e[9], J[6][54], Om[81];
            inertial_error(L, e);
            inertial_jacobians(L, J);
            double c = 0;
            for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) c += e[i] * L.info9[9 * i + j] * e[j];
            const double d2 = P.huber_inertial * P.huber_inertial;
            const double rho1 = (L.robust && c > d2) ? P.huber_inertial / std::sqrt(c) : 1.0;
            for (int k = 0; k < 81; k++) Om[k] = rho1 * L.info9[k];
            const int off[6] = {off_pose[L.kf1], off_v[L.kf1], off_g[L.kf1], off_a[L.kf1], off_pose[L.kf2], off_v[L.kf2]};
            const int dim[6] = {6, 3, 3, 3, 6, 3};
            for (int a = 0; a < 6; a++) {
                if (off[a] < 0) continue;
                for (int b = 0; b < 6; b++) add_block(off[a], dim[a], off[b], dim[b], J[a], J[b], Om, 9);
                for (int i = 0; i < dim[a]; i++) {                   /* b -= J^T Om e */
                    double s = 0;
                    for (int r = 0; r < 9; r++) { double t = 0; for (int q = 0; q < 9; q++) t += Om[9 * r + q] * e[q]; s += J[a][r * dim[a] + i] * t; }
                    bp[off[a] + i] -= s;
                }
            }
            /* random walks: e = b2 - b1, J1 = -I, J2 = +I */
            for (int which = 0; which < 2; which++) {
                const double* Om3 = which ? L.info_acc : L.info_gyro;
                const int o1 = which ? off_a[L.kf1] : off_g[L.kf1], o2 = which ? off_a[L.kf2] : off_g[L.kf2];
                double d[3], Od[3];
                for (int i = 0; i < 3; i++) d[i] = which ? kf[L.kf2].ba[i] - kf[L.kf1].ba[i] : kf[L.kf2].bg[i] - kf[L.kf1].bg[i];
                mat_vec(Om3, d, Od);
                for (int i = 0; i < 3; i++) {
                    for (int j = 0; j < 3; j++) {
                        if (o1 >= 0) Hpp[(size_t)(o1 + i) * np + o1 + j] += Om3[3 * i + j];
                        if (o2 >= 0) Hpp[(size_t)(o2 + i) * np + o2 + j] += Om3[3 * i + j];
                        if (o1 >= 0 && o2 >= 0) { Hpp[(size_t)(o1 + i) * np + o2 + j] -= Om3[3 * i + j]; Hpp[(size_t)(o2 + i) * np + o1 + j] -= Om3[3 * i + j]; }
                    }
                    if (o1 >= 0) bp[o1 + i] += Od[i];
                    if (o2 >= 0) bp[o2 + i] -= Od[i];
                }
            }
        }
        for (int e = 0; e < P.n_edges; e++) {
            const KF& k = kf[P.edge_kf[e]];
            const int pt = P.edge_point[e], op = off_pose[P.edge_kf[e]], ne = P.edge_stereo[e] ? 3 : 2;
            double Xc[3], Xb[3];
            project(k, &pts[3 * pt], Xc);
            mat_vec(Rbc, Xc, Xb);
            for (int i = 0; i < 3; i++) Xb[i] += tbc[i];
            const double iz = 1.0 / Xc[2], iz2 = 1.0 / (Xc[2] * Xc[2]);
            double pj[9] = {P.fx * iz, 0, -P.fx * Xc[0] * iz2, 0, P.fy * iz, -P.fy * Xc[1] * iz2, 0, 0, 0};     /* Pinhole::projectJac */
            if (ne == 3) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + P.bf * iz2; }
            double Ji[9], Jj[18], pr[9];
            mat_mul(pj, k.Rcw, Ji);
            for (int q = 0; q < 9; q++) Ji[q] = -Ji[q];                          /* -proj_jac Rcw */
            mat_mul(pj, Rcb, pr);
            const double x = Xb[0], y = Xb[1], z = Xb[2];
            const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 6; c++) Jj[6 * r + c] = pr[3 * r] * D[c] + pr[3 * r + 1] * D[6 + c] + pr[3 * r + 2] * D[12 + c];
            const double w = P.edge_inv_sigma2[e] * vrho1[e];
            const double* r = &verr[3 * e];
            for (int a = 0; a < 3; a++) {
                for (int b = 0; b < 3; b++) { double s = 0; for (int q = 0; q < ne; q++) s += Ji[3 * q + a] * Ji[3 * q + b]; Hll[9 * pt + 3 * a + b] += w * s; }
                double s = 0; for (int q = 0; q < ne; q++) s += Ji[3 * q + a] * r[q];
                bl[3 * pt + a] -= w * s;
            }
            if (op >= 0) {
                for (int a = 0; a < 6; a++) {
                    for (int b = 0; b < 6; b++) { double s = 0; for (int q = 0; q < ne; q++) s += Jj[6 * q + a] * Jj[6 * q + b]; Hpp[(size_t)(op + a) * np + op + b] += w * s; }
                    double s = 0; for (int q = 0; q < ne; q++) s += Jj[6 * q + a] * r[q];
                    bp[op + a] -= w * s;
                    for (int b = 0; b < 3; b++) { double t = 0; for (int q = 0; q < ne; q++) t += Jj[6 * q + a] * Ji[3 * q + b]; W[18 * e + 3 * a + b] = w * t; }
                }
            }
        }
    }

    /* Schur complement with lambda on every diagonal entry, solve, trial update; returns false when not positive definite */
    bool solve_and_update(double lambda, double* scale)
    {
        const OracleInertialProblem& P = *p;
        std::vector<double> S(Hpp), bs(bp), Dinv(9 * (size_t)P.n_points);
        for (int i = 0; i < np; i++) S[(size_t)i * np + i] += lambda;
        std::vector<std::vector<int>> by_pt(P.n_points);
        for (int e = 0; e < P.n_edges; e++) by_pt[P.edge_point[e]].push_back(e);
        for (int l = 0; l < P.n_points; l++) {
            double D[9];
            for (int k = 0; k < 9; k++) D[k] = Hll[9 * l + k] + ((k % 4 == 0) ? lambda : 0.0);
            if (!inv3(D, &Dinv[9 * l])) return false;
            const double* Di = &Dinv[9 * l];
            for (int ea : by_pt[l]) {
                const int oa = off_pose[P.edge_kf[ea]];
                if (oa < 0) continue;
                double Z[18];                                                   /* W_a Dinv */
                for (int r = 0; r < 6; r++) for (int c = 0; c < 3; c++) Z[3 * r + c] = W[18 * ea + 3 * r] * Di[c] + W[18 * ea + 3 * r + 1] * Di[3 + c] + W[18 * ea + 3 * r + 2] * Di[6 + c];
                for (int r = 0; r < 6; r++) bs[oa + r] -= Z[3 * r] * bl[3 * l] + Z[3 * r + 1] * bl[3 * l + 1] + Z[3 * r + 2] * bl[3 * l + 2];
                for (int eb : by_pt[l]) {
                    const int ob = off_pose[P.edge_kf[eb]];
                    if (ob < 0) continue;
                    for (int r = 0; r < 6; r++)
                        for (int c = 0; c < 6; c++)
                            S[(size_t)(oa + r) * np + ob + c] -= Z[3 * r] * W[18 * eb + 3 * c] + Z[3 * r + 1] * W[18 * eb + 3 * c + 1] + Z[3 * r + 2] * W[18 * eb + 3 * c + 2];
                }
            }
        }
        /* Cholesky of the reduced system */
        std::vector<double> Lm(S), y(bs);
        for (int j = 0; j < np; j++) {
            double d = Lm[(size_t)j * np + j];
            for (int k = 0; k < j; k++) d -= Lm[(size_t)j * np + k] * Lm[(size_t)j * np + k];
            if (!(d > 0.0) || !std::isfinite(d)) return false;
            const double ljj = std::sqrt(d);
            Lm[(size_t)j * np + j] = ljj;
            for (int i = j + 1; i < np; i++) {
                double s = Lm[(size_t)i * np + j];
                for (int k = 0; k < j; k++) s -= Lm[(size_t)i * np + k] * Lm[(size_t)j * np + k];
                Lm[(size_t)i * np + j] = s / ljj;
            }
        }
        for (int i = 0; i < np; i++) { double s = y[i]; for (int k = 0; k < i; k++) s -= Lm[(size_t)i * np + k] * y[k]; y[i] = s / Lm[(size_t)i * np + i]; }
        for (int i = np - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < np; k++) s -= Lm[(size_t)k * np + i] * y[k]; y[i] = s / Lm[(size_t)i * np + i]; }
        x.assign(np + 3 * (size_t)P.n_points, 0.0);
        for (int i = 0; i < np; i++) x[i] = y[i];
        double sc = 0;
        for (int i = 0; i < np; i++) sc += x[i] * (lambda * x[i] + bp[i]);
        for (int l = 0; l < P.n_points; l++) {
            double c[3] = {bl[3 * l], bl[3 * l + 1], bl[3 * l + 2]};
            for (int e : by_pt[l]) {
                const int o = off_pose[P.edge_kf[e]];
                if (o < 0) continue;
                for (int q = 0; q < 3; q++) { double s = 0; for (int r = 0; r < 6; r++) s += W[18 * e + 3 * r + q] * x[o + r]; c[q] -= s; }
            }
            const double* Di = &Dinv[9 * l];
            for (int a = 0; a < 3; a++) {
                const double xl = Di[3 * a] * c[0] + Di[3 * a + 1] * c[1] + Di[3 * a + 2] * c[2];
                x[np + 3 * l + a] = xl;
                sc += xl * (lambda * xl + bl[3 * l + a]);
            }
        }
        /* oplus */
        for (size_t i = 0; i < kf.size(); i++) {
            if (off_pose[i] >= 0) update_pose(kf[i], &x[off_pose[i]]);
            if (off_v[i] >= 0) for (int k = 0; k < 3; k++) kf[i].v[k] += x[off_v[i] + k];
            if (off_g[i] >= 0) for (int k = 0; k < 3; k++) kf[i].bg[k] += x[off_g[i] + k];
            if (off_a[i] >= 0) for (int k = 0; k < 3; k++) kf[i].ba[k] += x[off_a[i] + k];
        }
        for (size_t k = 0; k < pts.size(); k++) pts[k] += x[np + k];
        *scale = sc;
        return true;
    }
};

}  // namespace

extern "C" int inertial_oracle_solve(const OracleInertialProblem* P, double* Rwb_out, double* twb_out, double* vel_out, double* bg_out,
                                      double* ba_out, double* points_out, double* chi2_per_edge, uint8_t* depth_positive, OracleLbaStats* st_out)
{
    Problem pr;
    pr.p = P;
    std::memcpy(pr.Rcb, P->Rcb, sizeof(pr.Rcb)); std::memcpy(pr.tcb, P->tcb, sizeof(pr.tcb)); std::memcpy(pr.tbc, P->tbc, sizeof(pr.tbc));
    mat_tr(pr.Rcb, pr.Rbc);
    pr.kf.resize(P->n_kf);
    pr.off_pose.assign(P->n_kf, -1); pr.off_v.assign(P->n_kf, -1); pr.off_g.assign(P->n_kf, -1); pr.off_a.assign(P->n_kf, -1);
    for (int i = 0; i < P->n_kf; i++) {
        KF& k = pr.kf[i];
        std::memcpy(k.Rwb, P->Rwb + 9 * i, sizeof(k.Rwb)); std::memcpy(k.twb, P->twb + 3 * i, sizeof(k.twb));
        std::memcpy(k.v, P->vel + 3 * i, sizeof(k.v)); std::memcpy(k.bg, P->bg + 3 * i, sizeof(k.bg)); std::memcpy(k.ba, P->ba + 3 * i, sizeof(k.ba));
        k.its = 0;
        pr.camera_from_body(k);
        if (!P->pose_fixed[i]) { pr.off_pose[i] = pr.np; pr.np += 6; }
        if (P->has_imu[i] && !P->imu_fixed[i]) { pr.off_v[i] = pr.np; pr.off_g[i] = pr.np + 3; pr.off_a[i] = pr.np + 6; pr.np += 9; }
    }
    pr.pts.assign(P->points, P->points + 3 * (size_t)P->n_points);
    const int np = pr.np;
    pr.Hpp.assign((size_t)np * np, 0.0); pr.bp.assign(np, 0.0);
    pr.Hll.assign(9 * (size_t)P->n_points, 0.0); pr.bl.assign(3 * (size_t)P->n_points, 0.0); pr.W.assign(18 * (size_t)std::max(P->n_edges, 1), 0.0);
    pr.verr.assign(3 * (size_t)std::max(P->n_edges, 1), 0.0); pr.vrho1.assign(std::max(P->n_edges, 1), 1.0);
    OracleLbaStats st;
    std::memset(&st, 0, sizeof(st));
    double lambda = P->lambda_init, ni = 2;
    int nBad = 0;
    /* SparseOptimizer::optimize driving OptimizationAlgorithmLevenberg::solve, as in lba_oracle.cpp */
    for (int it = 0; it < P->max_iters; it++) {
        double currentChi = pr.errors();
        pr.linearize();
        const double iniChi = currentChi;
        if (it == 0) { st.chi2_initial = currentChi; ni = 2; nBad = 0; }
        double rho = 0;
        int qmax = 0;
        do {
            pr.saved = pr.kf; pr.pts_saved = pr.pts;               /* push() */
            double scale = 0;
            const bool ok = pr.solve_and_update(lambda, &scale);
            double tempChi = ok ? pr.errors() : std::numeric_limits<double>::max();
            rho = (currentChi - tempChi) / (scale + 1e-3);
            if (!ok) rho = -1;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;                               /* discardTop() */
            } else {
                lambda *= ni; ni *= 2;
                pr.kf = pr.saved; pr.pts = pr.pts_saved;            /* pop(): restores `its` as well */
            }
            qmax++; st.trials++;
        } while (rho < 0 && qmax < 10);
        st.iterations++;
        if (it < 16) st.chi2_trace[it] = currentChi;
        st.chi2_final = currentChi;
        if (qmax == 10 || rho == 0) { st.stop_reason = 1; break; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) { st.stop_reason = 2; break; }
    }
    st.lambda = lambda;
    pr.errors();
    for (int i = 0; i < P->n_kf; i++) {
        std::memcpy(Rwb_out + 9 * i, pr.kf[i].Rwb, 72); std::memcpy(twb_out + 3 * i, pr.kf[i].twb, 24); std::memcpy(vel_out + 3 * i, pr.kf[i].v, 24);
        std::memcpy(bg_out + 3 * i, pr.kf[i].bg, 24); std::memcpy(ba_out + 3 * i, pr.kf[i].ba, 24);
    }
    std::memcpy(points_out, pr.pts.data(), pr.pts.size() * sizeof(double));
    for (int e = 0; e < P->n_edges; e++) {
        const double* r = &pr.verr[3 * e];
        if (chi2_per_edge) chi2_per_edge[e] = P->edge_inv_sigma2[e] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        if (depth_positive) { double Xc[3]; pr.project(pr.kf[P->edge_kf[e]], &pr.pts[3 * P->edge_point[e]], Xc); depth_positive[e] = Xc[2] > 0.0; }
    }
    if (st_out) *st_out = st;
    return 0;
}

/* max |analytic - central-difference| over all inertial Jacobian entries of link l (test hook for the restated Jacobians) */
extern "C" double inertial_oracle_jacobian_check(const OracleInertialProblem* P, int l, double h)
{
    Problem pr;
    pr.p = P;
    std::memcpy(pr.Rcb, P->Rcb, sizeof(pr.Rcb)); std::memcpy(pr.tcb, P->tcb, sizeof(pr.tcb)); std::memcpy(pr.tbc, P->tbc, sizeof(pr.tbc));
    mat_tr(pr.Rcb, pr.Rbc);
    pr.kf.resize(P->n_kf);
    for (int i = 0; i < P->n_kf; i++) {
        KF& k = pr.kf[i];
        std::memcpy(k.Rwb, P->Rwb + 9 * i, 72); std::memcpy(k.twb, P->twb + 3 * i, 24); std::memcpy(k.v, P->vel + 3 * i, 24);
        std::memcpy(k.bg, P->bg + 3 * i, 24); std::memcpy(k.ba, P->ba + 3 * i, 24);
        k.its = 0; pr.camera_from_body(k);
    }
    const OracleInertialLink& L = P->links[l];
    double J[6][54];
    pr.inertial_jacobians(L, J);
    const int dim[6] = {6, 3, 3, 3, 6, 3};
    double worst = 0;
    for (int v = 0; v < 6; v++)
        for (int c = 0; c < dim[v]; c++) {
            double e1[9], e0[9];
            for (int sgn = 0; sgn < 2; sgn++) {
                std::vector<KF> keep = pr.kf;
                KF& k = pr.kf[v < 4 ? L.kf1 : L.kf2];
                double d[6] = {0, 0, 0, 0, 0, 0};
                d[c] = sgn ? h : -h;
                if (v == 0 || v == 4) { k.its = -100; pr.update_pose(k, d); }
                else if (v == 1 || v == 5) k.v[c] += d[c];
                else if (v == 2) k.bg[c] += d[c];
                else k.ba[c] += d[c];
                pr.inertial_error(L, sgn ? e1 : e0);
                pr.kf = keep;
            }
            for (int r = 0; r < 9; r++) worst = std::max(worst, std::fabs((e1[r] - e0[r]) / (2 * h) - J[v][r * dim[v] + c]));
        }
    return worst;
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * Optimizer::PoseInertialOptimizationLastKeyFrame (reference src/Optimizer.cc:4491-4873): the per-frame optimisation of the
 * inertial tracker.  15 unknowns (body pose, velocity, gyro bias, accelerometer bias of the current frame); the last key
 * frame's states are fixed.  Edges: EdgeMonoOnlyPose / EdgeStereoOnlyPose (unary, Huber), EdgeInertial, EdgeGyroRW, EdgeAccRW.
 * g2o Gauss-Newton (optimization_algorithm_gauss_newton.cpp:49-90: errors -> build -> dense solve -> update; NO step control),
 * 4 rounds x 10 iterations with re-classification of the observations after every round.  Quirk replicated: e->chi2() of an
 * ACTIVE edge is the error computed at the start of the last iteration (before the last update; nothing recomputes it), while
 * an outlier edge is recomputed at the current state (:4723-4726).  GROUNDWORK: oracle only, no HIP path yet.  PARITY UNPINNED.
 * Deviation: if the dense solve fails the round stops without applying an update (g2o applies the stale solution vector).
 * --------------------------------------------------------------------------------------------------------------------------- */
extern "C" int pose_inertial_oracle_optimize(const OraclePoseInertialProblem* P, double* Rwb_out, double* twb_out, double* vel_out,
                                             double* bg_out, double* ba_out, uint8_t* outlier, double* H15_out, int* n_bad_out)
{
    Problem pr;
    OracleInertialProblem dummy;
    std::memset(&dummy, 0, sizeof(dummy));
    pr.p = &dummy;
    std::memcpy(pr.Rcb, P->Rcb, sizeof(pr.Rcb)); std::memcpy(pr.tcb, P->tcb, sizeof(pr.tcb)); std::memcpy(pr.tbc, P->tbc, sizeof(pr.tbc));
    mat_tr(pr.Rcb, pr.Rbc);
    pr.kf.resize(2);                                        /* 0: last key frame (fixed), 1: current frame */
    for (int i = 0; i < 2; i++) {
        KF& k = pr.kf[i];
        std::memcpy(k.Rwb, P->Rwb + 9 * i, 72); std::memcpy(k.twb, P->twb + 3 * i, 24); std::memcpy(k.v, P->vel + 3 * i, 24);
        std::memcpy(k.bg, P->bg + 3 * i, 24); std::memcpy(k.ba, P->ba + 3 * i, 24);
        k.its = 0;
        pr.camera_from_body(k);
    }
    const OracleInertialLink& L = P->link;
    const int n = P->n;
    std::vector<double> err(3 * (size_t)std::max(n, 1), 0.0);
    std::vector<uint8_t> level(std::max(n, 1), 0);
    for (int i = 0; i < n; i++) outlier[i] = 0;
    bool robust = true;
    KF& F = pr.kf[1];
    auto edge_error = [&](int i, double* r) {
        double Xc[3];
        pr.project(F, P->Xw + 3 * (size_t)i, Xc);
        const double u = P->fx * Xc[0] / Xc[2] + P->cx, v = P->fy * Xc[1] / Xc[2] + P->cy;
        r[0] = P->obs[3 * i] - u; r[1] = P->obs[3 * i + 1] - v; r[2] = 0;
        if (P->stereo[i]) r[2] = P->obs[3 * i + 2] - (u - P->bf * (1 / Xc[2]));
    };
    auto edge_chi2 = [&](int i) { const double* r = &err[3 * (size_t)i]; return P->inv_sigma2[i] * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]); };
    auto edge_jac = [&](int i, double* Jj) {                /* EdgeMonoOnlyPose / EdgeStereoOnlyPose::linearizeOplus (G2oTypes.cc:380-450) */
        double Xc[3], Xb[3];
        pr.project(F, P->Xw + 3 * (size_t)i, Xc);
        mat_vec(pr.Rbc, Xc, Xb);
        for (int q = 0; q < 3; q++) Xb[q] += pr.tbc[q];
        const double iz = 1.0 / Xc[2], iz2 = 1.0 / (Xc[2] * Xc[2]);
        double pj[9] = {P->fx * iz, 0, -P->fx * Xc[0] * iz2, 0, P->fy * iz, -P->fy * Xc[1] * iz2, 0, 0, 0};
        if (P->stereo[i]) { pj[6] = pj[0]; pj[7] = pj[1]; pj[8] = pj[2] + P->bf * iz2; }
        double prb[9];
        mat_mul(pj, pr.Rcb, prb);
        const double x = Xb[0], y = Xb[1], z = Xb[2];
        const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 6; c++) Jj[6 * r + c] = prb[3 * r] * D[c] + prb[3 * r + 1] * D[6 + c] + prb[3 * r + 2] * D[12 + c];
    };
    const bool LF = P->last_frame != 0;                     /* PoseInertialOptimizationLastFrame (:4875-5285): the previous frame is free too */
    const int N = LF ? 30 : 15, oc = LF ? 15 : 0;            /* unknowns; offset of the current frame's block */
    KF& K0 = pr.kf[0];
    const float chi2MonoKF[4] = {12, 7.5, 5.991, 5.991}, chi2MonoF[4] = {5.991, 5.991, 5.991, 5.991}, chi2Stereo[4] = {15.6, 9.8, 7.815, 7.815};
    const float* chi2Mono = LF ? chi2MonoF : chi2MonoKF;
    /* EdgePriorPoseImu (G2oTypes.cc:720-760): error (er, et, ev, ebg, eba) of the previous frame's states against the prior, 15 x 15 Jacobian */
    auto prior_terms = [&](double* e15, double* J15) {
        double Rt[9], Rr[9];
        mat_tr(P->prior_Rwb, Rt);
        mat_mul(Rt, K0.Rwb, Rr);
        log_so3(Rr, e15);
        double dt3[3] = {K0.twb[0] - P->prior_twb[0], K0.twb[1] - P->prior_twb[1], K0.twb[2] - P->prior_twb[2]};
        mat_vec(Rt, dt3, e15 + 3);
        for (int q = 0; q < 3; q++) { e15[6 + q] = K0.v[q] - P->prior_vel[q]; e15[9 + q] = K0.bg[q] - P->prior_bg[q]; e15[12 + q] = K0.ba[q] - P->prior_ba[q]; }
        std::memset(J15, 0, 225 * sizeof(double));
        double iJ[9];
        inv_right_jac(e15, iJ);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { J15[15 * r + c] = iJ[3 * r + c]; J15[15 * (3 + r) + 3 + c] = Rr[3 * r + c]; }
        for (int q = 6; q < 15; q++) J15[15 * q + q] = 1.0;
    };
    int nBad = 0, nInliers = 0;
    for (int it = 0; it < 4; it++) {
        for (int k = 0; k < 10; k++) {                       /* optimize(its[it]): Gauss-Newton */
            double H[900], b[30];
            std::memset(H, 0, sizeof(H)); std::memset(b, 0, sizeof(b));
            for (int i = 0; i < n; i++) {
                if (level[i]) continue;
                edge_error(i, &err[3 * (size_t)i]);
                const int ne = P->stereo[i] ? 3 : 2;
                const double c = edge_chi2(i), delta = P->stereo[i] ? P->huber_stereo : P->huber_mono;
                const double rho1 = (robust && c > delta * delta) ? delta / std::sqrt(c) : 1.0;
                const double w = P->inv_sigma2[i] * rho1;
                double Jj[18];
                edge_jac(i, Jj);
                const double* r = &err[3 * (size_t)i];
                for (int a = 0; a < 6; a++) {
                    for (int c2 = 0; c2 < 6; c2++) { double s2 = 0; for (int q = 0; q < ne; q++) s2 += Jj[6 * q + a] * Jj[6 * q + c2]; H[N * (oc + a) + oc + c2] += w * s2; }
                    double s2 = 0; for (int q = 0; q < ne; q++) s2 += Jj[6 * q + a] * r[q];
                    b[oc + a] -= w * s2;
                }
            }
            double e9[9], J[6][54];
            pr.inertial_error(L, e9);
            pr.inertial_jacobians(L, J);
            {   /* the link: all six vertices when the previous frame is free, (pose, velocity) of the current frame otherwise */
                const int loc[6] = {0, 6, 9, 12, 15, 21}, dim[6] = {6, 3, 3, 3, 6, 3};
                const int goff[6] = {LF ? 0 : -1, LF ? 6 : -1, LF ? 9 : -1, LF ? 12 : -1, oc, oc + 6};
                (void)loc;
                double Oe[9];
                for (int r = 0; r < 9; r++) { double t = 0; for (int q = 0; q < 9; q++) t += L.info9[9 * r + q] * e9[q]; Oe[r] = t; }
                for (int va = 0; va < 6; va++) {
                    if (goff[va] < 0) continue;
                    for (int a = 0; a < dim[va]; a++) {
                        double s2 = 0;
                        for (int r = 0; r < 9; r++) s2 += J[va][r * dim[va] + a] * Oe[r];
                        b[goff[va] + a] -= s2;
                        for (int vb = 0; vb < 6; vb++) {
                            if (goff[vb] < 0) continue;
                            for (int c = 0; c < dim[vb]; c++) {
                                double h = 0;
                                for (int r = 0; r < 9; r++) { double t = 0; for (int q = 0; q < 9; q++) t += L.info9[9 * r + q] * J[vb][q * dim[vb] + c]; h += J[va][r * dim[va] + a] * t; }
                                H[N * (goff[va] + a) + goff[vb] + c] += h;
                            }
                        }
                    }
                }
            }
            for (int which = 0; which < 2; which++) {        /* random walks: e = b_cur - b_prev, J_prev = -I, J_cur = +I */
                const double* Om3 = which ? L.info_acc : L.info_gyro;
                const int o2 = oc + (which ? 12 : 9), o1 = LF ? (which ? 12 : 9) : -1;
                double d[3], Od[3];
                for (int q = 0; q < 3; q++) d[q] = which ? F.ba[q] - K0.ba[q] : F.bg[q] - K0.bg[q];
                mat_vec(Om3, d, Od);
                for (int a = 0; a < 3; a++) {
                    for (int c = 0; c < 3; c++) {
                        H[N * (o2 + a) + o2 + c] += Om3[3 * a + c];
                        if (o1 >= 0) { H[N * (o1 + a) + o1 + c] += Om3[3 * a + c]; H[N * (o1 + a) + o2 + c] -= Om3[3 * a + c]; H[N * (o2 + a) + o1 + c] -= Om3[3 * a + c]; }
                    }
                    b[o2 + a] -= Od[a];
                    if (o1 >= 0) b[o1 + a] += Od[a];
                }
            }
            if (LF) {                                        /* the prior on the previous frame, Huber delta 5 (:5084-5090) */
                double e15[15], J15[225], Oe[15];
                prior_terms(e15, J15);
                double c = 0;
                for (int r = 0; r < 15; r++) { double t = 0; for (int q = 0; q < 15; q++) t += P->prior_H[15 * r + q] * e15[q]; Oe[r] = t; c += e15[r] * t; }
                const double rho1 = c > 25.0 ? 5.0 / std::sqrt(c) : 1.0;
                for (int a = 0; a < 15; a++) {
                    double s2 = 0;
                    for (int r = 0; r < 15; r++) s2 += J15[15 * r + a] * Oe[r];
                    b[a] -= rho1 * s2;
                    for (int c2 = 0; c2 < 15; c2++) {
                        double h = 0;
                        for (int r = 0; r < 15; r++) { double t = 0; for (int q = 0; q < 15; q++) t += P->prior_H[15 * r + q] * J15[15 * q + c2]; h += J15[15 * r + a] * t; }
                        H[N * a + c2] += rho1 * h;
                    }
                }
            }
            /* dense Cholesky solve H x = b */
            double Lm[900], x[30];
            std::memcpy(Lm, H, sizeof(double) * N * N);
            bool ok = true;
            for (int j = 0; j < N && ok; j++) {
                double d = Lm[N * j + j];
                for (int q = 0; q < j; q++) d -= Lm[N * j + q] * Lm[N * j + q];
                if (!(d > 0.0) || !std::isfinite(d)) { ok = false; break; }
                const double ljj = std::sqrt(d);
                Lm[N * j + j] = ljj;
                for (int i = j + 1; i < N; i++) { double s2 = Lm[N * i + j]; for (int q = 0; q < j; q++) s2 -= Lm[N * i + q] * Lm[N * j + q]; Lm[N * i + j] = s2 / ljj; }
            }
            if (!ok) break;
            for (int i = 0; i < N; i++) { double s2 = b[i]; for (int q = 0; q < i; q++) s2 -= Lm[N * i + q] * x[q]; x[i] = s2 / Lm[N * i + i]; }
            for (int i = N - 1; i >= 0; i--) { double s2 = x[i]; for (int q = i + 1; q < N; q++) s2 -= Lm[N * q + i] * x[q]; x[i] = s2 / Lm[N * i + i]; }
            pr.update_pose(F, x + oc);
            for (int q = 0; q < 3; q++) { F.v[q] += x[oc + 6 + q]; F.bg[q] += x[oc + 9 + q]; F.ba[q] += x[oc + 12 + q]; }
            if (LF) {
                pr.update_pose(K0, x);
                for (int q = 0; q < 3; q++) { K0.v[q] += x[6 + q]; K0.bg[q] += x[9 + q]; K0.ba[q] += x[12 + q]; }
            }
        }
        nBad = 0; nInliers = 0;
        const float chi2close = 1.5 * chi2Mono[it];
        for (int pass = 0; pass < 2; pass++)                 /* mono edges first, then stereo (:4716-4789); the order only matters for counters */
            for (int i = 0; i < n; i++) {
                if ((P->stereo[i] != 0) != (pass == 1)) continue;
                if (outlier[i]) edge_error(i, &err[3 * (size_t)i]);
                const float chi2 = (float)edge_chi2(i);
                bool bad;
                if (!P->stereo[i]) {
                    double Xc[3];
                    pr.project(F, P->Xw + 3 * (size_t)i, Xc);
                    const bool bClose = P->close_point[i] != 0;
                    bad = (chi2 > chi2Mono[it] && !bClose) || (bClose && chi2 > chi2close) || !(Xc[2] > 0.0);
                } else
                    bad = chi2 > chi2Stereo[it];
                outlier[i] = bad; level[i] = bad;
                if (bad) nBad++; else nInliers++;
            }
        if (it == 2) robust = false;
        if (n + (LF ? 4 : 3) < 10) break;                    /* optimizer.edges().size() < 10 (:4794, :5200) */
    }
    if (nInliers < 30 && !P->rec_init) {                     /* recover not too bad points (:4802-4828) */
        nBad = 0;
        for (int i = 0; i < n; i++) {
            edge_error(i, &err[3 * (size_t)i]);
            if ((float)edge_chi2(i) < (P->stereo[i] ? 24.f : 18.f)) outlier[i] = 0; else nBad++;
        }
    }
    /* Hessian of the new prior (:4837-4870): inertial edge w.r.t. (pose, velocity), the random-walk informations, the inliers' 6 x 6 */
    double Hout[900];
    std::memset(Hout, 0, sizeof(Hout));
    {
        double J[6][54];
        pr.inertial_jacobians(L, J);
        const int dim[6] = {6, 3, 3, 3, 6, 3};
        const int goff[6] = {LF ? 0 : -1, LF ? 6 : -1, LF ? 9 : -1, LF ? 12 : -1, oc, oc + 6};      /* ei->GetHessian() / GetHessian2() */
        for (int va = 0; va < 6; va++) {
            if (goff[va] < 0) continue;
            for (int a = 0; a < dim[va]; a++)
                for (int vb = 0; vb < 6; vb++) {
                    if (goff[vb] < 0) continue;
                    for (int c = 0; c < dim[vb]; c++) {
                        double h = 0;
                        for (int r = 0; r < 9; r++) { double t = 0; for (int q = 0; q < 9; q++) t += L.info9[9 * r + q] * J[vb][q * dim[vb] + c]; h += J[va][r * dim[va] + a] * t; }
                        Hout[N * (goff[va] + a) + goff[vb] + c] += h;
                    }
                }
        }
        for (int which = 0; which < 2; which++) {
            const double* Om3 = which ? L.info_acc : L.info_gyro;
            const int o2 = oc + (which ? 12 : 9), o1 = LF ? (which ? 12 : 9) : -1;
            for (int a = 0; a < 3; a++)
                for (int c = 0; c < 3; c++) {
                    Hout[N * (o2 + a) + o2 + c] += Om3[3 * a + c];
                    if (o1 >= 0) { Hout[N * (o1 + a) + o1 + c] += Om3[3 * a + c]; Hout[N * (o1 + a) + o2 + c] -= Om3[3 * a + c]; Hout[N * (o2 + a) + o1 + c] -= Om3[3 * a + c]; }
                }
        }
        if (LF) {                                            /* ep->GetHessian(): J^T information J, no robust weight */
            double e15[15], J15[225];
            prior_terms(e15, J15);
            for (int a = 0; a < 15; a++)
                for (int c2 = 0; c2 < 15; c2++) {
                    double h = 0;
                    for (int r = 0; r < 15; r++) { double t = 0; for (int q = 0; q < 15; q++) t += P->prior_H[15 * r + q] * J15[15 * q + c2]; h += J15[15 * r + a] * t; }
                    Hout[N * a + c2] += h;
                }
        }
        for (int i = 0; i < n; i++) {
            if (outlier[i]) continue;
            const int ne = P->stereo[i] ? 3 : 2;
            double Jj[18];
            edge_jac(i, Jj);
            for (int a = 0; a < 6; a++)
                for (int c = 0; c < 6; c++) { double s2 = 0; for (int q = 0; q < ne; q++) s2 += Jj[6 * q + a] * Jj[6 * q + c]; Hout[N * (oc + a) + oc + c] += P->inv_sigma2[i] * s2; }
        }
    }
    std::memcpy(Rwb_out, F.Rwb, 72); std::memcpy(twb_out, F.twb, 24); std::memcpy(vel_out, F.v, 24); std::memcpy(bg_out, F.bg, 24); std::memcpy(ba_out, F.ba, 24);
    if (H15_out) std::memcpy(H15_out, Hout, sizeof(double) * N * N);     /* 15 x 15, or 30 x 30 (before Optimizer::Marginalize) for the last-frame variant */
    if (n_bad_out) *n_bad_out = nBad;
    return n - nBad;
}
