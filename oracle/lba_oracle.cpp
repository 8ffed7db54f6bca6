/*
 * ORACLE (test infrastructure, not product) -- CPU restatement of the numerical core of
 * Optimizer::LocalBundleAdjustment: the g2o Levenberg-Marquardt loop over
 * EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ edges with Huber kernels, BlockSolver_6_3 with
 * Schur complement, on a flattened (SoA) problem.
 *
 * Follows (read as text; nothing copied):
 *   /root/reference/src/Optimizer.cc:1188-1411               solver set-up, vertices, edges, optimize(10)
 *   /root/reference/src/OptimizableTypes.cpp:139-160, include/OptimizableTypes.h:99-110   mono edge
 *   /root/reference/src/CameraModels/Pinhole.cpp:43-49,71-81  project / projectJac (double)
 *   Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:190-274 stereo edge (float invz quirk)
 *   Thirdparty/g2o/g2o/types/se3quat.h:41-296, se3_ops.hpp    SE3Quat, exp, skew
 *   Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:73-76, types_sba.h:52-56  oplus
 *   Thirdparty/g2o/g2o/core/base_binary_edge.hpp:55-120       constructQuadraticForm (robust branch)
 *   Thirdparty/g2o/g2o/core/base_edge.h:58-102, robust_kernel_impl.cpp:65-91   chi2, Huber
 *   Thirdparty/g2o/g2o/core/block_solver.hpp:354-604          Schur solve, buildSystem, setLambda, restoreDiagonal
 *   Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-194  LM control
 *   Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:61-114,354-435           errors, chi2, optimize, update
 * Third-party arithmetic NOT in /root/reference: Eigen (3x3 inverse, Quaterniond(Matrix3d),
 * SimplicialLDLT).  Restated with a closed-form 3x3 inverse, Shepperd's method and a dense
 * LDL^T -- results agree with any accurate implementation far below the 1e-4 bar.
 * PARITY UNPINNED (no reference tests; Eigen/g2o not buildable here).
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Quat { double x, y, z, w; };
struct Pose { Quat q; double t[3]; };

void quat_normalize(Quat& q)     // SE3Quat::normalizeRotation (se3quat.h:280-285)
{
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}

void quat_to_R(const Quat& q, double R[9])   // Eigen QuaternionBase::toRotationMatrix
{
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

Quat quat_from_R(const double R[9])          // Eigen Quaternion(Matrix3) (Shepperd)
{
    Quat q;
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (R[7] - R[5]) * t;
        q.y = (R[2] - R[6]) * t;
        q.z = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
        v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}

void quat_rotate(const Quat& q, const double v[3], double out[3])   // Eigen _transformVector
{
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
    out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
    out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}

Quat quat_mul(const Quat& a, const Quat& b)
{
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

void pose_map(const Pose& T, const double X[3], double out[3])   // SE3Quat::map (:217-220)
{
    quat_rotate(T.q, X, out);
    out[0] += T.t[0]; out[1] += T.t[1]; out[2] += T.t[2];
}

// SE3Quat::exp (se3quat.h:223-257); update = (omega, upsilon)
Pose se3_exp(const double u[6])
{
    const double om[3] = {u[0], u[1], u[2]};
    const double up[3] = {u[3], u[4], u[5]};
    const double theta = std::sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += O[i * 3 + k] * O[k * 3 + j];
            O2[i * 3 + j] = s;
        }
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }   // quirk: I+O+O^2, V=R
    } else {
        const double a = std::sin(theta) / theta;
        const double b = (1 - std::cos(theta)) / (theta * theta);
        const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
        for (int i = 0; i < 9; i++) {
            const double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    Pose p;
    p.q = quat_from_R(R);
    for (int i = 0; i < 3; i++) p.t[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
    quat_normalize(p.q);           // SE3Quat(Quaterniond, Vector3d) ctor
    return p;
}

Pose pose_mul(const Pose& a, const Pose& b)  // SE3Quat::operator* (:104-110)
{
    Pose r = a;
    double rt[3];
    quat_rotate(a.q, b.t, rt);
    r.t[0] += rt[0]; r.t[1] += rt[1]; r.t[2] += rt[2];
    r.q = quat_mul(a.q, b.q);
    quat_normalize(r.q);
    return r;
}

bool inv3(const double A[9], double Ai[9])   // closed-form inverse (Eigen compute_inverse_size3 equivalent)
{
    const double c00 = A[4] * A[8] - A[5] * A[7];
    const double c01 = A[5] * A[6] - A[3] * A[8];
    const double c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    const double id = 1.0 / det;
    Ai[0] = c00 * id; Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id; Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Ai[3] = c01 * id; Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id; Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Ai[6] = c02 * id; Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id; Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    return det != 0;
}

// dense LDL^T solve of an SPD system (stand-in for Eigen::SimplicialLDLT, linear_solver_eigen.h:94-124)
bool ldlt_solve(std::vector<double>& A, int n, const double* b, double* x)
{
    std::vector<double> D(n);
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * D[k];
        if (d == 0 || !std::isfinite(d)) return false;
        D[j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * D[k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * x[k];
        x[i] = s;
    }
    for (int i = 0; i < n; i++) x[i] /= D[i];
    for (int i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * x[k];
        x[i] = s;
    }
    return true;
}

struct Lba {
    const OracleLbaProblem* p;
    std::vector<Pose> poses;
    std::vector<double> pts;
    std::vector<int> pose_col;      // index among non-fixed poses, -1 if fixed
    int nP, nL, nE;
    double dsqr_mono, dsqr_stereo;
    std::vector<double> err;        // 3 per edge (edge._error as of the last computeActiveErrors)
    std::vector<double> Hpp, Hll, Hpl, bp, bl;   // Hpp nP x 36, Hll nL x 9, Hpl nE x 18 (6x3 row-major), b
    std::vector<double> x;          // solution [6nP | 3nL]

    void init(const OracleLbaProblem* pr)
    {
        p = pr;
        poses.resize(p->n_poses);
        pose_col.assign(p->n_poses, -1);
        nP = 0;
        for (int i = 0; i < p->n_poses; i++) {
            Pose& T = poses[i];
            T.q.x = p->pose_q[4 * i]; T.q.y = p->pose_q[4 * i + 1]; T.q.z = p->pose_q[4 * i + 2]; T.q.w = p->pose_q[4 * i + 3];
            for (int k = 0; k < 3; k++) T.t[k] = p->pose_t[3 * i + k];
            quat_normalize(T.q);    // SE3Quat(Quaterniond, Vector3d) (Optimizer.cc:1217)
            if (!p->pose_fixed[i]) pose_col[i] = nP++;
        }
        nL = p->n_points;
        nE = p->n_edges;
        pts.assign(p->points, p->points + 3 * (size_t)nL);
        dsqr_mono = p->huber_mono * p->huber_mono;        // RobustKernelHuber::setDelta
        dsqr_stereo = p->huber_stereo * p->huber_stereo;
        err.assign(3 * (size_t)nE, 0.0);
        Hpp.resize((size_t)nP * 36); Hll.resize((size_t)nL * 9); Hpl.resize((size_t)nE * 18);
        bp.resize((size_t)nP * 6); bl.resize((size_t)nL * 3);
        x.assign((size_t)nP * 6 + (size_t)nL * 3, 0.0);
    }

    int edim(int e) const { return p->edge_stereo[e] ? 3 : 2; }

    void compute_errors()       // SparseOptimizer::computeActiveErrors
    {
        for (int e = 0; e < nE; e++) {
            double Xc[3];
            pose_map(poses[p->edge_pose[e]], &pts[3 * (size_t)p->edge_point[e]], Xc);
            const double* obs = p->edge_obs + 3 * (size_t)e;
            double* r = &err[3 * (size_t)e];
            if (!p->edge_stereo[e]) {
                r[0] = obs[0] - (p->fx * Xc[0] / Xc[2] + p->cx);     // Pinhole::project(Vector3d)
                r[1] = obs[1] - (p->fy * Xc[1] / Xc[2] + p->cy);
                r[2] = 0;
            } else {
                const float invz = 1.0f / Xc[2];                     // float quirk (types_six_dof_expmap.cpp:191)
                const double u = Xc[0] * invz * p->fx + p->cx;
                const double v = Xc[1] * invz * p->fy + p->cy;
                const float bf_f = (float)p->bf;
                const double ur = u - bf_f * invz;
                r[0] = obs[0] - u; r[1] = obs[1] - v; r[2] = obs[2] - ur;
            }
        }
    }

    double edge_chi2(int e) const   // BaseEdge::chi2, information = invSigma2 * I
    {
        const double* r = &err[3 * (size_t)e];
        const double w = p->edge_inv_sigma2[e];
        double c = r[0] * (w * r[0]) + r[1] * (w * r[1]);
        if (p->edge_stereo[e]) c += r[2] * (w * r[2]);
        return c;
    }

    void robustify(int e, double chi, double rho[3]) const
    {
        const double delta = p->edge_stereo[e] ? p->huber_stereo : p->huber_mono;
        const double dsqr = p->edge_stereo[e] ? dsqr_stereo : dsqr_mono;
        if (delta <= 0) { rho[0] = chi; rho[1] = 1; rho[2] = 0; return; }
        if (chi <= dsqr) { rho[0] = chi; rho[1] = 1.; rho[2] = 0.; }
        else {
            const double sqrte = std::sqrt(chi);
            rho[0] = 2 * sqrte * delta - dsqr;
            rho[1] = delta / sqrte;
            rho[2] = -0.5 * rho[1] / chi;
        }
    }

    double robust_chi2() const      // SparseOptimizer::activeRobustChi2
    {
        double chi = 0, rho[3];
        for (int e = 0; e < nE; e++) { robustify(e, edge_chi2(e), rho); chi += rho[0]; }
        return chi;
    }

    void build_system()             // BlockSolver::buildSystem
    {
        std::fill(Hpp.begin(), Hpp.end(), 0.0); std::fill(Hll.begin(), Hll.end(), 0.0);
        std::fill(Hpl.begin(), Hpl.end(), 0.0); std::fill(bp.begin(), bp.end(), 0.0); std::fill(bl.begin(), bl.end(), 0.0);
        for (int e = 0; e < nE; e++) {
            const int ip = p->edge_pose[e], il = p->edge_point[e];
            const Pose& T = poses[ip];
            double Xc[3], R[9];
            pose_map(T, &pts[3 * (size_t)il], Xc);
            quat_to_R(T.q, R);
            const double xx = Xc[0], yy = Xc[1], zz = Xc[2];
            const int D = edim(e);
            double Ji[9], Jj[18];       // Ji: D x 3 (point), Jj: D x 6 (pose)
            if (!p->edge_stereo[e]) {
                // -projectJac (Pinhole.cpp:71-81), OptimizableTypes.cpp:149-159
                const double pj[6] = {-(p->fx / zz), -0.0, -(-p->fx * xx / (zz * zz)),
                                      -0.0, -(p->fy / zz), -(-p->fy * yy / (zz * zz))};
                for (int r = 0; r < 2; r++)
                    for (int c = 0; c < 3; c++)
                        Ji[r * 3 + c] = pj[r * 3] * R[c] + pj[r * 3 + 1] * R[3 + c] + pj[r * 3 + 2] * R[6 + c];
                const double S[18] = {0, zz, -yy, 1, 0, 0,  -zz, 0, xx, 0, 1, 0,  yy, -xx, 0, 0, 0, 1};
                for (int r = 0; r < 2; r++)
                    for (int c = 0; c < 6; c++)
                        Jj[r * 6 + c] = pj[r * 3] * S[c] + pj[r * 3 + 1] * S[6 + c] + pj[r * 3 + 2] * S[12 + c];
            } else {
                const double z2 = zz * zz, fx = p->fx, fy = p->fy, bf = p->bf;
                Ji[0] = -fx * R[0] / zz + fx * xx * R[6] / z2;
                Ji[1] = -fx * R[1] / zz + fx * xx * R[7] / z2;
                Ji[2] = -fx * R[2] / zz + fx * xx * R[8] / z2;
                Ji[3] = -fy * R[3] / zz + fy * yy * R[6] / z2;
                Ji[4] = -fy * R[4] / zz + fy * yy * R[7] / z2;
                Ji[5] = -fy * R[5] / zz + fy * yy * R[8] / z2;
                Ji[6] = Ji[0] - bf * R[6] / z2;
                Ji[7] = Ji[1] - bf * R[7] / z2;
                Ji[8] = Ji[2] - bf * R[8] / z2;
                Jj[0] = xx * yy / z2 * fx;  Jj[1] = -(1 + (xx * xx / z2)) * fx; Jj[2] = yy / zz * fx;
                Jj[3] = -1. / zz * fx;      Jj[4] = 0;                          Jj[5] = xx / z2 * fx;
                Jj[6] = (1 + yy * yy / z2) * fy; Jj[7] = -xx * yy / z2 * fy;    Jj[8] = -xx / zz * fy;
                Jj[9] = 0;                  Jj[10] = -1. / zz * fy;             Jj[11] = yy / z2 * fy;
                Jj[12] = Jj[0] - bf * yy / z2; Jj[13] = Jj[1] + bf * xx / z2;   Jj[14] = Jj[2];
                Jj[15] = Jj[3];             Jj[16] = 0;                         Jj[17] = Jj[5] - bf / z2;
            }
            // constructQuadraticForm, robust branch (base_binary_edge.hpp:91-114)
            const double w = p->edge_inv_sigma2[e];
            const double* r = &err[3 * (size_t)e];
            double rho[3];
            robustify(e, edge_chi2(e), rho);
            const double wr = rho[1] * w;                 // weightedOmega = rho[1] * information
            double omega_r[3];
            for (int d = 0; d < D; d++) omega_r[d] = (-(w * r[d])) * rho[1];
            // from = point (never fixed), to = pose
            double* Hl = &Hll[(size_t)il * 9];
            double* bL = &bl[(size_t)il * 3];
            for (int a = 0; a < 3; a++) {
                double s = 0;
                for (int d = 0; d < D; d++) s += Ji[d * 3 + a] * omega_r[d];
                bL[a] += s;
                for (int b = 0; b < 3; b++) {
                    double h = 0;
                    for (int d = 0; d < D; d++) h += Ji[d * 3 + a] * wr * Ji[d * 3 + b];
                    Hl[a * 3 + b] += h;
                }
            }
            const int pc = pose_col[ip];
            if (pc >= 0) {
                double* Hp = &Hpp[(size_t)pc * 36];
                double* bP = &bp[(size_t)pc * 6];
                double* W = &Hpl[(size_t)e * 18];          // 6x3: B^T * wOmega * A
                for (int a = 0; a < 6; a++) {
                    double s = 0;
                    for (int d = 0; d < D; d++) s += Jj[d * 6 + a] * omega_r[d];
                    bP[a] += s;
                    for (int b = 0; b < 6; b++) {
                        double h = 0;
                        for (int d = 0; d < D; d++) h += Jj[d * 6 + a] * wr * Jj[d * 6 + b];
                        Hp[a * 6 + b] += h;
                    }
                    for (int b = 0; b < 3; b++) {
                        double h = 0;
                        for (int d = 0; d < D; d++) h += Jj[d * 6 + a] * wr * Ji[d * 3 + b];
                        W[a * 3 + b] += h;
                    }
                }
            }
        }
    }

    double max_diagonal() const     // computeLambdaInit (levenberg.cpp:171-185)
    {
        double m = 0;
        for (int i = 0; i < nP; i++) for (int j = 0; j < 6; j++) m = std::max(std::fabs(Hpp[(size_t)i * 36 + j * 7]), m);
        for (int i = 0; i < nL; i++) for (int j = 0; j < 3; j++) m = std::max(std::fabs(Hll[(size_t)i * 9 + j * 4]), m);
        return m;
    }

    // BlockSolver::solve with Schur complement (block_solver.hpp:354-486); lambda added to all diagonals
    bool solve(double lambda)
    {
        const int n = 6 * nP;
        std::vector<double> S((size_t)n * n, 0.0), bs(n), Dinv((size_t)nL * 9);
        for (int i = 0; i < nP; i++)
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++)
                    S[(size_t)(6 * i + a) * n + 6 * i + b] = Hpp[(size_t)i * 36 + a * 6 + b] + (a == b ? lambda : 0.0);
        for (int i = 0; i < n; i++) bs[i] = bp[i];
        // edges grouped by landmark
        std::vector<std::vector<int> > by_l(nL);
        for (int e = 0; e < nE; e++) if (pose_col[p->edge_pose[e]] >= 0) by_l[p->edge_point[e]].push_back(e);
        for (int l = 0; l < nL; l++) {
            double Dm[9];
            for (int k = 0; k < 9; k++) Dm[k] = Hll[(size_t)l * 9 + k] + (k % 4 == 0 ? lambda : 0.0);
            double* Di = &Dinv[(size_t)l * 9];
            inv3(Dm, Di);
            double db[3];
            for (int a = 0; a < 3; a++) db[a] = Di[a * 3] * bl[3 * l] + Di[a * 3 + 1] * bl[3 * l + 1] + Di[a * 3 + 2] * bl[3 * l + 2];
            const std::vector<int>& es = by_l[l];
            for (size_t a = 0; a < es.size(); a++) {
                const double* Bi = &Hpl[(size_t)es[a] * 18];
                const int i1 = pose_col[p->edge_pose[es[a]]];
                double BD[18];
                for (int r = 0; r < 6; r++)
                    for (int c = 0; c < 3; c++)
                        BD[r * 3 + c] = Bi[r * 3] * Di[c] + Bi[r * 3 + 1] * Di[3 + c] + Bi[r * 3 + 2] * Di[6 + c];
                for (int r = 0; r < 6; r++) bs[6 * i1 + r] -= Bi[r * 3] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
                for (size_t b = 0; b < es.size(); b++) {
                    const double* Bj = &Hpl[(size_t)es[b] * 18];
                    const int i2 = pose_col[p->edge_pose[es[b]]];
                    for (int r = 0; r < 6; r++)
                        for (int c = 0; c < 6; c++)
                            S[(size_t)(6 * i1 + r) * n + 6 * i2 + c] -= BD[r * 3] * Bj[c * 3] + BD[r * 3 + 1] * Bj[c * 3 + 1] + BD[r * 3 + 2] * Bj[c * 3 + 2];
                }
            }
        }
        if (n > 0 && !ldlt_solve(S, n, bs.data(), x.data())) return false;
        // landmarks: xl = Dinv (bl - Hpl^T xp)
        for (int l = 0; l < nL; l++) {
            double c[3] = {bl[3 * l], bl[3 * l + 1], bl[3 * l + 2]};
            const std::vector<int>& es = by_l[l];
            for (size_t a = 0; a < es.size(); a++) {
                const double* B = &Hpl[(size_t)es[a] * 18];
                const double* xp = &x[6 * (size_t)pose_col[p->edge_pose[es[a]]]];
                for (int k = 0; k < 3; k++) {
                    double s = 0;
                    for (int r = 0; r < 6; r++) s += B[r * 3 + k] * xp[r];
                    c[k] -= s;
                }
            }
            const double* Di = &Dinv[(size_t)l * 9];
            for (int a = 0; a < 3; a++) x[(size_t)n + 3 * l + a] = Di[a * 3] * c[0] + Di[a * 3 + 1] * c[1] + Di[a * 3 + 2] * c[2];
        }
        return true;
    }

    void update()                   // SparseOptimizer::update -> oplus
    {
        for (int i = 0; i < p->n_poses; i++) {
            const int c = pose_col[i];
            if (c < 0) continue;
            poses[i] = pose_mul(se3_exp(&x[6 * (size_t)c]), poses[i]);
        }
        const size_t off = 6 * (size_t)nP;
        for (size_t k = 0; k < 3 * (size_t)nL; k++) pts[k] += x[off + k];
    }

    double compute_scale(double lambda) const   // levenberg.cpp:187-194
    {
        double scale = 0;
        const size_t np = 6 * (size_t)nP;
        for (size_t j = 0; j < np; j++) scale += x[j] * (lambda * x[j] + bp[j]);
        for (size_t j = 0; j < 3 * (size_t)nL; j++) scale += x[np + j] * (lambda * x[np + j] + bl[j]);
        return scale;
    }
};

}  // namespace

extern "C" int lba_oracle_solve(const OracleLbaProblem* pr, const volatile uint8_t* stop_flag, int max_iters, double lambda_init,
                                double* poses_q_out, double* poses_t_out, double* points_out,
                                double* chi2_per_edge, uint8_t* depth_positive, OracleLbaStats* st)
{
    Lba s;
    s.init(pr);
    OracleLbaStats stats;
    std::memset(&stats, 0, sizeof(stats));
    double lambda = -1, ni = 2;
    int nBad = 0;
    auto terminate = [&]() { return stop_flag && *stop_flag; };
    stats.stop_reason = 0;
    // SparseOptimizer::optimize (:354-419)
    for (int it = 0; it < max_iters; it++) {
        if (terminate()) { stats.stop_reason = 3; break; }
        // OptimizationAlgorithmLevenberg::solve (:61-169)
        s.compute_errors();
        double currentChi = s.robust_chi2();
        double tempChi = currentChi;
        const double iniChi = currentChi;
        if (it == 0) stats.chi2_initial = currentChi;
        s.build_system();
        if (it == 0) {
            lambda = lambda_init > 0 ? lambda_init : 1e-5 * s.max_diagonal();
            ni = 2; nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        bool stopped = false;
        do {
            std::vector<Pose> backup_poses = s.poses;       // push()
            std::vector<double> backup_pts = s.pts;
            const bool ok2 = s.solve(lambda);
            if (ok2) s.update();
            s.compute_errors();
            tempChi = s.robust_chi2();
            if (!ok2) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double scale = s.compute_scale(lambda);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                const double scaleFactor = std::max(1. / 3., alpha);
                lambda *= scaleFactor;
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                s.poses = backup_poses;                     // pop()
                s.pts = backup_pts;
            }
            qmax++;
            stats.trials++;
            stopped = terminate();
        } while (rho < 0 && qmax < 10 && !stopped);
        stats.iterations++;
        if (it < 16) stats.chi2_trace[it] = currentChi;
        stats.chi2_final = currentChi;
        if (qmax == 10 || rho == 0) { stats.stop_reason = 1; break; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) { stats.stop_reason = 2; break; }
    }
    stats.lambda = lambda;
    for (int i = 0; i < pr->n_poses; i++) {
        poses_q_out[4 * i] = s.poses[i].q.x; poses_q_out[4 * i + 1] = s.poses[i].q.y;
        poses_q_out[4 * i + 2] = s.poses[i].q.z; poses_q_out[4 * i + 3] = s.poses[i].q.w;
        for (int k = 0; k < 3; k++) poses_t_out[3 * i + k] = s.poses[i].t[k];
    }
    std::memcpy(points_out, s.pts.data(), sizeof(double) * 3 * (size_t)s.nL);
    // epilogue inputs (Optimizer.cc:1417-1460): chi2 of the edge's stored error, depth from current estimates
    for (int e = 0; e < s.nE; e++) {
        if (chi2_per_edge) chi2_per_edge[e] = s.edge_chi2(e);
        if (depth_positive) {
            double Xc[3];
            pose_map(s.poses[pr->edge_pose[e]], &s.pts[3 * (size_t)pr->edge_point[e]], Xc);
            depth_positive[e] = Xc[2] > 0.0;
        }
    }
    if (st) *st = stats;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Sharded form of the same algorithm (landmarks partitioned over ranks, poses replicated).
// Used by the world_size-2 gloo tests to check the distributed LM driver against the single-rank oracle.
// Reduce buffer layout = the product's: [ S (n x n) | b_schur (n) | b_p (n) | diag(Hpp) (n) ], all additive.
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct OracleShard {
    Lba s;
    OracleLbaProblem prob;
    std::vector<double> q, t, pts, obs, w;
    std::vector<uint8_t> fixed, stereo;
    std::vector<int32_t> ep, eo;
    std::vector<double> reduce;
    std::vector<double> Dinv;
    std::vector<std::vector<int> > by_l;
    std::vector<Pose> trial_poses;
    std::vector<double> trial_pts;
    int n;
};

}  // namespace

extern "C" {

void* lba_oracle_shard_create(const OracleLbaProblem* p)
{
    OracleShard* o = new OracleShard();
    o->q.assign(p->pose_q, p->pose_q + 4 * (size_t)p->n_poses);
    o->t.assign(p->pose_t, p->pose_t + 3 * (size_t)p->n_poses);
    o->fixed.assign(p->pose_fixed, p->pose_fixed + p->n_poses);
    o->pts.assign(p->points, p->points + 3 * (size_t)p->n_points);
    o->ep.assign(p->edge_point, p->edge_point + p->n_edges);
    o->eo.assign(p->edge_pose, p->edge_pose + p->n_edges);
    o->obs.assign(p->edge_obs, p->edge_obs + 3 * (size_t)p->n_edges);
    o->w.assign(p->edge_inv_sigma2, p->edge_inv_sigma2 + p->n_edges);
    o->stereo.assign(p->edge_stereo, p->edge_stereo + p->n_edges);
    o->prob = *p;
    o->prob.pose_q = o->q.data(); o->prob.pose_t = o->t.data(); o->prob.pose_fixed = o->fixed.data();
    o->prob.points = o->pts.data(); o->prob.edge_point = o->ep.data(); o->prob.edge_pose = o->eo.data();
    o->prob.edge_obs = o->obs.data(); o->prob.edge_inv_sigma2 = o->w.data(); o->prob.edge_stereo = o->stereo.data();
    o->s.init(&o->prob);
    o->n = 6 * o->s.nP;
    o->reduce.assign((size_t)o->n * o->n + 3 * (size_t)o->n, 0.0);
    o->Dinv.assign(9 * (size_t)o->s.nL, 0.0);
    o->by_l.resize(o->s.nL);
    for (int e = 0; e < o->s.nE; e++)
        if (o->s.pose_col[o->prob.edge_pose[e]] >= 0) o->by_l[o->prob.edge_point[e]].push_back(e);
    return o;
}

void lba_oracle_shard_destroy(void* h) { delete (OracleShard*)h; }
int64_t lba_oracle_shard_reduce_len(void* h) { return (int64_t)((OracleShard*)h)->reduce.size(); }
double* lba_oracle_shard_reduce_buffer(void* h) { return ((OracleShard*)h)->reduce.data(); }

void lba_oracle_shard_linearize(void* h, double* chi2, double* max_diag_poses, double* max_diag_landmarks)
{
    OracleShard* o = (OracleShard*)h;
    o->s.compute_errors();
    *chi2 = o->s.robust_chi2();
    o->s.build_system();
    double mp = 0, ml = 0;
    for (int i = 0; i < o->s.nP; i++) for (int j = 0; j < 6; j++) mp = std::max(std::fabs(o->s.Hpp[(size_t)i * 36 + j * 7]), mp);
    for (int i = 0; i < o->s.nL; i++) for (int j = 0; j < 3; j++) ml = std::max(std::fabs(o->s.Hll[(size_t)i * 9 + j * 4]), ml);
    *max_diag_poses = mp; *max_diag_landmarks = ml;
}

void lba_oracle_shard_reduce(void* h, double lambda)
{
    OracleShard* o = (OracleShard*)h;
    Lba& s = o->s;
    const int n = o->n;
    std::fill(o->reduce.begin(), o->reduce.end(), 0.0);
    double* S = o->reduce.data();
    double* bs = S + (size_t)n * n;
    double* bpf = bs + n;
    double* dg = bpf + n;
    for (int i = 0; i < s.nP; i++)
        for (int a = 0; a < 6; a++) {
            for (int b = 0; b < 6; b++) S[(size_t)(6 * i + a) * n + 6 * i + b] = s.Hpp[(size_t)i * 36 + a * 6 + b];
            bs[6 * i + a] = s.bp[6 * i + a];
            bpf[6 * i + a] = s.bp[6 * i + a];
            dg[6 * i + a] = s.Hpp[(size_t)i * 36 + a * 7];
        }
    for (int l = 0; l < s.nL; l++) {
        double Dm[9];
        for (int k = 0; k < 9; k++) Dm[k] = s.Hll[(size_t)l * 9 + k] + (k % 4 == 0 ? lambda : 0.0);
        double* Di = &o->Dinv[(size_t)l * 9];
        inv3(Dm, Di);
        double db[3];
        for (int a = 0; a < 3; a++) db[a] = Di[a * 3] * s.bl[3 * l] + Di[a * 3 + 1] * s.bl[3 * l + 1] + Di[a * 3 + 2] * s.bl[3 * l + 2];
        const std::vector<int>& es = o->by_l[l];
        for (size_t a = 0; a < es.size(); a++) {
            const double* Bi = &s.Hpl[(size_t)es[a] * 18];
            const int i1 = s.pose_col[o->prob.edge_pose[es[a]]];
            double BD[18];
            for (int r = 0; r < 6; r++)
                for (int c = 0; c < 3; c++) BD[r * 3 + c] = Bi[r * 3] * Di[c] + Bi[r * 3 + 1] * Di[3 + c] + Bi[r * 3 + 2] * Di[6 + c];
            for (int r = 0; r < 6; r++) bs[6 * i1 + r] -= Bi[r * 3] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
            for (size_t b = 0; b < es.size(); b++) {
                const double* Bj = &s.Hpl[(size_t)es[b] * 18];
                const int i2 = s.pose_col[o->prob.edge_pose[es[b]]];
                for (int r = 0; r < 6; r++)
                    for (int c = 0; c < 6; c++)
                        S[(size_t)(6 * i1 + r) * n + 6 * i2 + c] -= BD[r * 3] * Bj[c * 3] + BD[r * 3 + 1] * Bj[c * 3 + 1] + BD[r * 3 + 2] * Bj[c * 3 + 2];
            }
        }
    }
}

int lba_oracle_shard_finish(void* h, double lambda, double* chi2_new, double* scale_poses, double* scale_landmarks)
{
    OracleShard* o = (OracleShard*)h;
    Lba& s = o->s;
    const int n = o->n;
    std::vector<double> S(o->reduce.begin(), o->reduce.begin() + (size_t)n * n);
    const double* bs = o->reduce.data() + (size_t)n * n;
    const double* bpf = bs + n;
    for (int i = 0; i < n; i++) S[(size_t)i * n + i] += lambda;
    bool ok = true;
    if (n > 0) ok = ldlt_solve(S, n, bs, s.x.data());
    double sp = 0, sl = 0;
    if (ok) {
        for (int l = 0; l < s.nL; l++) {
            double c[3] = {s.bl[3 * l], s.bl[3 * l + 1], s.bl[3 * l + 2]};
            const std::vector<int>& es = o->by_l[l];
            for (size_t a = 0; a < es.size(); a++) {
                const double* B = &s.Hpl[(size_t)es[a] * 18];
                const double* xp = &s.x[6 * (size_t)s.pose_col[o->prob.edge_pose[es[a]]]];
                for (int k = 0; k < 3; k++) { double v = 0; for (int r = 0; r < 6; r++) v += B[r * 3 + k] * xp[r]; c[k] -= v; }
            }
            const double* Di = &o->Dinv[(size_t)l * 9];
            for (int a = 0; a < 3; a++) {
                const double xl = Di[a * 3] * c[0] + Di[a * 3 + 1] * c[1] + Di[a * 3 + 2] * c[2];
                s.x[(size_t)n + 3 * l + a] = xl;
                sl += xl * (lambda * xl + s.bl[3 * l + a]);
            }
        }
        for (int j = 0; j < n; j++) sp += s.x[j] * (lambda * s.x[j] + bpf[j]);
    }
    // trial state
    std::vector<Pose> keep_poses = s.poses;
    std::vector<double> keep_pts = s.pts;
    if (ok) s.update();
    s.compute_errors();
    *chi2_new = s.robust_chi2();
    o->trial_poses = s.poses; o->trial_pts = s.pts;
    s.poses = keep_poses; s.pts = keep_pts;
    *scale_poses = sp; *scale_landmarks = sl;
    return ok ? 1 : 0;
}

void lba_oracle_shard_accept(void* h, int accept)
{
    OracleShard* o = (OracleShard*)h;
    if (accept) { o->s.poses = o->trial_poses; o->s.pts = o->trial_pts; }
}

void lba_oracle_shard_download(void* h, double* q, double* t, double* pts, double* chi2, uint8_t* depth)
{
    OracleShard* o = (OracleShard*)h;
    Lba& s = o->s;
    for (int i = 0; i < o->prob.n_poses; i++) {
        q[4 * i] = s.poses[i].q.x; q[4 * i + 1] = s.poses[i].q.y; q[4 * i + 2] = s.poses[i].q.z; q[4 * i + 3] = s.poses[i].q.w;
        for (int k = 0; k < 3; k++) t[3 * i + k] = s.poses[i].t[k];
    }
    std::memcpy(pts, s.pts.data(), sizeof(double) * 3 * (size_t)s.nL);
    for (int e = 0; e < s.nE; e++) {
        chi2[e] = s.edge_chi2(e);
        double Xc[3];
        pose_map(s.poses[o->prob.edge_pose[e]], &s.pts[3 * (size_t)o->prob.edge_point[e]], Xc);
        depth[e] = Xc[2] > 0.0;
    }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Optimizer::PoseOptimization (reference src/Optimizer.cc:814-1115): motion-only BA of one frame.  Unary edges
// EdgeSE3ProjectXYZOnlyPose (src/OptimizableTypes.cpp:49-63, include/OptimizableTypes.h:35-54) and
// g2o::EdgeStereoSE3ProjectXYZOnlyPose (types_six_dof_expmap.cpp:338-395), BaseUnaryEdge::constructQuadraticForm
// (core/base_unary_edge.hpp:44-73), LinearSolverDense on the single 6x6 block, the same Levenberg loop as above,
// 4 rounds x optimize(10) restarting from the frame pose, outlier re-classification with float chi2 (:1016-1100),
// robust kernel removed after the third round, early exit when fewer than 10 edges exist.
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct PoseOpt {
    const OraclePoseProblem* p;
    Pose T;
    std::vector<double> err;            // 3 per edge: edge._error as last computed
    std::vector<uint8_t> active;        // level 0
    bool robust;
    double dsqr_mono, dsqr_stereo;
    double H[36], b[6], x[6];

    void compute_error(int e, const Pose& P)
    {
        double Xc[3];
        pose_map(P, p->Xw + 3 * (size_t)e, Xc);
        const double* obs = p->obs + 3 * (size_t)e;
        double* r = &err[3 * (size_t)e];
        if (!p->stereo[e]) {
            r[0] = obs[0] - (p->fx * Xc[0] / Xc[2] + p->cx);
            r[1] = obs[1] - (p->fy * Xc[1] / Xc[2] + p->cy);
            r[2] = 0;
        } else {
            const float invz = 1.0f / Xc[2];
            const double u = Xc[0] * invz * p->fx + p->cx;
            const double v = Xc[1] * invz * p->fy + p->cy;
            r[0] = obs[0] - u; r[1] = obs[1] - v; r[2] = obs[2] - (u - p->bf * invz);     // double bf * float invz here
        }
    }
    double chi2(int e) const
    {
        const double* r = &err[3 * (size_t)e];
        const double w = p->inv_sigma2[e];
        double c = r[0] * (w * r[0]) + r[1] * (w * r[1]);
        if (p->stereo[e]) c += r[2] * (w * r[2]);
        return c;
    }
    void robustify(int e, double chi, double rho[2]) const
    {
        const double delta = p->stereo[e] ? p->huber_stereo : p->huber_mono;
        const double dsqr = p->stereo[e] ? dsqr_stereo : dsqr_mono;
        if (!robust || chi <= dsqr) { rho[0] = chi; rho[1] = 1.; }
        else { const double s = std::sqrt(chi); rho[0] = 2 * s * delta - dsqr; rho[1] = delta / s; }
    }
    void active_errors() { for (int e = 0; e < p->n; e++) if (active[e]) compute_error(e, T); }
    double active_robust_chi2() const
    {
        double c = 0, rho[2];
        for (int e = 0; e < p->n; e++) if (active[e]) { robustify(e, chi2(e), rho); c += rho[0]; }
        return c;
    }
    void build()
    {
        std::memset(H, 0, sizeof(H)); std::memset(b, 0, sizeof(b));
        for (int e = 0; e < p->n; e++) {
            if (!active[e]) continue;
            double Xc[3], J[18];
            pose_map(T, p->Xw + 3 * (size_t)e, Xc);
            const double xx = Xc[0], yy = Xc[1], zz = Xc[2];
            const int D = p->stereo[e] ? 3 : 2;
            if (!p->stereo[e]) {
                const double pj[6] = {-(p->fx / zz), -0.0, -(-p->fx * xx / (zz * zz)), -0.0, -(p->fy / zz), -(-p->fy * yy / (zz * zz))};
                const double S[18] = {0, zz, -yy, 1, 0, 0, -zz, 0, xx, 0, 1, 0, yy, -xx, 0, 0, 0, 1};
                for (int r = 0; r < 2; r++)
                    for (int c = 0; c < 6; c++) J[r * 6 + c] = pj[r * 3] * S[c] + pj[r * 3 + 1] * S[6 + c] + pj[r * 3 + 2] * S[12 + c];
            } else {
                const double invz = 1.0 / zz, invz_2 = invz * invz, fx = p->fx, fy = p->fy, bf = p->bf;
                J[0] = xx * yy * invz_2 * fx; J[1] = -(1 + (xx * xx * invz_2)) * fx; J[2] = yy * invz * fx; J[3] = -invz * fx; J[4] = 0; J[5] = xx * invz_2 * fx;
                J[6] = (1 + yy * yy * invz_2) * fy; J[7] = -xx * yy * invz_2 * fy; J[8] = -xx * invz * fy; J[9] = 0; J[10] = -invz * fy; J[11] = yy * invz_2 * fy;
                J[12] = J[0] - bf * yy * invz_2; J[13] = J[1] + bf * xx * invz_2; J[14] = J[2]; J[15] = J[3]; J[16] = 0; J[17] = J[5] - bf * invz_2;
            }
            const double w = p->inv_sigma2[e];
            const double* r = &err[3 * (size_t)e];
            double rho[2];
            robustify(e, chi2(e), rho);
            for (int a = 0; a < 6; a++) {
                double s = 0;
                for (int d = 0; d < D; d++) s += J[d * 6 + a] * (w * r[d]);
                b[a] -= rho[1] * s;
                for (int c = 0; c < 6; c++) {
                    double h = 0;
                    for (int d = 0; d < D; d++) h += J[d * 6 + a] * (rho[1] * w) * J[d * 6 + c];
                    H[a * 6 + c] += h;
                }
            }
        }
    }
    int iters_done = 0, trials_done = 0;
    double last_chi = 0;
    // g2o optimize(iters) with Levenberg on the single pose vertex
    void optimize(int iters)
    {
        iters_done = 0; trials_done = 0; last_chi = 0;
        int n_active = 0;
        for (int e = 0; e < p->n; e++) n_active += active[e];
        if (n_active == 0) return;
        double lambda = -1, ni = 2;
        int nBad = 0;
        for (int it = 0; it < iters; it++) {
            active_errors();
            double currentChi = active_robust_chi2(), tempChi = currentChi;
            const double iniChi = currentChi;
            build();
            if (it == 0) {
                double m = 0;
                for (int j = 0; j < 6; j++) m = std::max(std::fabs(H[j * 7]), m);
                lambda = 1e-5 * m; ni = 2; nBad = 0;
            }
            double rho = 0;
            int qmax = 0;
            do {
                const Pose backup = T;
                std::vector<double> A(H, H + 36);
                for (int j = 0; j < 6; j++) A[j * 7] += lambda;
                const bool ok2 = ldlt_solve(A, 6, b, x);
                if (ok2) T = pose_mul(se3_exp(x), T);
                active_errors();
                tempChi = active_robust_chi2();
                if (!ok2) tempChi = std::numeric_limits<double>::max();
                rho = currentChi - tempChi;
                double scale = 0;
                for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && std::isfinite(tempChi)) {
                    double alpha = 1. - std::pow((2 * rho - 1), 3);
                    alpha = std::min(alpha, 2. / 3.);
                    lambda *= std::max(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                } else {
                    lambda *= ni; ni *= 2;
                    T = backup;
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            iters_done++; trials_done += qmax; last_chi = currentChi;
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
            if (nBad >= 3) break;
        }
    }
};

}  // namespace

extern "C" int pose_oracle_optimize(const OraclePoseProblem* p, double* q_out, double* t_out, uint8_t* outlier_out, int* n_bad_out,
                                    OraclePoseStats* stats)
{
    if (stats) std::memset(stats, 0, sizeof(*stats));
    PoseOpt s;
    s.p = p;
    s.err.assign(3 * (size_t)std::max(p->n, 1), 0.0);
    s.active.assign(std::max(p->n, 1), 1);
    s.robust = true;
    s.dsqr_mono = p->huber_mono * p->huber_mono; s.dsqr_stereo = p->huber_stereo * p->huber_stereo;
    Pose T0;
    T0.q.x = p->q[0]; T0.q.y = p->q[1]; T0.q.z = p->q[2]; T0.q.w = p->q[3];
    for (int k = 0; k < 3; k++) T0.t[k] = p->t[k];
    quat_normalize(T0.q);
    s.T = T0;
    std::vector<uint8_t> outlier(std::max(p->n, 1), 0);
    int nBad = 0;
    const int nInitial = p->n;
    if (nInitial >= 3) {
        const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
        for (int it = 0; it < 4; it++) {
            s.T = T0;                               // vSE3->setEstimate(pFrame->GetPose()) every round (:1007-1008)
            s.optimize(10);
            if (stats) { stats->iterations[it] = s.iters_done; stats->trials[it] = s.trials_done; stats->chi2[it] = s.last_chi; }
            nBad = 0;
            for (int e = 0; e < p->n; e++) {
                if (outlier[e]) s.compute_error(e, s.T);
                const float chi2 = (float)s.chi2(e);
                if (chi2 > (p->stereo[e] ? chi2Stereo : chi2Mono)) { outlier[e] = 1; s.active[e] = 0; nBad++; }
                else { outlier[e] = 0; s.active[e] = 1; }
            }
            if (it == 2) s.robust = false;          // setRobustKernel(0) for the last round
            if (p->n < 10) break;
        }
    }
    q_out[0] = s.T.q.x; q_out[1] = s.T.q.y; q_out[2] = s.T.q.z; q_out[3] = s.T.q.w;
    for (int k = 0; k < 3; k++) t_out[k] = s.T.t[k];
    if (outlier_out) std::memcpy(outlier_out, outlier.data(), p->n);
    if (n_bad_out) *n_bad_out = nBad;
    return nInitial < 3 ? 0 : nInitial - nBad;
}
