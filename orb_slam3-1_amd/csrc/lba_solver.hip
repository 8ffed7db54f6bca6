// lba_solver.hip -- gfx950 kernels + C ABI for the numerical core of Optimizer::LocalBundleAdjustment
// (reference src/Optimizer.cc:1116-1498) = g2o Levenberg-Marquardt (optimization_algorithm_levenberg.cpp:61-194)
// over EdgeSE3ProjectXYZ / EdgeStereoSE3ProjectXYZ edges with Huber kernels and BlockSolver_6_3 with Schur
// complement (block_solver.hpp:354-604).  All arithmetic is f64 like g2o.
//
// Device data (SoA, HBM): poses [P][7] (qx qy qz qw tx ty tz), points [L][3], edges in caller order, CSR of the
// edges of every landmark and of every non-fixed pose, and a CSR "pair list": for every non-zero 6x6 block (i<=j)
// of the reduced camera system, the (edge_a, edge_b) pairs of landmarks seen by both poses.
//
// One LM trial:
//   k_schur_landmarks  D^-1 = (Hll + lambda I)^-1, db = D^-1 b_l, Z_e = W_e D^-1          (1 thread / landmark)
//   k_schur_blocks     S_ij = [Hpp_ii] - sum_pairs Z_a W_b^T ,  b_s = b_p - sum W_e db        (1 wave / block)
//   (multi-GPU: the caller all-reduces [S | b_s | b_p | diag Hpp] here -- RCCL over xGMI)
//   k_add_lambda, blocked Cholesky (k_chol_panel / k_chol_update per 60-column step), k_chol_solve
//   k_chol_solve_update  substitution, then trial poses = oplus(poses, x_p) and the pose part of the scale sum
//   k_update_errors    x_l = D^-1 (b_l - W^T x_p), trial points, residuals + Huber rho of the trial state per landmark; the last
//                      workgroup sums chi2 / scale in a fixed order
// Every reduction is ordered (CSR gather or fixed tree), so results are reproducible run to run.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbslam3_hip.h"
#include "se3_device.h"

namespace orbx {
int fail(int code, const char* fmt, ...);
}
using orbx::fail;

#define LBA_HIP(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(ORBX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace lba {

constexpr int NB = 60;      // Cholesky block size (10 poses)

struct Cam { double fx, fy, cx, cy, bf, huber_mono, huber_stereo, dsqr_mono, dsqr_stereo; };

struct Dev {        // device pointers of one problem (passed by value to kernels)
    int nPoses, nP, nL, nE, n;      // n = 6 nP
    const int* pose_col;            // [nPoses] column among non-fixed poses or -1
    const int* col_pose;            // [nP] pose index of column
    const int* e_point; const int* e_pose; const double* e_obs; const double* e_w; const uint8_t* e_stereo;
    const int* l_off; const int* l_edge;        // edges of a landmark (caller order)
    const int* p_off; const int* p_edge;        // edges of a non-fixed pose (by column)
    const int* b_i; const int* b_j; const int* b_off; const int2* b_pair; int nBlocks;
    double* Hll; double* bl; double* Hpp; double* bp; double* W; double* Z; double* Dinv; double* db;
    double* err; double* rho0;      // [nE][3], [nE]
    double* x;                      // [n + 3 nL]
    double* part;                   // scale partials [nL + nP]
    double* chi_part;               // robust chi2 of a landmark's edges [nL] (k_update_errors)
    unsigned int* ticket;           // workgroups of k_update_errors that are done (the last one reduces)
    double* scal;                   // [16] scalars: 0 chi2, 1 max diag (poses), 2 max diag (landmarks), 3 scale poses, 4 scale landmarks, 5 chol fail flag
    Cam cam;
};

using namespace se3;

// residual of one edge (EdgeSE3ProjectXYZ::computeError / EdgeStereoSE3ProjectXYZ::computeError)
__device__ __forceinline__ void edge_residual(const Cam& c, const double* Xc, const double* obs, int stereo, double* r)
{
    if (!stereo) {
        r[0] = obs[0] - (c.fx * Xc[0] / Xc[2] + c.cx);
        r[1] = obs[1] - (c.fy * Xc[1] / Xc[2] + c.cy);
        r[2] = 0;
    } else {
        const float invz = (float)(1.0 / Xc[2]);         // 1.0f/double rounded to float (types_six_dof_expmap.cpp:191)
        const double u = Xc[0] * (double)invz * c.fx + c.cx;
        const double v = Xc[1] * (double)invz * c.fy + c.cy;
        const double ur = u - (double)((float)c.bf * invz);
        r[0] = obs[0] - u; r[1] = obs[1] - v; r[2] = obs[2] - ur;
    }
}

__device__ __forceinline__ void huber(const Cam& c, int stereo, double chi, double& rho0, double& rho1)
{
    const double delta = stereo ? c.huber_stereo : c.huber_mono;
    const double dsqr = stereo ? c.dsqr_stereo : c.dsqr_mono;
    if (delta <= 0 || chi <= dsqr) { rho0 = chi; rho1 = 1.0; }
    else { const double s = sqrt(chi); rho0 = 2 * s * delta - dsqr; rho1 = delta / s; }
}

// The per-edge 6 x 3 blocks W_e / Z_e.  A lane that walks its own block with 8-byte accesses makes 18 memory transactions per
// block and wave-instruction slot, and the memory system is bound by the NUMBER of such transactions, not by their bytes
// (k_schur_landmarks_b: 129 memory instructions per wave at ~3 000 cycles each with 32 windows per launch) -- so a block is stored
// as two 16-byte aligned halves of 9 doubles + 1 of padding (rows 0-2 | rows 3-5: 160 bytes) and moves as 5 + 5 16-byte accesses;
// the Schur kernel, whose lanes each take one half of a Z block and one half of a W block, reads a half in 5 accesses without
// any alignment case.
constexpr int kBlk = 20;        // doubles per stored block
__device__ __forceinline__ size_t lba_blk(int e) { return (size_t)kBlk * (size_t)e; }
__device__ __forceinline__ void blk_load_half(const double* __restrict__ blk, int half, double* v)       // rows 3 half .. 3 half + 2
{
    const double2* q = (const double2*)(blk + 10 * half);
#pragma unroll
    for (int i = 0; i < 4; i++) { const double2 t = q[i]; v[2 * i] = t.x; v[2 * i + 1] = t.y; }
    v[8] = q[4].x;
}
__device__ __forceinline__ void blk_load(const double* __restrict__ blk, double* v)
{
    blk_load_half(blk, 0, v);
    blk_load_half(blk, 1, v + 9);
}
__device__ __forceinline__ void blk_store(double* __restrict__ blk, const double* v)
{
    double2* q = (double2*)blk;
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int i = 0; i < 4; i++) q[5 * h + i] = make_double2(v[9 * h + 2 * i], v[9 * h + 2 * i + 1]);
        q[5 * h + 4] = make_double2(v[9 * h + 8], 0.0);
    }
}

// Jacobians of one edge: Ji (D x 3, point) and Jj (D x 6, pose), rows padded to 3
__device__ inline void edge_jacobians(const Cam& c, const double* T, const double* Xc, int stereo, double* Ji, double* Jj)
{
    double R[9];
    quat_to_R(T, R);
    const double x = Xc[0], y = Xc[1], z = Xc[2];
    if (!stereo) {
        const double p00 = -(c.fx / z), p02 = c.fx * x / (z * z), p11 = -(c.fy / z), p12 = c.fy * y / (z * z);    // -projectJac
        for (int k = 0; k < 3; k++) {
            Ji[k] = p00 * R[k] + p02 * R[6 + k];
            Ji[3 + k] = p11 * R[3 + k] + p12 * R[6 + k];
            Ji[6 + k] = 0;
        }
        // SE3deriv rows: [0 z -y 1 0 0; -z 0 x 0 1 0; y -x 0 0 0 1]
        Jj[0] = p02 * y;            Jj[1] = p00 * z + p02 * (-x); Jj[2] = p00 * (-y);
        Jj[3] = p00;                Jj[4] = 0;                    Jj[5] = p02;
        Jj[6] = p11 * (-z) + p12 * y; Jj[7] = p12 * (-x);         Jj[8] = p11 * x;
        Jj[9] = 0;                  Jj[10] = p11;                 Jj[11] = p12;
        for (int k = 12; k < 18; k++) Jj[k] = 0;
    } else {
        const double z2 = z * z, fx = c.fx, fy = c.fy, bf = c.bf;
        Ji[0] = -fx * R[0] / z + fx * x * R[6] / z2; Ji[1] = -fx * R[1] / z + fx * x * R[7] / z2; Ji[2] = -fx * R[2] / z + fx * x * R[8] / z2;
        Ji[3] = -fy * R[3] / z + fy * y * R[6] / z2; Ji[4] = -fy * R[4] / z + fy * y * R[7] / z2; Ji[5] = -fy * R[5] / z + fy * y * R[8] / z2;
        Ji[6] = Ji[0] - bf * R[6] / z2; Ji[7] = Ji[1] - bf * R[7] / z2; Ji[8] = Ji[2] - bf * R[8] / z2;
        Jj[0] = x * y / z2 * fx;  Jj[1] = -(1 + (x * x / z2)) * fx; Jj[2] = y / z * fx;  Jj[3] = -1. / z * fx; Jj[4] = 0; Jj[5] = x / z2 * fx;
        Jj[6] = (1 + y * y / z2) * fy; Jj[7] = -x * y / z2 * fy; Jj[8] = -x / z * fy; Jj[9] = 0; Jj[10] = -1. / z * fy; Jj[11] = y / z2 * fy;
        Jj[12] = Jj[0] - bf * y / z2; Jj[13] = Jj[1] + bf * x / z2; Jj[14] = Jj[2]; Jj[15] = Jj[3]; Jj[16] = 0; Jj[17] = Jj[5] - bf / z2;
    }
}

// ---- errors of a state (SparseOptimizer::computeActiveErrors + per-edge robust chi2) ----
__device__ __forceinline__ void errors_body(Dev d, const double* __restrict__ poses, const double* __restrict__ pts, const int bx)
{
    const int e = bx * 256 + threadIdx.x;
    if (e >= d.nE) return;
    double Xc[3], r[3];
    pose_map(poses + 7 * (size_t)d.e_pose[e], pts + 3 * (size_t)d.e_point[e], Xc);
    const int st = d.e_stereo[e];
    edge_residual(d.cam, Xc, d.e_obs + 3 * (size_t)e, st, r);
    const double w = d.e_w[e];
    double chi = r[0] * (w * r[0]) + r[1] * (w * r[1]);
    if (st) chi += r[2] * (w * r[2]);
    double rho0, rho1;
    huber(d.cam, st, chi, rho0, rho1);
    d.err[3 * (size_t)e] = r[0]; d.err[3 * (size_t)e + 1] = r[1]; d.err[3 * (size_t)e + 2] = r[2];
    d.rho0[e] = rho0;
}
__global__ __launch_bounds__(256) void k_errors(Dev d, const double* __restrict__ poses, const double* __restrict__ pts)
{
    errors_body(d, poses, pts, (int)blockIdx.x);
}

// ---- buildSystem, landmark side: Hll, bl and the Hpl blocks W_e = B^T (rho1 Omega) A (6x3) ----
// 8 lanes per landmark (a landmark has ~10 edges): lane q takes edges q, q+8, ...; fixed butterfly reduction.
// Schur, landmark side, for one landmark held by 8 lanes: D^-1 of A = Hll + lambda I, db = D^-1 bl, Z_e = W_e D^-1 of its edges
__device__ __forceinline__ void schur_landmark(const Dev& d, int l, int sub, const double* A, double b0, double b1, double b2)
{
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
    const double id = 1.0 / det;
    double Di[9];
    Di[0] = c00 * id; Di[1] = (A[2] * A[7] - A[1] * A[8]) * id; Di[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Di[3] = c01 * id; Di[4] = (A[0] * A[8] - A[2] * A[6]) * id; Di[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Di[6] = c02 * id; Di[7] = (A[1] * A[6] - A[0] * A[7]) * id; Di[8] = (A[0] * A[4] - A[1] * A[3]) * id;
    if (sub == 0) {
        for (int k = 0; k < 9; k++) d.Dinv[9 * (size_t)l + k] = Di[k];
        for (int a = 0; a < 3; a++) d.db[3 * (size_t)l + a] = Di[a * 3] * b0 + Di[a * 3 + 1] * b1 + Di[a * 3 + 2] * b2;
    }
    for (int k = d.l_off[l] + sub; k < d.l_off[l + 1]; k += 8) {
        const int e = d.l_edge[k];
        if (d.pose_col[d.e_pose[e]] < 0) continue;
        double W[18], Z[18];
        blk_load(d.W + lba_blk(e), W);
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) Z[r * 3 + c] = W[r * 3] * Di[c] + W[r * 3 + 1] * Di[3 + c] + W[r * 3 + 2] * Di[6 + c];
        blk_store(d.Z + lba_blk(e), Z);
    }
}

// (device function: the merged launch k_lin_all runs it in the workgroups behind the pose ones; lambda >= 0 also performs the landmark side of
// the Schur complement for that lambda -- the W_e of a landmark's edges are written and read back by the same lanes)
__device__ __forceinline__ void lin_landmarks_body(const Dev& d, const double* __restrict__ poses, const double* __restrict__ pts, int l, double lambda)
{
    const int sub = threadIdx.x & 7;
    const bool live = l < d.nL;
    double acc[9];      // Hll upper triangle (6) + bl (3)
    for (int k = 0; k < 9; k++) acc[k] = 0;
    if (live) {
        const double* X = pts + 3 * (size_t)l;
        for (int k = d.l_off[l] + sub; k < d.l_off[l + 1]; k += 8) {
            const int e = d.l_edge[k];
            const int ip = d.e_pose[e];
            const double* T = poses + 7 * (size_t)ip;
            const int st = d.e_stereo[e];
            double Xc[3], Ji[9], Jj[18];
            pose_map(T, X, Xc);
            edge_jacobians(d.cam, T, Xc, st, Ji, Jj);
            const double* r = d.err + 3 * (size_t)e;
            const double w = d.e_w[e];
            double chi = r[0] * (w * r[0]) + r[1] * (w * r[1]);
            if (st) chi += r[2] * (w * r[2]);
            double rho0, rho1;
            huber(d.cam, st, chi, rho0, rho1);
            const double wr = rho1 * w;
            double orr[3];
            for (int q = 0; q < 3; q++) orr[q] = (-(w * r[q])) * rho1;
            // fixed trip counts (row 2 of a mono edge is zero, so its terms add +0) keep Ji/Jj in registers
#pragma unroll
            for (int a = 0; a < 3; a++) {
#pragma unroll
                for (int c = a; c < 3; c++) {
                    double h = 0;
#pragma unroll
                    for (int q = 0; q < 3; q++) h += Ji[q * 3 + a] * wr * Ji[q * 3 + c];
                    acc[a * 3 - (a * (a - 1)) / 2 + (c - a)] += h;
                }
                double sv = 0;
#pragma unroll
                for (int q = 0; q < 3; q++) sv += Ji[q * 3 + a] * orr[q];
                acc[6 + a] += sv;
            }
            if (d.pose_col[ip] >= 0) {
                double W[18];
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        double h = 0;
#pragma unroll
                        for (int q = 0; q < 3; q++) h += Jj[q * 6 + a] * wr * Ji[q * 3 + c];
                        W[a * 3 + c] = h;
                    }
                blk_store(d.W + lba_blk(e), W);
            }
        }
    }
    for (int k = 0; k < 9; k++)
        for (int o = 4; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
    if (live && sub == 0) {
        double* H = d.Hll + 9 * (size_t)l;
        H[0] = acc[0]; H[1] = acc[1]; H[2] = acc[2];
        H[3] = acc[1]; H[4] = acc[3]; H[5] = acc[4];
        H[6] = acc[2]; H[7] = acc[4]; H[8] = acc[5];
        d.bl[3 * (size_t)l] = acc[6]; d.bl[3 * (size_t)l + 1] = acc[7]; d.bl[3 * (size_t)l + 2] = acc[8];
    }
    if (live && lambda >= 0.0) {
        const double A[9] = {acc[0] + lambda, acc[1], acc[2], acc[1], acc[3] + lambda, acc[4], acc[2], acc[4], acc[5] + lambda};
        schur_landmark(d, l, sub, A, acc[6], acc[7], acc[8]);
    }
}

// ---- buildSystem, pose side: Hpp (6x6) and bp; one 256-thread workgroup per non-fixed pose, fixed reduction tree ----
__device__ __forceinline__ void lin_all_body(Dev d, const double* __restrict__ poses, const double* __restrict__ pts, double lambda, const int bx)
{
    __shared__ double s_part[4][27];
    if ((int)bx >= d.nP) {          // landmark workgroups: 32 landmarks x 8 lanes
        lin_landmarks_body(d, poses, pts, ((int)bx - d.nP) * 32 + (threadIdx.x >> 3), lambda);
        return;
    }
    const int col = bx, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ip = d.col_pose[col];
    const double* T = poses + 7 * (size_t)ip;
    double acc[27];      // 21 upper-triangular entries of Hpp + 6 of bp
    for (int k = 0; k < 27; k++) acc[k] = 0;
    for (int k = d.p_off[col] + tid; k < d.p_off[col + 1]; k += 256) {
        const int e = d.p_edge[k];
        const int st = d.e_stereo[e];
        double Xc[3], Ji[9], Jj[18];
        pose_map(T, pts + 3 * (size_t)d.e_point[e], Xc);
        edge_jacobians(d.cam, T, Xc, st, Ji, Jj);
        const double* r = d.err + 3 * (size_t)e;
        const double w = d.e_w[e];
        double chi = r[0] * (w * r[0]) + r[1] * (w * r[1]);
        if (st) chi += r[2] * (w * r[2]);
        double rho0, rho1;
        huber(d.cam, st, chi, rho0, rho1);
        const double wr = rho1 * w;
        const double orr[3] = {(-(w * r[0])) * rho1, (-(w * r[1])) * rho1, (-(w * r[2])) * rho1};
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int c = a; c < 6; c++) {
                double h = 0;
#pragma unroll
                for (int q = 0; q < 3; q++) h += Jj[q * 6 + a] * wr * Jj[q * 6 + c];
                acc[a * 6 - (a * (a - 1)) / 2 + (c - a)] += h;
            }
            double sv = 0;
#pragma unroll
            for (int q = 0; q < 3; q++) sv += Jj[q * 6 + a] * orr[q];
            acc[21 + a] += sv;
        }
    }
    for (int k = 0; k < 27; k++)
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
    if (lane == 0) for (int k = 0; k < 27; k++) s_part[wave][k] = acc[k];
    __syncthreads();
    if (tid < 27) {
        const double v = ((s_part[0][tid] + s_part[1][tid]) + s_part[2][tid]) + s_part[3][tid];
        if (tid < 21) {
            int a = 0, rem = tid;
            while (rem >= 6 - a) { rem -= 6 - a; a++; }
            const int c = a + rem;
            double* H = d.Hpp + 36 * (size_t)col;
            H[a * 6 + c] = v; H[c * 6 + a] = v;
        } else {
            d.bp[6 * (size_t)col + (tid - 21)] = v;
        }
    }
}
__global__ __launch_bounds__(256) void k_lin_all(Dev d, const double* __restrict__ poses, const double* __restrict__ pts, double lambda)
{
    lin_all_body(d, poses, pts, lambda, (int)blockIdx.x);
}

// ---- deterministic scalar reductions (single workgroup) ----
// mode 0: chi2 = sum rho0, max diagonals.  mode 1: chi2 = sum rho0, scale = sum partials.
// The results also go to a host-mapped, coherent buffer followed by a sequence number (system-scope release), so that the
// host reads them by polling that word instead of a device-to-host copy plus a stream synchronisation per LM trial.
__device__ __forceinline__ void reduce_body(Dev d, int mode, double* __restrict__ hmap, unsigned long long seq, const int bx)
{
    // fixed tree: a strided partial per thread, a butterfly inside each wave, the 16 wave results by wave 0 -- one barrier instead
    // of ten (the kernel is a single workgroup on the critical path of every Levenberg trial)
    __shared__ double s_a[16], s_b[16], s_c[16];
    const int tid = threadIdx.x;
    double a = 0, b = 0, c = 0;
    for (int e = tid; e < d.nE; e += 1024) a += d.rho0[e];
    if (mode == 0) {
        for (int i = tid; i < d.nP * 6; i += 1024) b = fmax(b, fabs(d.Hpp[36 * (size_t)(i / 6) + (i % 6) * 7]));
        for (int i = tid; i < d.nL * 3; i += 1024) c = fmax(c, fabs(d.Hll[9 * (size_t)(i / 3) + (i % 3) * 4]));
    } else {
        for (int i = tid; i < d.nP; i += 1024) b += d.part[d.nL + i];
        for (int i = tid; i < d.nL; i += 1024) c += d.part[i];
    }
    auto combine = [&](int o, int width) {
        const double a2 = __shfl_xor(a, o, width), b2 = __shfl_xor(b, o, width), c2 = __shfl_xor(c, o, width);
        a += a2;
        if (mode == 0) { b = fmax(b, b2); c = fmax(c, c2); } else { b += b2; c += c2; }
    };
    for (int o = 32; o > 0; o >>= 1) combine(o, 64);
    if ((tid & 63) == 0) { s_a[tid >> 6] = a; s_b[tid >> 6] = b; s_c[tid >> 6] = c; }
    __syncthreads();
    if (tid < 64) {
        a = tid < 16 ? s_a[tid] : 0.0; b = tid < 16 ? s_b[tid] : 0.0; c = tid < 16 ? s_c[tid] : 0.0;
        for (int o = 8; o > 0; o >>= 1) combine(o, 16);
        if (tid == 0) { s_a[0] = a; s_b[0] = b; s_c[0] = c; }
    }
    if (tid == 0) {
        d.scal[0] = s_a[0];
        if (mode == 0) { d.scal[1] = s_b[0]; d.scal[2] = s_c[0]; }
        else { d.scal[3] = s_b[0]; d.scal[4] = s_c[0]; }
        if (hmap) {
            hmap[0] = s_a[0];
            if (mode == 0) { hmap[1] = s_b[0]; hmap[2] = s_c[0]; }
            else { hmap[3] = s_b[0]; hmap[4] = s_c[0]; hmap[5] = d.scal[5]; }
            __threadfence_system();
            __hip_atomic_store((unsigned long long*)(hmap + 8), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ __launch_bounds__(1024) void k_reduce(Dev d, int mode, double* __restrict__ hmap, unsigned long long seq)
{
    reduce_body(d, mode, hmap, seq, (int)blockIdx.x);
}

// ---- Schur, landmark side (block_solver.hpp:381-395): Dinv, db, Z_e = W_e Dinv; 8 lanes per landmark ----
__device__ __forceinline__ void schur_landmarks_body(Dev d, double lambda, const int bx)
{
    const int l = bx * 8 + (threadIdx.x >> 3);
    const int sub = threadIdx.x & 7;
    if (l >= d.nL) return;
    double A[9];
    for (int k = 0; k < 9; k++) A[k] = d.Hll[9 * (size_t)l + k] + ((k % 4 == 0) ? lambda : 0.0);
    const double* b = d.bl + 3 * (size_t)l;
    schur_landmark(d, l, sub, A, b[0], b[1], b[2]);
}
__global__ __launch_bounds__(64) void k_schur_landmarks(Dev d, double lambda)
{
    schur_landmarks_body(d, lambda, (int)blockIdx.x);
}

// ---- Schur, pose side: one workgroup per 6x6 block (i<=j) of the reduced camera system ----
// 7 groups of 36 lanes split the block's (edge_a, edge_b) pair list; partials are combined in a fixed order.
// Every block of the upper triangle has an entry (empty pair list for poses that share no landmark), so S is fully
// overwritten and needs no memset.  lambda_diag is added to the diagonal here in the single-GPU path (0 when the caller
// all-reduces partial systems first and adds lambda afterwards).
// Workgroups [nBlocks, nBlocks + nP): b_schur = b_p - sum_e W_e db_l(e) (block_solver.hpp:413,436-439), plus copies of
// b_p and diag(Hpp) for the reduce buffer (additive over shards).
constexpr int kSchurThreads = 256, kSchurGroups = 64;      // 64 groups of 4 lanes walk a block's pair list, a lane owns a 3 x 3 corner
                                                            // of the 6 x 6 block.  (History: 7 groups x 36 lanes with one entry per lane:
                                                            // the diagonal blocks' chains of dependent loads were the kernel's time; 28 x 36:
                                                            // 22 us per window, but every lane loaded 6 doubles for 3 FMAs -- with 32 windows
                                                            // per launch the kernel moved 7 TB/s out of the caches and took 60 % of a round.
                                                            // A 3 x 3 corner loads 18 doubles for 27 FMAs.)
__device__ __forceinline__ void schur_blocks_body(Dev d, double* __restrict__ S, double lambda_diag,
                                                      double* __restrict__ bs, double* __restrict__ bp_out, double* __restrict__ diag_out, const int bx)
{
    __shared__ double s_part[kSchurGroups][37];
    __shared__ double s_seg[4][36];
    const int blk = bx, tid = threadIdx.x;
    if (blk >= d.nBlocks) {
        const int i = blk - d.nBlocks, lane = tid & 63, wave = tid >> 6;
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int k = d.p_off[i] + tid; k < d.p_off[i + 1]; k += kSchurThreads) {
            const int e = d.p_edge[k];
            double W[18];
            blk_load(d.W + lba_blk(e), W);
            const double* db = d.db + 3 * (size_t)d.e_point[e];
            const double b0 = db[0], b1 = db[1], b2 = db[2];
#pragma unroll
            for (int r = 0; r < 6; r++) acc[r] += W[r * 3] * b0 + W[r * 3 + 1] * b1 + W[r * 3 + 2] * b2;
        }
        for (int r = 0; r < 6; r++)
            for (int o = 32; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o);
        if (lane == 0) for (int r = 0; r < 6; r++) s_part[wave][r] = acc[r];
        __syncthreads();
        if (tid < 6) {
            double a = 0;
            for (int w = 0; w < kSchurThreads / 64; w++) a += s_part[w][tid];
            bs[6 * i + tid] = d.bp[6 * (size_t)i + tid] - a;
            bp_out[6 * i + tid] = d.bp[6 * (size_t)i + tid];
            diag_out[6 * i + tid] = d.Hpp[36 * (size_t)i + tid * 7];
        }
        return;
    }
    const int i = d.b_i[blk], j = d.b_j[blk];
    const int n = d.n;
    const int g = tid >> 2, q = tid & 3, rb = q >> 1, cb = q & 1;
    {
        double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int k = d.b_off[blk] + g; k < d.b_off[blk + 1]; k += kSchurGroups) {
            const int2 pr = d.b_pair[k];
            double z[9], w[9];
            blk_load_half(d.Z + lba_blk(pr.x), rb, z);
            blk_load_half(d.W + lba_blk(pr.y), cb, w);
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) acc[a][b] += z[3 * a] * w[3 * b] + z[3 * a + 1] * w[3 * b + 1] + z[3 * a + 2] * w[3 * b + 2];
        }
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) s_part[g][(3 * rb + a) * 6 + 3 * cb + b] = acc[a][b];
    }
    __syncthreads();
    // fixed order: four segments of 16 groups, then the four segment sums
    if (tid < 144) {
        const int ent = tid % 36, seg = tid / 36;
        double sum = s_part[16 * seg][ent];
#pragma unroll
        for (int u = 1; u < 16; u++) sum += s_part[16 * seg + u][ent];
        s_seg[seg][ent] = sum;
    }
    __syncthreads();
    if (tid < 36) {
        const int r = tid / 6, c = tid - r * 6;
        const double sum = ((s_seg[0][tid] + s_seg[1][tid]) + s_seg[2][tid]) + s_seg[3][tid];
        double v = ((i == j) ? d.Hpp[36 * (size_t)i + tid] : 0.0) - sum;
        if (i == j && r == c) v += lambda_diag;
        S[(size_t)(6 * i + r) * n + 6 * j + c] = v;
        if (i != j) S[(size_t)(6 * j + c) * n + 6 * i + r] = v;
    }
}
__global__ __launch_bounds__(kSchurThreads) void k_schur_blocks(Dev d, double* __restrict__ S, double lambda_diag,
                                                      double* __restrict__ bs, double* __restrict__ bp_out, double* __restrict__ diag_out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) d.scal[5] = 0.0;        // the factorisation's failure flag of this trial (it was a memset per trial)
    schur_blocks_body(d, S, lambda_diag, bs, bp_out, diag_out, (int)blockIdx.x);
}

__global__ void k_add_lambda(double* S, int n, double lambda)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) S[(size_t)i * n + i] += lambda;
}

// ---- blocked right-looking Cholesky of the (dense, small) reduced camera system, lower triangle ----
// Per 60-column step: k_chol_diag factors the diagonal block AND inverts it (Gauss-Jordan on [L | I]) in LDS with
// O(1)-depth steps; k_chol_panel then gets the rows below as a dense product X = A * Linv^T (no substitution chains);
// k_chol_update applies the trailing update.  The substitutions in k_chol_solve also only need Linv.
// Factor AND invert one diagonal block held in registers.  Thread (ty, tx) of a 16 x 16 grid owns the elements
// (ty + 16a, tx + 16b), a, b < 4, of the 64 x 64-padded block L (identity beyond nb) and of X = L^-1.
// (History: one column per barrier 33.7 us per 60-column block, two columns 27.5 us, four columns 19.1 us.)
// FOUR columns per barrier.  The owners publish the raw columns j0..j0+3 of L and rows j0..j0+3 of X; every thread factors the
// 4 x 4 pivot block P = Lp Lp^T itself and forms M = Lp^-1 (replicated: no broadcast), then
//   U = A[:, j0..j0+3] M^T  (its rows / its columns),   Xn = M X[j0..j0+3][:],
//   rows below the pivot block:  L -= U U^T,  X -= U Xn;   rows of the pivot block: X <- Xn.
// L itself is not an output (only X = L^-1 is), so finished columns are never written back, and garbage above the diagonal of
// the diagonal 16 x 16 tiles is never read (columns are consumed from their diagonal element downwards).
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
struct CholVec4 { double col[2][4][64], row[2][4][64]; };
__device__ __forceinline__ double rsqrt_newton(double d)
{
    double inv = __builtin_amdgcn_rsq(d);
    inv = inv * fma(-0.5 * d * inv, inv, 1.5);
    return inv * fma(-0.5 * d * inv, inv, 1.5);
}
// The rank-4 updates of the step run on the f64 matrix pipe.  A 256-thread workgroup is one wave per SIMD,
// where a v_fma_f64 issues every ~8.5 clocks and v_mfma_f64_16x16x4 (2048 FLOP) every 64 (tools/probes/f64_rates.hip): the
// 128 FMAs per thread of the register-tile update become at most 5 MFMAs, and a lane only prepares the operands the MFMA takes
// from it (one value of U, one of V or Xn per column block: 4 FMAs each on M's row k = lane >> 4) instead of the 48 values
// its 4 x 4 tile would need.  Thread (ty, tx) = lane (ty & 3) * 16 + tx of wave ty >> 2 owns rows ty + 16 a: exactly the rows
// of accumulator component a when the wave feeds the MFMA rows m -> 4 w + (m & 3) + 16 (m >> 2) (as k_chol_step does), so
// Lacc[b][a] / Xacc[b][a] ARE the thread's elements (ty + 16 a, tx + 16 b).
#ifdef LBA_STEP_TIMING       // cycle split of the 4-column groups of chol_tile_mfma (thread 0 of the factoring workgroup)
__device__ unsigned long long d_tile_prof[8];
#define LBA_TTICK(k) if (threadIdx.x == 0) { const long long t_now = clock64(); d_tile_prof[k] += (unsigned long long)(t_now - t_tile); t_tile = t_now; }
#else
#define LBA_TTICK(k)
#endif
__device__ __forceinline__ bool chol_tile_mfma(double (&Lr)[4][4], int nb, double* __restrict__ Li, CholVec4& sv)
{
#ifdef LBA_STEP_TIMING
    long long t_tile = clock64();
#endif
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int lk = ty & 3, wv4 = ty & ~3;                   // MFMA k index of this lane; first row of the wave's row group
    const int r_u = wv4 + (tx & 3) + 16 * (tx >> 2);        // the row whose U value this lane feeds (MFMA row m = tx)
    mfma_d4 Lacc[4], Xacc[4];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int a = 0; a < 4; a++) { Lacc[b][a] = Lr[a][b]; Xacc[b][a] = (ty + 16 * a == tx + 16 * b) ? 1.0 : 0.0; }
    bool failed = false;
#pragma unroll
    for (int ja = 0; ja < 4; ja++) {
        for (int jy = 0; jy < 16; jy += 4) {
            const int j0 = 16 * ja + jy;
            if (j0 >= nb || failed) break;                  // a partial last group pairs with the identity padding
            const int p = (jy >> 2) & 1;
            const int ko = tx - jy;                         // 0..3: this thread owns a pivot column
            const bool own_rows = wv4 == jy;                // wave-uniform: this wave owns the pivot rows j0 + (ty & 3)
            LBA_TTICK(0)
            if (ko >= 0 && ko < 4) {
#pragma unroll
                for (int a = 0; a < 4; a++) sv.col[p][ko][ty + 16 * a] = Lacc[ja][a];
            }
            if (own_rows) {
#pragma unroll
                for (int b = 0; b < 4; b++) sv.row[p][lk][tx + 16 * b] = Xacc[b][ja];
            }
            LBA_TTICK(1)
            __syncthreads();
            LBA_TTICK(2)
            // the operand reads go out first: they land while the pivot chain below runs
            // A operand: -U[r_u][lk], zero for the rows of the pivot block and above (they take no update)
            const double c0u = sv.col[p][0][r_u], c1u = sv.col[p][1][r_u], c2u = sv.col[p][2][r_u], c3u = sv.col[p][3][r_u];
            double cv[4][4], rv[4][4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int c = tx + 16 * b;
                if (b >= ja) {
#pragma unroll
                    for (int k = 0; k < 4; k++) cv[b][k] = sv.col[p][k][c];
                }
                if (b <= ja) {
#pragma unroll
                    for (int k = 0; k < 4; k++) rv[b][k] = sv.row[p][k][c];
                }
            }
            // pivot block (lower triangle): P[k][m] = column m, row j0 + k
            const double P00 = sv.col[p][0][j0], P10 = sv.col[p][0][j0 + 1], P20 = sv.col[p][0][j0 + 2], P30 = sv.col[p][0][j0 + 3];
            const double P11 = sv.col[p][1][j0 + 1], P21 = sv.col[p][1][j0 + 2], P31 = sv.col[p][1][j0 + 3];
            const double P22 = sv.col[p][2][j0 + 2], P32 = sv.col[p][2][j0 + 3], P33 = sv.col[p][3][j0 + 3];
            bool ok = (P00 > 0.0) && isfinite(P00);                                 // (checked once per group: one uniform branch, not four)
            const double i0 = rsqrt_newton(P00);
            const double l10 = P10 * i0, l20 = P20 * i0, l30 = P30 * i0;
            const double d1 = fma(-l10, l10, P11);
            ok = ok && (d1 > 0.0) && isfinite(d1);
            const double i1 = rsqrt_newton(d1);
            const double l21 = fma(-l20, l10, P21) * i1, l31 = fma(-l30, l10, P31) * i1;
            const double d2 = fma(-l21, l21, fma(-l20, l20, P22));
            ok = ok && (d2 > 0.0) && isfinite(d2);
            const double i2 = rsqrt_newton(d2);
            const double l32 = fma(-l31, l21, fma(-l30, l20, P32)) * i2;
            const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, P33)));
            ok = ok && (d3 > 0.0) && isfinite(d3);
            const double i3 = rsqrt_newton(d3);
            if (!ok) { failed = true; break; }          // uniform: same values in every thread
            // M = Lp^-1 (lower triangular); this lane needs row lk of it
            const double M10 = -(l10 * i0) * i1;
            const double M20 = -fma(l21, M10, l20 * i0) * i2, M21 = -(l21 * i1) * i2;
            const double M30 = -fma(l32, M20, fma(l31, M10, l30 * i0)) * i3, M31 = -fma(l32, M21, l31 * i1) * i3, M32 = -(l32 * i2) * i3;
            const double m0 = lk == 0 ? i0 : lk == 1 ? M10 : lk == 2 ? M20 : M30;
            const double m1 = lk == 0 ? 0.0 : lk == 1 ? i1 : lk == 2 ? M21 : M31;
            const double m2 = lk < 2 ? 0.0 : lk == 2 ? i2 : M32;
            const double m3 = lk < 3 ? 0.0 : i3;
#ifdef LBA_STEP_TIMING
            if (threadIdx.x == 0 && m3 == 12345.678) d_tile_prof[7] += 1;      // (keeps the pivot chain ahead of the tick)
#endif
            LBA_TTICK(3)
            // independent FMA trees the scheduler can interleave, the MFMAs back to back after them
            double au = fma(m1, c1u, m0 * c0u) + fma(m3, c3u, m2 * c2u);
            au = (r_u > j0 + 3) ? -au : 0.0;
            double vb[4], xb[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int c = tx + 16 * b;
                if (b >= ja) {      // columns right of the pivot group belong to L:  L -= U V^T
                    vb[b] = fma(m1, cv[b][1], m0 * cv[b][0]) + fma(m3, cv[b][3], m2 * cv[b][2]);
                    if (b == ja && c <= j0 + 3) vb[b] = 0.0;
                }
                if (b <= ja)        // the others to X:  X -= U Xn, and the pivot rows of X become Xn
                    xb[b] = fma(m1, rv[b][1], m0 * rv[b][0]) + fma(m3, rv[b][3], m2 * rv[b][2]);
            }
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int c = tx + 16 * b;
                if (b >= ja) Lacc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(au, vb[b], Lacc[b], 0, 0, 0);
                if (b <= ja) {
                    const double xop = (b == ja && c > j0 + 3) ? 0.0 : xb[b];
                    Xacc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(au, xop, Xacc[b], 0, 0, 0);
                    if (own_rows) Xacc[b][ja] = xb[b];      // X[j0 + lk][c] = Xn[lk][c]
                }
            }
        }
    }
    LBA_TTICK(0)
    if (failed) return false;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = ty + 16 * a, c = tx + 16 * b;
            if (r < nb && c < nb) Li[r * NB + c] = (c <= r) ? Xacc[b][a] : 0.0;
        }
    return true;
}

// (Measured, round 2, per 60-column block: register-tile VALU updates with 16 x 16 threads 19.4 us; the same with 32 x 32 threads
// and 2 x 2 tiles 26.1 us; this MFMA form 18.7 us.  The step is bound by its dependent chain -- barrier, pivot loads, four pivots
// of rsqrt + two Newton steps at ~15 clocks per dependent v_fma_f64 (tools/probes/f64_rates.hip) -- not by f64 issue.
// EIGHT columns per barrier (8 x 8 pivot block and its inverse replicated in every thread, two MFMA k-steps per update) was built
// and is bit-compatible, but slower: 29 us per block against 22 -- the replicated pivot algebra grows with the cube of the group
// width and outweighs the publish / barrier / operand rounds it saves.  Per 4-column group (tools/lba_step_timing.py): operands +
// MFMA 1400-1700 cycles, pivot block + M 1000-1300, publish 475, barrier 290.)
__device__ __forceinline__ void chol_diag_body(const double* __restrict__ S, int n, int k0, int nb,
                                                   double* __restrict__ Linv, double* __restrict__ scal, const int bx)
{
    __shared__ CholVec4 sv;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    double Lr[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = ty + 16 * a, c = tx + 16 * b;
            Lr[a][b] = (r < nb && c < nb) ? S[(size_t)(k0 + r) * n + k0 + c] : ((r == c) ? 1.0 : 0.0);
        }
    if (!chol_tile_mfma(Lr, nb, Linv + (size_t)(k0 / NB) * NB * NB, sv) && tid == 0) scal[5] = 1.0;
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_chol_diag(const double* __restrict__ S, int n, int k0, int nb,
                                                   double* __restrict__ Linv, double* __restrict__ scal)
{
    chol_diag_body(S, n, k0, nb, Linv, scal, (int)blockIdx.x);
}

// One launch per block column K (instead of panel + update + the next diagonal factorisation): the workgroup of trailing
// tile (bi, bj), K < bj <= bi, recomputes the two panel blocks it needs, X_i = A_iK Linv_K^T and X_j (a 60^3 product each --
// cheaper than a launch), applies T_ij -= X_i X_j^T to its register tile, and the workgroup of tile (K+1, K+1) goes straight
// on to factor and invert it, while the other tiles are still being updated.  The tiles of block column K+1 also store their
// X_i (= L_iK) into a second (n+1) x n buffer Lp, which the substitution kernel then reads.  The right-hand side (row n of the (n+1) x n buffer) rides along as a 61st row of the last block row.
#ifdef LBA_STEP_TIMING       // phase times (wall clock ticks, 100 MHz) of the factoring workgroup of k_chol_step, summed (tools/lba_step_timing.py)
__device__ unsigned long long d_step_prof[8];
#define LBA_STICK(k) if (bi == K + 1 && bj == K + 1 && threadIdx.x == 0) { const unsigned long long t_now = wall_clock64(); d_step_prof[k] += t_now - t_prev; t_prev = t_now; }
#else
#define LBA_STICK(k)
#endif
constexpr int kFusedMaxBlocks = 8;       // up to 480 reduced unknowns (80 key frames); larger systems keep panel / update launches
static_assert(NB + 16 * 28 >= kFusedMaxBlocks * NB, "k_chol_solve<true> prefetches at most 28 rows per row group");
constexpr int kStepLds = (NB * (NB + 1) + 2 * 64 * (NB + 1)) * 8 + (int)sizeof(CholVec4);
__device__ __forceinline__ void chol_step_body(double* __restrict__ S, double* __restrict__ Lp, int n, int K, int nblk,
                                                   double* __restrict__ Linv, double* __restrict__ scal, const int bx, double* __restrict__ sm_step)
{
    constexpr int P = NB + 1;
    double* sI = sm_step;
    double* sXi = sI + NB * P;
    double* sXj = sXi + 64 * P;
    CholVec4& sv = *(CholVec4*)(sXj + 64 * P);
    if (scal[5] != 0.0) return;         // an earlier diagonal block was not positive definite
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    int li = 0, t = bx;
    while (t > li) { t -= li + 1; li++; }
    const int bi = K + 1 + li, bj = K + 1 + t;
    const bool diag_tile = bi == bj;
    const int k0 = K * NB;                                              // block K is a full one (it is not the last)
    const int r0 = bi * NB, nri = min(NB, n - r0) + (bi == nblk - 1 ? 1 : 0);       // + the right-hand-side row
    const int c0 = bj * NB, ncj = min(NB, n - c0);
    const double* Lk = Linv + (size_t)K * NB * NB;
#ifdef LBA_STEP_TIMING
    unsigned long long t_prev = wall_clock64();
#endif
    {
        // staging: all loads of a thread are issued before the first LDS store (16-byte loads; n = 6 * poses is even and
        // every row segment starts at an even column, so the double2 accesses are aligned)
        constexpr int H = NB / 2, kIt = (64 * H + 255) / 256;          // 30 double2 per row, 8 rounds
        double2 vI[kIt], vA[kIt], vB[kIt];
#pragma unroll
        for (int it = 0; it < kIt; it++) {
            const int i = tid + 256 * it, r = i / H, q2 = i - r * H;
            vI[it] = make_double2(0.0, 0.0); vA[it] = vI[it]; vB[it] = vI[it];
            if (r < NB) vI[it] = *(const double2*)(Lk + r * NB + 2 * q2);
            if (r < nri) vA[it] = *(const double2*)(S + (size_t)(r0 + r) * n + k0 + 2 * q2);
            if (!diag_tile && r < ncj) vB[it] = *(const double2*)(S + (size_t)(c0 + r) * n + k0 + 2 * q2);
        }
#pragma unroll
        for (int it = 0; it < kIt; it++) {
            const int i = tid + 256 * it, r = i / H, q2 = i - r * H;
            if (r < NB) { sI[r * P + 2 * q2] = vI[it].x; sI[r * P + 2 * q2 + 1] = vI[it].y; }
            if (r < 64) {
                sXi[r * P + 2 * q2] = vA[it].x; sXi[r * P + 2 * q2 + 1] = vA[it].y;
                sXj[r * P + 2 * q2] = vB[it].x; sXj[r * P + 2 * q2 + 1] = vB[it].y;
            }
        }
    }
    __syncthreads();
    LBA_STICK(0)
    // X = A Linv^T, X[r][c] = sum_{q <= c} A[r][q] Linv[c][q], on the f64 matrix pipe (v_mfma_f64_16x16x4: lane l feeds
    // A[l & 15][k = l >> 4] and B[k = l >> 4][l & 15], 1/8 of the LDS bytes of a register-blocked VALU product).  Wave w owns
    // rows 16w .. 16w+15 (it reads and overwrites only those, so the product is done in place); column block C needs the
    // k-steps up to its last column only (Linv is lower triangular).
    {
        const int wv = tid >> 6, ln = tid & 63, lr = ln & 15, lk = ln >> 4;
        mfma_d4 xa[4], xb[4];
#pragma unroll
        for (int C = 0; C < 4; C++) { xa[C] = mfma_d4{0.0, 0.0, 0.0, 0.0}; xb[C] = xa[C]; }
        const double* pa = sXi + (16 * wv + lr) * P + lk;
        const double* pb = sXj + (16 * wv + lr) * P + lk;
#pragma unroll
        for (int ks = 0; ks < NB / 4; ks++) {
            const double av = pa[4 * ks];
            const double bv = diag_tile ? 0.0 : pb[4 * ks];
#pragma unroll
            for (int C = 0; C < 4; C++) {
                if (ks >= 4 * C + 4) continue;              // compile-time: above the diagonal of Linv
                const int c = 16 * C + lr;
                const double lv = (c < NB) ? sI[c * P + 4 * ks + lk] : 0.0;
                xa[C] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, lv, xa[C], 0, 0, 0);
                if (!diag_tile) xb[C] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, lv, xb[C], 0, 0, 0);
            }
        }
        // results: lane l, component i = row (l >> 4) + 4 i, column l & 15 of the 16 x 16 block
#pragma unroll
        for (int C = 0; C < 4; C++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int c = 16 * C + lr;
                if (c < NB) {
                    sXi[(16 * wv + lk + 4 * i) * P + c] = xa[C][i];
                    if (!diag_tile) sXj[(16 * wv + lk + 4 * i) * P + c] = xb[C][i];
                }
            }
    }
    __syncthreads();
    LBA_STICK(1)
    if (bj == K + 1) {      // this tile's X_i is L_iK: keep it -- in Lp, because the other tiles of this block row still read A_iK from S
        for (int i = tid; i < nri * NB; i += 256) { const int r = i / NB, q = i - r * NB; Lp[(size_t)(r0 + r) * n + k0 + q] = sXi[r * P + q]; }
    }
    LBA_STICK(2)
    const double* sB = diag_tile ? sXi : sXj;
    double Lr[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = ty + 16 * a, c = tx + 16 * b;
            Lr[a][b] = (r < nri && c < ncj) ? S[(size_t)(r0 + r) * n + c0 + c] : 0.0;
        }
    LBA_STICK(3)
    {
        // T -= X_i X_j^T on the matrix pipe.  Wave w feeds the 16 rows its threads own (local row m = global row
        // 4w + (m & 3) + 16 (m >> 2)), so component i of column block C of the result IS this thread's element (ty + 16 i, tx + 16 C).
        const int wv = tid >> 6, ln = tid & 63, lr = ln & 15, lk = ln >> 4;
        const double* pa = sXi + (4 * wv + (lr & 3) + 16 * (lr >> 2)) * P + lk;
        const double* pb = sB + lr * P + lk;
        mfma_d4 acc[4];
#pragma unroll
        for (int C = 0; C < 4; C++) acc[C] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < NB / 4; ks++) {
            const double av = pa[4 * ks];
#pragma unroll
            for (int C = 0; C < 4; C++) acc[C] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, pb[16 * C * P + 4 * ks], acc[C], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) Lr[a][b] -= acc[b][a];
    }
    LBA_STICK(4)
    const bool factor_here = diag_tile && bi == K + 1;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = ty + 16 * a, c = tx + 16 * b;
            const bool live = r < nri && c < ncj && (!diag_tile || c <= r);
            // the tile that is factored next stays in registers, except a right-hand-side row riding along with it
            if (live && (!factor_here || r >= ncj)) S[(size_t)(r0 + r) * n + c0 + c] = Lr[a][b];
            if (factor_here && (r >= ncj || c >= ncj)) Lr[a][b] = (r == c) ? 1.0 : 0.0;
        }
    LBA_STICK(5)
    if (factor_here) {
        if (!chol_tile_mfma(Lr, ncj, Linv + (size_t)bi * NB * NB, sv) && tid == 0) scal[5] = 1.0;
    }
    LBA_STICK(6)
#ifdef LBA_STEP_TIMING
    if (bi == K + 1 && bj == K + 1 && threadIdx.x == 0) d_step_prof[7] += 1;
#endif
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_chol_step(double* __restrict__ S, double* __restrict__ Lp, int n, int K, int nblk,
                                                   double* __restrict__ Linv, double* __restrict__ scal)
{
    extern __shared__ __align__(16) double sm_step[];
    chol_step_body(S, Lp, n, K, nblk, Linv, scal, (int)blockIdx.x, sm_step);
}

// ---- the whole factorisation in ONE launch (round 3): a workgroup per lower-triangle tile (r, c) of the block matrix ----
// k_chol_diag + k_chol_step x (nblk - 1) is a chain of launches whose critical path is the factoring workgroup of every block
// column; between two of them lie a launch boundary, a tile write-back and a tile load.  Here tile (r, c) is ONE workgroup
// for its whole life: it loads its tile into registers once, and for K = 0 .. c-1 waits until block column K is factored
// (flag fac[K]) and the tiles (r, K), (c, K) are final (flags done[.][K]), recomputes the two panel blocks X_r = A_rK Linv_K^T,
// X_c (as k_chol_step does), applies T -= X_r X_c^T in registers, and at the end either factors and inverts its tile (r == c,
// publishes fac[c]) or writes it back (publishes done[r][c]).  The panel inputs of a step are final long before the pivot block
// they wait for, so everything except [load Linv_K, panel product, update] is off the critical path.
// Synchronisation between workgroups (other CUs, other XCDs): producer stores, workgroup barrier, thread 0: agent-scope
// release fence + flag store; consumer thread 0: agent-scope spin on the flag, acquire fence, workgroup barrier, plain loads
// (MI355X_MICROARCH.md, correctness boundaries).  Flags carry the EPOCH of the trial (no reset between trials).  A workgroup
// only waits for workgroups of smaller linear index (column-major tile order), so the grid cannot deadlock as long as every XCD
// starts its workgroups in index order; the launch sites keep the grid within what is resident at once anyway.  Every spin is
// bounded and also watches the failure flag (a pivot block that is not positive definite ends the factorisation for everybody).
constexpr int kFlowFlags = kFusedMaxBlocks + kFusedMaxBlocks * kFusedMaxBlocks;
__device__ __forceinline__ bool flow_wait(const unsigned* flag, unsigned epoch, double* scal)
{
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        int ok = 0;
        for (int spin = 0; spin < (1 << 21); spin++) {
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) { ok = 1; break; }
            if (__longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)(scal + 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0.0) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)(scal + 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0.0)
            __hip_atomic_store((unsigned long long*)(scal + 5), (unsigned long long)__double_as_longlong(2.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // timed out
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_ok = ok;
    }
    __syncthreads();
    const bool r = s_ok != 0;
    __syncthreads();
    return r;
}
__device__ __forceinline__ void flow_publish(unsigned* flag, unsigned epoch)
{
    __syncthreads();                // every thread's stores of the tile / the inverted block are issued and counted
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void chol_flow_body(double* __restrict__ S, double* __restrict__ Lp, int n, int nblk, double* __restrict__ Linv,
                                               double* __restrict__ scal, unsigned* __restrict__ flow, unsigned epoch, const int bx, double* __restrict__ sm_step)
{
    constexpr int P = NB + 1;
    double* sI = sm_step;
    double* sXi = sI + NB * P;
    double* sXj = sXi + 64 * P;
    CholVec4& sv = *(CholVec4*)(sXj + 64 * P);
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    // column-major tile order: (0,0) (1,0) .. (nblk-1,0) (1,1) (2,1) ..
    int c = 0, t = bx;
    while (t >= nblk - c) { t -= nblk - c; c++; }
    const int r = c + t;
    unsigned* fac = flow;
    unsigned* done = flow + kFusedMaxBlocks;
    const bool diag_tile = r == c;
    const int r0 = r * NB, nri = min(NB, n - r0) + (r == nblk - 1 ? 1 : 0);         // + the right-hand-side row
    const int c0 = c * NB, ncj = min(NB, n - c0);
    if (!diag_tile && c == 0) return;        // the tiles of block column 0 are final as they are: nothing to do, nothing to publish
    // the tile, in registers for the workgroup's whole life
    double Lr[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int rr = ty + 16 * a, cc = tx + 16 * b;
            Lr[a][b] = (rr < nri && cc < ncj) ? S[(size_t)(r0 + rr) * n + c0 + cc] : 0.0;
        }
    for (int K = 0; K < c; K++) {
        const int k0 = K * NB;
        // the panel inputs A_rK, A_cK: final once the workgroups (r, K), (c, K) are done (block column 0: from the start)
        if (K > 0) {
            if (!flow_wait(done + r * kFusedMaxBlocks + K, epoch, scal)) return;
            if (!diag_tile && !flow_wait(done + c * kFusedMaxBlocks + K, epoch, scal)) return;
        }
        {
            constexpr int H = NB / 2, kIt = (64 * H + 255) / 256;
            double2 vA[kIt], vB[kIt];
#pragma unroll
            for (int it = 0; it < kIt; it++) {
                const int i = tid + 256 * it, rr = i / H, q2 = i - rr * H;
                vA[it] = make_double2(0.0, 0.0); vB[it] = vA[it];
                if (rr < nri) vA[it] = *(const double2*)(S + (size_t)(r0 + rr) * n + k0 + 2 * q2);
                if (!diag_tile && rr < ncj) vB[it] = *(const double2*)(S + (size_t)(c0 + rr) * n + k0 + 2 * q2);
            }
#pragma unroll
            for (int it = 0; it < kIt; it++) {
                const int i = tid + 256 * it, rr = i / H, q2 = i - rr * H;
                if (rr < 64) {
                    sXi[rr * P + 2 * q2] = vA[it].x; sXi[rr * P + 2 * q2 + 1] = vA[it].y;
                    sXj[rr * P + 2 * q2] = vB[it].x; sXj[rr * P + 2 * q2 + 1] = vB[it].y;
                }
            }
        }
        // the inverted pivot block of column K: the critical wait
        if (!flow_wait(fac + K, epoch, scal)) return;
        {
            constexpr int H = NB / 2, kIt = (NB * H + 255) / 256;
            const double* Lk = Linv + (size_t)K * NB * NB;
            double2 vI[kIt];
#pragma unroll
            for (int it = 0; it < kIt; it++) {
                const int i = tid + 256 * it, rr = i / H, q2 = i - rr * H;
                vI[it] = make_double2(0.0, 0.0);
                if (rr < NB) vI[it] = *(const double2*)(Lk + rr * NB + 2 * q2);
            }
#pragma unroll
            for (int it = 0; it < kIt; it++) {
                const int i = tid + 256 * it, rr = i / H, q2 = i - rr * H;
                if (rr < NB) { sI[rr * P + 2 * q2] = vI[it].x; sI[rr * P + 2 * q2 + 1] = vI[it].y; }
            }
        }
        __syncthreads();
        // X = A Linv^T on the f64 matrix pipe, in place (as k_chol_step)
        {
            const int wv = tid >> 6, ln = tid & 63, lr = ln & 15, lk = ln >> 4;
            mfma_d4 xa[4], xb[4];
#pragma unroll
            for (int C = 0; C < 4; C++) { xa[C] = mfma_d4{0.0, 0.0, 0.0, 0.0}; xb[C] = xa[C]; }
            const double* pa = sXi + (16 * wv + lr) * P + lk;
            const double* pb = sXj + (16 * wv + lr) * P + lk;
#pragma unroll
            for (int ks = 0; ks < NB / 4; ks++) {
                const double av = pa[4 * ks];
                const double bv = diag_tile ? 0.0 : pb[4 * ks];
#pragma unroll
                for (int C = 0; C < 4; C++) {
                    if (ks >= 4 * C + 4) continue;
                    const int cc = 16 * C + lr;
                    const double lv = (cc < NB) ? sI[cc * P + 4 * ks + lk] : 0.0;
                    xa[C] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, lv, xa[C], 0, 0, 0);
                    if (!diag_tile) xb[C] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, lv, xb[C], 0, 0, 0);
                }
            }
#pragma unroll
            for (int C = 0; C < 4; C++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int cc = 16 * C + lr;
                    if (cc < NB) {
                        sXi[(16 * wv + lk + 4 * i) * P + cc] = xa[C][i];
                        if (!diag_tile) sXj[(16 * wv + lk + 4 * i) * P + cc] = xb[C][i];
                    }
                }
        }
        __syncthreads();
        if (c == K + 1) {       // this tile's X_r is L_rK: the substitution kernel reads it from Lp
            for (int i = tid; i < nri * NB; i += 256) { const int rr = i / NB, q = i - rr * NB; Lp[(size_t)(r0 + rr) * n + k0 + q] = sXi[rr * P + q]; }
        }
        const double* sB = diag_tile ? sXi : sXj;
        {
            const int wv = tid >> 6, ln = tid & 63, lr = ln & 15, lk = ln >> 4;
            const double* pa = sXi + (4 * wv + (lr & 3) + 16 * (lr >> 2)) * P + lk;
            const double* pb = sB + lr * P + lk;
            mfma_d4 acc[4];
#pragma unroll
            for (int C = 0; C < 4; C++) acc[C] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < NB / 4; ks++) {
                const double av = pa[4 * ks];
#pragma unroll
                for (int C = 0; C < 4; C++) acc[C] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, pb[16 * C * P + 4 * ks], acc[C], 0, 0, 0);
            }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) Lr[a][b] -= acc[b][a];
        }
        __syncthreads();                // sXi / sXj are free for the next block column
    }
    if (!diag_tile) {
        // final A_rc (the panel input of block column c for the tiles to its right)
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int rr = ty + 16 * a, cc = tx + 16 * b;
                if (rr < nri && cc < ncj) S[(size_t)(r0 + rr) * n + c0 + cc] = Lr[a][b];
            }
        flow_publish(done + r * kFusedMaxBlocks + c, epoch);
        return;
    }
    // diagonal tile: a right-hand-side row riding along goes back to S (the substitution reads it there), then factor + invert
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int rr = ty + 16 * a, cc = tx + 16 * b;
            if (rr < nri && cc < ncj && rr >= ncj) S[(size_t)(r0 + rr) * n + c0 + cc] = Lr[a][b];
            if (rr >= ncj || cc >= ncj) Lr[a][b] = (rr == cc) ? 1.0 : 0.0;
        }
    if (!chol_tile_mfma(Lr, ncj, Linv + (size_t)c * NB * NB, sv)) {
        if (tid == 0) __hip_atomic_store((unsigned long long*)(scal + 5), (unsigned long long)__double_as_longlong(1.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;                         // (the waiters watch the failure flag)
    }
    flow_publish(fac + c, epoch);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_chol_flow(double* __restrict__ S, double* __restrict__ Lp, int n, int nblk,
                                                   double* __restrict__ Linv, double* __restrict__ scal, unsigned* __restrict__ flow, unsigned epoch)
{
    extern __shared__ __align__(16) double sm_step[];
    chol_flow_body(S, Lp, n, nblk, Linv, scal, flow, epoch, (int)blockIdx.x, sm_step);
}

constexpr int kPanelRows = 64;
// nr = n + 1: the right-hand side b_schur is stored right behind S in the reduce buffer, i.e. it IS row n of an
// (n+1) x n row-major matrix; carrying it through panel/update as an extra row performs the forward substitution
// y = L^-1 b for free.
__global__ __launch_bounds__(1024) void k_chol_panel(double* __restrict__ S, int n, int nr, int k0, int nb, const double* __restrict__ Linv,
                                                     const double* __restrict__ scal)
{
    constexpr int P = NB + 1;
    __shared__ double sI[NB * P];
    __shared__ double sA[kPanelRows * P];
    if (scal[5] != 0.0) return;         // diagonal block was not positive definite
    const int tid = threadIdx.x;
    const int row0 = k0 + nb + blockIdx.x * kPanelRows;
    const int nrows = min(kPanelRows, nr - row0);
    const double* Li = Linv + (size_t)(k0 / NB) * NB * NB;
    for (int i = tid; i < NB * NB; i += 1024) { const int r = i / NB, c = i - r * NB; sI[r * P + c] = (r < nb && c < nb) ? Li[r * NB + c] : 0.0; }
    for (int i = tid; i < nrows * nb; i += 1024) { const int r = i / nb, c = i - r * nb; sA[r * P + c] = S[(size_t)(row0 + r) * n + k0 + c]; }
    __syncthreads();
    // X = A * Linv^T :  X[r][c] = sum_{q <= c} A[r][q] * Linv[c][q]   (Linv is stored with explicit zeros above the diagonal)
    // thread = (row, group of 4 adjacent columns): 4 independent accumulators share every a[q] load
    const int r = tid >> 4, cg = tid & 15;
    if (r < nrows && cg < NB / 4) {
        const double* a = sA + r * P;
        const double* l0 = sI + (4 * cg) * P;
        double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
        const int qmax = min(nb, 4 * cg + 4);
#pragma unroll 4
        for (int q = 0; q < qmax; q++) {
            const double aq = a[q];
            acc0 += aq * l0[q]; acc1 += aq * l0[P + q]; acc2 += aq * l0[2 * P + q]; acc3 += aq * l0[3 * P + q];
        }
        double* out = S + (size_t)(row0 + r) * n + k0 + 4 * cg;
        if (4 * cg < nb) out[0] = acc0;
        if (4 * cg + 1 < nb) out[1] = acc1;
        if (4 * cg + 2 < nb) out[2] = acc2;
        if (4 * cg + 3 < nb) out[3] = acc3;
    }
}

// trailing update S22 -= L21 L21^T (lower triangle), 32x32 tiles, panel rows staged in LDS
__global__ __launch_bounds__(256) void k_chol_update(double* __restrict__ S, int n, int nr, int k0, int nb, const double* __restrict__ scal)
{
    __shared__ double sA[32 * (NB + 1)], sB[32 * (NB + 1)];
    const int base = k0 + nb;
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti || scal[5] != 0.0) return;
    const int r0 = base + ti * 32, c0 = base + tj * 32;
    const int tid = threadIdx.x, P = NB + 1;
    for (int i = tid; i < 32 * nb; i += 256) {
        const int r = i / nb, q = i % nb;
        sA[r * P + q] = (r0 + r < nr) ? S[(size_t)(r0 + r) * n + k0 + q] : 0.0;
        sB[r * P + q] = (c0 + r < n) ? S[(size_t)(c0 + r) * n + k0 + q] : 0.0;
    }
    __syncthreads();
    const int tr = tid / 32, tc = tid % 32;
    for (int rr = tr; rr < 32; rr += 8) {
        const int r = r0 + rr, c = c0 + tc;
        if (r < nr && c < n && c <= r) {
            double sv = 0;
#pragma unroll 4
            for (int q = 0; q < nb; q++) sv += sA[rr * P + q] * sB[tc * P + q];
            S[(size_t)r * n + c] -= sv;
        }
    }
}

// x = L^-T y by block back-substitution with the inverted diagonal blocks (y = L^-1 b was produced by the factorisation
// itself, see k_chol_panel); one 1024-thread workgroup, 16 row groups x 64 columns, coalesced along the columns.
__device__ __forceinline__ double row16_sum(double v)          // sum over the 16 lanes of a DPP row (every lane gets it)
{
    v += __shfl_xor(v, 1, 16); v += __shfl_xor(v, 2, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 8, 16);
    return v;
}
// PRE (the fused path, n <= 480): the L_JK rows of block K-1 are fetched into registers while block K is being processed, so the
// serial sweep never waits for global memory.
constexpr int kSolvePre = 28;       // rows below the first block of a 480-unknown system / 16 row groups, rounded up
template <bool PRE>
__device__ __forceinline__ void chol_solve_body(const double* __restrict__ S, int n, const double* __restrict__ Linv,
                                                     const double* __restrict__ yin, const double* __restrict__ yin_last,
                                                     double* __restrict__ x, const double* __restrict__ scal, int last_forward, const int bx, double* __restrict__ sm)
{
    constexpr int P = NB + 1;
    double* y = sm;
    double* t = sm + n;
    double* part = t + 64;
    double* sL = part + 16 * 64;
    const int tid = threadIdx.x;
    if (scal[5] != 0.0) { for (int i = tid; i < n; i += 1024) x[i] = 0.0; return; }
    const int nblk = (n + NB - 1) / NB;
    // fused factorisation: S = the L panels, yin = their right-hand-side row, yin_last = the updated b of the last block
    for (int i = tid; i < n; i += 1024) y[i] = (last_forward && i >= (nblk - 1) * NB) ? yin_last[i] : yin[i];
    const int g64 = tid >> 6, r64 = tid & 63;       // 16 groups x 64 rows
    // backward sweep: x_K = Linv_KK^T (y_K - sum_{J>K} L_JK^T x_J); the inverted diagonal block is staged in LDS (its loads
    // are in flight together with those of the L_JK rows)
    double pre[PRE ? kSolvePre : 1];
    double li[4];                       // the inverted diagonal block on its way to LDS (PRE: fetched one block ahead)
    for (int K = nblk - 1; K >= 0; K--) {
        const int k0 = K * NB, nb = min(NB, n - k0);
        {
            if (K == nblk - 1 || !PRE) {
                const double* Li = Linv + (size_t)K * NB * NB;
#pragma unroll
                for (int it = 0; it < 4; it++) { const int i = tid + 1024 * it; li[it] = (i < NB * NB) ? Li[i] : 0.0; }
            }
            __syncthreads();            // y complete (first round) / previous block done with sL, t, part
#pragma unroll
            for (int it = 0; it < 4; it++) { const int i = tid + 1024 * it; if (i < NB * NB) { const int r = i / NB; sL[r * P + i - r * NB] = li[it]; } }
        }
        const int col = tid >> 4, sub = tid & 15;       // 16 lanes per column for the small matrix-vector products (row-wide reductions)
        if (K == nblk - 1 && last_forward) {    // the fused factorisation stops at the last diagonal block: y = L^-1 b for that block
            __syncthreads();
            double sv = 0;
            if (col < nb)
                for (int q = sub; q <= col; q += 16) sv += y[k0 + q] * sL[col * P + q];
            sv = row16_sum(sv);
            __syncthreads();
            if (col < nb && sub == 0) y[k0 + col] = sv;
            __syncthreads();
        }
        {
            double sv = 0;
            if (PRE) {
#pragma unroll
                for (int j = 0; j < kSolvePre; j++) { const int q = k0 + nb + g64 + 16 * j; if (q < n) sv += pre[j] * y[q]; }
            } else if (r64 < nb) {
#pragma unroll 8
                for (int q = k0 + nb + g64; q < n; q += 16) sv += S[(size_t)q * n + k0 + r64] * y[q];
            }
            part[g64 * 64 + r64] = sv;
            if (PRE && K > 0) {         // rows k0 + g64 + 16 j of block column K-1 (a full block): in flight during the rest of this block
                const double* Li = Linv + (size_t)(K - 1) * NB * NB;
#pragma unroll
                for (int it = 0; it < 4; it++) { const int i = tid + 1024 * it; li[it] = (i < NB * NB) ? Li[i] : 0.0; }
#pragma unroll
                for (int j = 0; j < kSolvePre; j++) {
                    const int q = k0 + g64 + 16 * j;
                    pre[j] = 0.0;
                    if (q < n && r64 < NB) pre[j] = S[(size_t)q * n + (k0 - NB) + r64];
                }
            }
        }
        __syncthreads();
        {
            const double tot = row16_sum(part[sub * 64 + col]);
            if (sub == 0 && col < nb) t[col] = y[k0 + col] - tot;
        }
        __syncthreads();
        {
            double sv = 0;
            if (col < nb)
                for (int q = col + sub; q < nb; q += 16) sv += sL[q * P + col] * t[q];
            sv = row16_sum(sv);
            if (sub == 0 && col < nb) y[k0 + col] = sv;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) x[i] = y[i];
}
template <bool PRE>
__global__ __launch_bounds__(1024) void k_chol_solve(const double* __restrict__ S, int n, const double* __restrict__ Linv,
                                                     const double* __restrict__ yin, const double* __restrict__ yin_last,
                                                     double* __restrict__ x, const double* __restrict__ scal, int last_forward)
{
    extern __shared__ double sm[];      // y[n], t[64], part[16][64], Linv block [NB][NB + 1]
    chol_solve_body<PRE>(S, n, Linv, yin, yin_last, x, scal, last_forward, (int)blockIdx.x, sm);
}

// ---- the trial's tail in two launches (round 3; it was four: substitution, k_backsub_update, k_errors, k_reduce) ----
// (a) the substitution workgroup goes straight on to the trial poses: oplus of every pose and the pose part of the scale sum
__device__ __forceinline__ void pose_update_tail(const Dev& d, double lambda, const double* __restrict__ bp_full,
                                                 const double* __restrict__ poses, double* __restrict__ poses_new)
{
    __syncthreads();                // x of this workgroup's substitution is complete (global memory, same workgroup)
    for (int ip = threadIdx.x; ip < d.nPoses; ip += blockDim.x) {
        const int col = d.pose_col[ip];
        if (col < 0) {
            for (int k = 0; k < 7; k++) poses_new[7 * (size_t)ip + k] = poses[7 * (size_t)ip + k];
        } else {
            const double* xp = d.x + 6 * (size_t)col;
            pose_oplus(poses + 7 * (size_t)ip, xp, poses_new + 7 * (size_t)ip);
            double sc = 0;
            for (int a = 0; a < 6; a++) sc += xp[a] * (lambda * xp[a] + bp_full[6 * (size_t)col + a]);
            d.part[d.nL + col] = sc;
        }
    }
}
template <bool PRE>
__global__ __launch_bounds__(1024) void k_chol_solve_update(const double* __restrict__ S, int n, const double* __restrict__ Linv,
                                                            const double* __restrict__ yin, const double* __restrict__ yin_last,
                                                            const double* __restrict__ scal, int last_forward,
                                                            Dev d, double lambda, const double* __restrict__ bp_full,
                                                            const double* __restrict__ poses, double* __restrict__ poses_new)
{
    extern __shared__ double sm[];
    if (n > 0) chol_solve_body<PRE>(S, n, Linv, yin, yin_last, d.x, scal, last_forward, 0, sm);
    pose_update_tail(d, lambda, bp_full, poses, poses_new);
}

// (b) EIGHT lanes per landmark (a landmark has ~10 edges; one thread per landmark walked them one after the other and the
// launch took 35 us): the lanes split the edges for the back-substitution sum W_e^T x_p and for the errors + robust chi2 of the
// landmark's own edges at the trial state (every edge belongs to exactly one landmark), butterfly sums inside the 8 lanes;
// the workgroup that finishes last sums the per-landmark partials in a fixed order and publishes the scalars to the host
// (as k_reduce mode 1 did).
constexpr int kUpdThreads = 256, kUpdLandmarks = kUpdThreads / 8;
__device__ __forceinline__ void st_agent(double* p, double v)       // agent-scope store (reaches memory every XCD sees)
{
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void update_errors_body(const Dev& d, double lambda, const double* __restrict__ pts,
                                                   const double* __restrict__ poses_new, double* __restrict__ pts_new,
                                                   double* __restrict__ hmap, unsigned long long seq, const int bx, const int n_blocks)
{
    __shared__ double s_a[kUpdThreads / 64], s_b[kUpdThreads / 64], s_c[kUpdThreads / 64];
    __shared__ unsigned int s_ticket;
    const int tid = threadIdx.x, sub = tid & 7;
    const int l = bx * kUpdLandmarks + (tid >> 3);
    const bool live = l < d.nL;
    double cs[3] = {0, 0, 0};
    int k0 = 0, k1 = 0;
    if (live) {
        k0 = d.l_off[l]; k1 = d.l_off[l + 1];
        for (int k = k0 + sub; k < k1; k += 8) {
            const int e = d.l_edge[k];
            const int col = d.pose_col[d.e_pose[e]];
            if (col < 0) continue;
            double W[18], xp[6];
            blk_load(d.W + lba_blk(e), W);
            {
                const double2* x2 = (const double2*)(d.x + 6 * (size_t)col);        // 48-byte rows of a 256-byte aligned array
                const double2 a0 = x2[0], a1 = x2[1], a2 = x2[2];
                xp[0] = a0.x; xp[1] = a0.y; xp[2] = a1.x; xp[3] = a1.y; xp[4] = a2.x; xp[5] = a2.y;
            }
#pragma unroll
            for (int q = 0; q < 3; q++) {
                double s2 = 0;
#pragma unroll
                for (int r = 0; r < 6; r++) s2 += W[r * 3 + q] * xp[r];
                cs[q] += s2;
            }
        }
    }
    for (int q = 0; q < 3; q++)
        for (int o = 4; o > 0; o >>= 1) cs[q] += __shfl_xor(cs[q], o);
    double chi_l = 0;
    if (live) {
        const double c[3] = {d.bl[3 * (size_t)l] - cs[0], d.bl[3 * (size_t)l + 1] - cs[1], d.bl[3 * (size_t)l + 2] - cs[2]};
        const double* Di = d.Dinv + 9 * (size_t)l;
        double sc = 0, Xn[3];
        for (int a = 0; a < 3; a++) {
            const double xl = Di[a * 3] * c[0] + Di[a * 3 + 1] * c[1] + Di[a * 3 + 2] * c[2];
            Xn[a] = pts[3 * (size_t)l + a] + xl;
            sc += xl * (lambda * xl + d.bl[3 * (size_t)l + a]);
            if (sub == 0) { d.x[(size_t)d.n + 3 * (size_t)l + a] = xl; pts_new[3 * (size_t)l + a] = Xn[a]; }
        }
        if (sub == 0) st_agent(d.part + l, sc);
        // errors of the landmark's edges at the trial state (computeActiveErrors + robustify)
        for (int k = k0 + sub; k < k1; k += 8) {
            const int e = d.l_edge[k];
            double Xc[3], r[3];
            pose_map(poses_new + 7 * (size_t)d.e_pose[e], Xn, Xc);
            const int st = d.e_stereo[e];
            edge_residual(d.cam, Xc, d.e_obs + 3 * (size_t)e, st, r);
            const double w = d.e_w[e];
            double chi = r[0] * (w * r[0]) + r[1] * (w * r[1]);
            if (st) chi += r[2] * (w * r[2]);
            double rho0, rho1;
            huber(d.cam, st, chi, rho0, rho1);
            d.err[3 * (size_t)e] = r[0]; d.err[3 * (size_t)e + 1] = r[1]; d.err[3 * (size_t)e + 2] = r[2];
            d.rho0[e] = rho0;
            chi_l += rho0;
        }
    }
    for (int o = 4; o > 0; o >>= 1) chi_l += __shfl_xor(chi_l, o);
    if (live && sub == 0) st_agent(d.chi_part + l, chi_l);
    // ---- the last workgroup to get here reduces.  The partials come from workgroups on other XCDs, whose L2 is not coherent with
    // this one's: they are STORED and LOADED at agent scope (write-through / cache-bypassing accesses of exactly these words) and
    // drained before the ticket.  (A __threadfence() pair here is a whole-L2 write-back + invalidate per workgroup: with 32 windows
    // per launch -- 2 000 workgroups -- it made this kernel 380 us.) ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");         // this thread's stores have left (s_waitcnt vmcnt(0))
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(d.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != (unsigned)(n_blocks - 1)) return;
    auto ld = [](const double* p) {     // agent-scope load: never served from a cache that another XCD's store has not reached
        return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    double a = 0, b = 0, c = 0;
    for (int i = tid; i < d.nL; i += kUpdThreads) { a += ld(d.chi_part + i); c += ld(d.part + i); }
    for (int i = tid; i < d.nP; i += kUpdThreads) b += ld(d.part + d.nL + i);
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
    if ((tid & 63) == 0) { s_a[tid >> 6] = a; s_b[tid >> 6] = b; s_c[tid >> 6] = c; }
    __syncthreads();
    if (tid == 0) {
        a = ((s_a[0] + s_a[1]) + s_a[2]) + s_a[3]; b = ((s_b[0] + s_b[1]) + s_b[2]) + s_b[3]; c = ((s_c[0] + s_c[1]) + s_c[2]) + s_c[3];
        d.scal[0] = a; d.scal[3] = b; d.scal[4] = c;
        __hip_atomic_store(d.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hmap) {
            hmap[0] = a; hmap[3] = b; hmap[4] = c; hmap[5] = d.scal[5];
            __threadfence_system();
            __hip_atomic_store((unsigned long long*)(hmap + 8), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ __launch_bounds__(kUpdThreads) void k_update_errors(Dev d, double lambda, const double* __restrict__ pts, const double* __restrict__ poses_new,
                                                               double* __restrict__ pts_new, double* __restrict__ hmap, unsigned long long seq)
{
    update_errors_body(d, lambda, pts, poses_new, pts_new, hmap, seq, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------
// Batched windows (lba_solve_batch): the SAME kernel bodies, one launch per stage for W independent windows, grid.y = window.
// The single-workgroup stages of a window (diagonal factorisation, block-column steps, substitution, reductions) leave the chip
// idle; W windows side by side fill it -- SURVEY.md 0 / 7 step 6: "throughput only by batching many windows per launch".
// BWin = what the single-window launches pass by value (resident on the device, written once per batch); BDynAll = the per-round
// state of every window (lambda, accepted-state index, which stages it takes part in), 2 KB passed by value with each launch.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMaxBatch = 64;
struct BWin {
    Dev d;
    double* poses[2]; double* pts[2];
    double* S; double* bs; double* bpf; double* diag; double* Lp; double* Linv;
    double* hmap;               // this window's 16 host-mapped scalars
    unsigned* flow;             // k_chol_flow flags of this window
    int nblk;
};
struct BDyn { double lambda, hint; unsigned long long seq; int cur, flags; };
struct BDynAll { BDyn w[kMaxBatch]; };
enum { kBwErrors = 1, kBwLin = 2, kBwReduce0 = 4, kBwTrial = 8, kBwSchurLm = 16 };

__global__ __launch_bounds__(256) void k_errors_b(const BWin* __restrict__ wins, BDynAll dyn, int trial_state)
{
    const BDyn y = dyn.w[blockIdx.y];
    if (!(y.flags & (trial_state ? kBwTrial : kBwErrors))) return;
    const BWin& w = wins[blockIdx.y];
    if ((int)blockIdx.x * 256 >= w.d.nE) return;
    const int st = trial_state ? 1 - y.cur : y.cur;
    errors_body(w.d, w.poses[st], w.pts[st], (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void k_lin_all_b(const BWin* __restrict__ wins, BDynAll dyn)
{
    const BDyn y = dyn.w[blockIdx.y];
    if (!(y.flags & kBwLin)) return;
    const BWin& w = wins[blockIdx.y];
    if ((int)blockIdx.x >= w.d.nP + (w.d.nL + 31) / 32) return;
    lin_all_body(w.d, w.poses[y.cur], w.pts[y.cur], y.hint, (int)blockIdx.x);
}
__global__ __launch_bounds__(1024) void k_reduce_b(const BWin* __restrict__ wins, BDynAll dyn, int mode)
{
    const BDyn y = dyn.w[blockIdx.y];
    if (!(y.flags & (mode ? kBwTrial : kBwReduce0))) return;
    const BWin& w = wins[blockIdx.y];
    reduce_body(w.d, mode, w.hmap, y.seq, 0);
}
__global__ __launch_bounds__(64) void k_schur_landmarks_b(const BWin* __restrict__ wins, BDynAll dyn)
{
    const BDyn y = dyn.w[blockIdx.y];
    if (!(y.flags & kBwSchurLm)) return;
    const BWin& w = wins[blockIdx.y];
    if ((int)blockIdx.x * 8 >= w.d.nL) return;
    schur_landmarks_body(w.d, y.lambda, (int)blockIdx.x);
}
__global__ __launch_bounds__(kSchurThreads) void k_schur_blocks_b(const BWin* __restrict__ wins, BDynAll dyn)
{
    const BDyn y = dyn.w[blockIdx.y];
    if (!(y.flags & kBwTrial)) return;
    const BWin& w = wins[blockIdx.y];
    if (blockIdx.x == 0 && threadIdx.x == 0) w.d.scal[5] = 0.0;          // (the single-window path clears the failure flag with a memset)
    if ((int)blockIdx.x >= w.d.nBlocks + w.d.nP) return;
    schur_blocks_body(w.d, w.S, y.lambda, w.bs, w.bpf, w.diag, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_chol_diag_b(const BWin* __restrict__ wins, BDynAll dyn)
{
    const BDyn y = dyn.w[blockIdx.y];
    if (!(y.flags & kBwTrial)) return;
    const BWin& w = wins[blockIdx.y];
    if (w.d.n <= 0) return;
    chol_diag_body(w.S, w.d.n, 0, min(NB, w.d.n), w.Linv, w.d.scal, 0);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_chol_step_b(const BWin* __restrict__ wins, BDynAll dyn, int K)
{
    extern __shared__ __align__(16) double sm_step[];
    const BDyn y = dyn.w[blockIdx.y];
    if (!(y.flags & kBwTrial)) return;
    const BWin& w = wins[blockIdx.y];
    const int T = w.nblk - 1 - K;
    if (T <= 0 || (int)blockIdx.x >= T * (T + 1) / 2) return;
    chol_step_body(w.S, w.Lp, w.d.n, K, w.nblk, w.Linv, w.d.scal, (int)blockIdx.x, sm_step);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_chol_flow_b(const BWin* __restrict__ wins, BDynAll dyn)
{
    extern __shared__ __align__(16) double sm_step[];
    const BDyn y = dyn.w[blockIdx.y];
    if (!(y.flags & kBwTrial)) return;
    const BWin& w = wins[blockIdx.y];
    if (w.d.n <= 0 || (int)blockIdx.x >= w.nblk * (w.nblk + 1) / 2) return;
    chol_flow_body(w.S, w.Lp, w.d.n, w.nblk, w.Linv, w.d.scal, w.flow, (unsigned)y.seq, (int)blockIdx.x, sm_step);
}
__global__ __launch_bounds__(1024) void k_chol_solve_b(const BWin* __restrict__ wins, BDynAll dyn)
{
    extern __shared__ double sm[];
    const BDyn y = dyn.w[blockIdx.y];
    if (!(y.flags & kBwTrial)) return;
    const BWin& w = wins[blockIdx.y];
    const int n = w.d.n;
    if (n <= 0) return;
    chol_solve_body<true>(w.Lp, n, w.Linv, w.Lp + (size_t)n * n, w.bs, w.d.x, w.d.scal, 1, 0, sm);
}
__global__ __launch_bounds__(1024) void k_chol_solve_update_b(const BWin* __restrict__ wins, BDynAll dyn)
{
    extern __shared__ double sm[];
    const BDyn y = dyn.w[blockIdx.y];
    if (!(y.flags & kBwTrial)) return;
    const BWin& w = wins[blockIdx.y];
    const int n = w.d.n;
    if (n > 0) chol_solve_body<true>(w.Lp, n, w.Linv, w.Lp + (size_t)n * n, w.bs, w.d.x, w.d.scal, 1, 0, sm);
    pose_update_tail(w.d, y.lambda, w.bpf, w.poses[y.cur], w.poses[1 - y.cur]);
}
__global__ __launch_bounds__(kUpdThreads) void k_update_errors_b(const BWin* __restrict__ wins, BDynAll dyn)
{
    const BDyn y = dyn.w[blockIdx.y];
    if (!(y.flags & kBwTrial)) return;
    const BWin& w = wins[blockIdx.y];
    const int nb = max((w.d.nL + kUpdLandmarks - 1) / kUpdLandmarks, 1);
    if ((int)blockIdx.x >= nb) return;
    update_errors_body(w.d, y.lambda, w.pts[y.cur], w.poses[1 - y.cur], w.pts[1 - y.cur], w.hmap, y.seq, (int)blockIdx.x, nb);
}

__global__ __launch_bounds__(256) void k_epilogue(Dev d, const double* __restrict__ poses, const double* __restrict__ pts,
                                                  double* __restrict__ chi2, uint8_t* __restrict__ depth_pos)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= d.nE) return;
    const double* r = d.err + 3 * (size_t)e;
    const double w = d.e_w[e];
    double chi = r[0] * (w * r[0]) + r[1] * (w * r[1]);
    if (d.e_stereo[e]) chi += r[2] * (w * r[2]);
    chi2[e] = chi;
    double Xc[3];
    pose_map(poses + 7 * (size_t)d.e_pose[e], pts + 3 * (size_t)d.e_point[e], Xc);
    depth_pos[e] = Xc[2] > 0.0;
}

__global__ void k_normalize_poses(double* poses, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) quat_normalize(poses + 7 * (size_t)i);       // SE3Quat(Quaterniond, Vector3d) ctor (Optimizer.cc:1217)
}

}  // namespace lba

namespace lba {
// stages of the per-stage profile (lba_shard_profile_read); kStageIdle = host gaps between the groups of launches
enum { kStageLinearize = 0, kStageSchur = 1, kStageFactor = 2, kStageSolve = 3, kStageUpdate = 4, kStageReduce = 5, kStageIdle = 6, kStageCount = 7 };
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct lba_shard {
    int device = 0;
    hipStream_t stream = nullptr;
    lba::Dev d;
    int nblk = 0;
    std::vector<void*> allocs;
    double *poses[2] = {nullptr, nullptr}, *pts[2] = {nullptr, nullptr};
    double *poses0 = nullptr, *pts0 = nullptr;      // initial estimates (lba_shard_reset)
    int cur = 0;                // index of the accepted state; 1-cur holds the trial state
    double* reduce = nullptr;   // [n*n | bs n | bp n | diag n]
    double* Linv = nullptr;
    double* Lp = nullptr;       // L panels of the fused factorisation, (n+1) x n like the reduce buffer's S | b_schur
    double hint_lambda = -1.0;  // lambda the next linearisation may pre-compute the landmark side of the Schur complement for
    double schur_lambda = -1.0; // lambda that pre-computation is valid for (consumed by the next lba_shard_reduce)
    double* Ldiag = nullptr;
    bool sync_after_reduce = true;      // lba_solve() keeps everything on one stream and turns this off
    bool lambda_in_reduce = false;      // single-GPU: add lambda to diag(S) inside k_schur_blocks (no all-reduce in between)
    bool lambda_added = false;
    double* d_chi2 = nullptr;
    uint8_t* d_depth = nullptr;
    double* h_scal = nullptr;   // pinned, host-mapped and coherent [16]: [0..5] scalars, [8] sequence number of the last k_reduce
    double* d_hmap = nullptr;   // the same buffer as the device sees it
    unsigned long long seq = 0;
    int64_t reduce_len = 0;
    bool err_valid = false;
    bool err_current = false;           // d.err / d.rho0 belong to the accepted state poses[cur]
    double chi_current = 0, chi_trial = 0, mdp_cached = 0, mdl_cached = 0;

    // optional per-stage timing with HIP events on the shard's stream (lba_shard_profile_*): mark(stage) closes the previous
    // interval and opens one that is charged to `stage`
    static constexpr int kProfMarks = 1024;
    bool profile = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_stage;
    int prof_n = 0;
    hipEvent_t ev_fence = nullptr;      // stream hand-over to / from the collective's stream (lba_shard_fence_*)
    unsigned* flow = nullptr;           // k_chol_flow: per-tile flags (epoch of the trial that published them)
    unsigned flow_epoch = 0;
    double* d_coll = nullptr;           // lba_shard_optimize: device scratch of the scalar all-reduces (chi2 / scale / flags) ...
    double* h_coll = nullptr;           // ... and its pinned host mirror
    void mark(int stage)
    {
        if (!profile || prof_n >= kProfMarks) return;
        (void)hipEventRecord(prof_ev[prof_n], stream);
        prof_stage[prof_n++] = stage;
    }

    // optional bump arena owned by an lba_solver (avoids ~40 hipMalloc/hipFree per LocalBundleAdjustment call)
    uint8_t* arena = nullptr;
    size_t arena_cap = 0, arena_off = 0, bytes_wanted = 0, upload_bytes = 0;
    bool owns_stream = true, owns_hscal = true;
    // optional pinned mirror of the arena's prefix (also the solver's): the problem arrays are packed there and go up in ONE
    // asynchronous copy instead of ~20 synchronous ones from pageable memory
    uint8_t* stage = nullptr;
    size_t stage_cap = 0, stage_end = 0;

    template <typename T>
    int dalloc(T** p, size_t count)
    {
        *p = nullptr;
        const size_t bytes = (std::max(count, (size_t)1) * sizeof(T) + 255) & ~(size_t)255;
        bytes_wanted += bytes;
        if (arena && arena_off + bytes <= arena_cap) {
            *p = (T*)(arena + arena_off);
            arena_off += bytes;
            return ORBX_OK;
        }
        LBA_HIP(hipMalloc((void**)p, bytes));
        allocs.push_back(*p);
        return ORBX_OK;
    }
#define LBA_TRY_RET(x) do { const int r_ = (x); if (r_) return r_; } while (0)
    template <typename T>
    int upload(const T** p, const std::vector<T>& v)
    {
        T* q;
        int r = dalloc(&q, v.size());
        if (r) return r;
        if (!v.empty()) LBA_TRY_RET(put(q, v.data(), v.size() * sizeof(T)));
        *p = q;
        return ORBX_OK;
    }
    template <typename T>
    int upload_raw(const T** p, const T* src, size_t count)
    {
        T* q;
        int r = dalloc(&q, count);
        if (r) return r;
        if (count) LBA_TRY_RET(put(q, src, count * sizeof(T)));
        *p = q;
        return ORBX_OK;
    }
    // host -> device: through the pinned mirror when the destination lies in the mirrored arena prefix (flushed by flush_stage)
    int put(void* dst, const void* src, size_t bytes)
    {
        const uint8_t* d8 = (const uint8_t*)dst;
        if (stage && arena && d8 >= arena && (size_t)(d8 - arena) + bytes <= stage_cap) {
            const size_t off = (size_t)(d8 - arena);
            std::memcpy(stage + off, src, bytes);
            stage_end = std::max(stage_end, off + bytes);
            return ORBX_OK;
        }
        LBA_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        return ORBX_OK;
    }
    int flush_stage()
    {
        if (stage_end > 0) LBA_HIP(hipMemcpyAsync(arena, stage, stage_end, hipMemcpyHostToDevice, stream));
        stage_end = 0;
        return ORBX_OK;
    }
    double* S() { return reduce; }
    double* bs() { return reduce + (size_t)d.n * d.n; }
    double* bpf() { return reduce + (size_t)d.n * d.n + d.n; }
    double* diag() { return reduce + (size_t)d.n * d.n + 2 * (size_t)d.n; }
};

static int shard_validate(const LbaProblem* p)
{
    if (!p) return fail(ORBX_ERR_ARG, "NULL problem");
    if (p->n_poses < 1 || p->n_points < 0 || p->n_edges < 0) return fail(ORBX_ERR_ARG, "bad problem sizes");
    if (!p->pose_q || !p->pose_t || !p->pose_fixed) return fail(ORBX_ERR_ARG, "NULL pose arrays");
    if (p->n_points > 0 && !p->points) return fail(ORBX_ERR_ARG, "NULL points");
    if (p->n_edges > 0 && (!p->edge_point || !p->edge_pose || !p->edge_obs || !p->edge_inv_sigma2 || !p->edge_stereo))
        return fail(ORBX_ERR_ARG, "NULL edge arrays");
    for (int e = 0; e < p->n_edges; e++)
        if (p->edge_point[e] < 0 || p->edge_point[e] >= p->n_points || p->edge_pose[e] < 0 || p->edge_pose[e] >= p->n_poses)
            return fail(ORBX_ERR_ARG, "edge %d references vertex out of range", e);
    return ORBX_OK;
}

struct lba_solver {
    int device = 0;
    uint8_t* arena = nullptr;
    size_t arena_cap = 0;
    hipStream_t stream = nullptr;
    double* h_scal = nullptr;
    uint8_t* stage = nullptr;       // pinned mirror of the arena prefix that holds the uploaded arrays
    size_t stage_cap = 0;
};

extern "C" void lba_shard_destroy(lba_shard* s);

// grow a solver's arena / pinned staging mirror so that the next window of this size needs no hipMalloc
static void solver_grow(lba_solver* sv, size_t wanted, size_t wanted_stage)
{
    if (wanted > sv->arena_cap) {
        if (sv->arena) (void)hipFree(sv->arena);
        sv->arena = nullptr; sv->arena_cap = 0;
        const size_t cap = wanted + wanted / 4 + (1 << 20);
        if (hipMalloc((void**)&sv->arena, cap) == hipSuccess) sv->arena_cap = cap;
    }
    if (wanted_stage > sv->stage_cap) {
        if (sv->stage) (void)hipHostFree(sv->stage);
        sv->stage = nullptr; sv->stage_cap = 0;
        const size_t cap = wanted_stage + wanted_stage / 4 + (1 << 16);
        if (hipHostMalloc((void**)&sv->stage, cap) == hipSuccess) sv->stage_cap = cap;
    }
}

static int shard_create_impl(int device, const LbaProblem* p, lba_shard** out, lba_solver* owner)
{
    if (!out) return fail(ORBX_ERR_ARG, "out is NULL");
    *out = nullptr;
    int r = shard_validate(p);
    if (r) return r;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ORBX_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(ORBX_ERR_ARG, "device %d out of range", device);
    LBA_HIP(hipSetDevice(device));
    lba_shard* s = new lba_shard();
    s->device = device;
    if (owner) {
        s->arena = owner->arena; s->arena_cap = owner->arena_cap;
        s->stream = owner->stream; s->owns_stream = false;
        s->h_scal = owner->h_scal; s->owns_hscal = false;
        s->stage = owner->stage; s->stage_cap = owner->stage_cap;
    }
    std::memset(&s->d, 0, sizeof(s->d));
    lba::Dev& d = s->d;
    static const bool build_timing = std::getenv("ORBX_LBA_TIMING") != nullptr;
    const auto tb0 = std::chrono::steady_clock::now();
    d.nPoses = p->n_poses; d.nL = p->n_points; d.nE = p->n_edges;
    std::vector<int> pose_col(p->n_poses, -1), col_pose;
    for (int i = 0; i < p->n_poses; i++) if (!p->pose_fixed[i]) { pose_col[i] = (int)col_pose.size(); col_pose.push_back(i); }
    d.nP = (int)col_pose.size();
    d.n = 6 * d.nP;
    // CSR by landmark / by pose column (caller order inside)
    std::vector<int> l_off(d.nL + 1, 0), p_off(d.nP + 1, 0);
    for (int e = 0; e < d.nE; e++) { l_off[p->edge_point[e] + 1]++; const int c = pose_col[p->edge_pose[e]]; if (c >= 0) p_off[c + 1]++; }
    for (int l = 0; l < d.nL; l++) l_off[l + 1] += l_off[l];
    for (int c = 0; c < d.nP; c++) p_off[c + 1] += p_off[c];
    std::vector<int> l_edge(std::max(d.nE, 1)), p_edge(std::max(p_off[d.nP], 1)), lc(l_off.begin(), l_off.end() - 1), pc(p_off.begin(), p_off.end() - 1);
    for (int e = 0; e < d.nE; e++) { l_edge[lc[p->edge_point[e]]++] = e; const int c = pose_col[p->edge_pose[e]]; if (c >= 0) p_edge[pc[c]++] = e; }
    const auto tb1 = std::chrono::steady_clock::now();
    // pair list per block (i <= j).  A landmark is seen at most once by a pose (one observation per key frame), so a block gets at most ONE
    // pair per landmark and its pairs stand in landmark order whatever the order inside a landmark: each landmark's (column, edge) list
    // is sorted by column once and only the a <= b half is walked (110 k steps for the bench window instead of two passes of 200 k with
    // a test).  Two edges of one landmark on one pose (never built by the reference's graph walks) take the general double loop, whose
    // order inside a landmark is the caller's.
    std::vector<int> ecol(std::max(d.nE, 1));
    for (int e = 0; e < d.nE; e++) ecol[e] = pose_col[p->edge_pose[e]];
    std::vector<int> cnt((size_t)d.nP * d.nP, 0);
    std::vector<int> s_col(std::max(d.nE, 1)), s_edge(std::max(d.nE, 1)), s_off(d.nL + 1, 0);
    bool dup = false;
    {
        int w = 0;
        for (int l = 0; l < d.nL; l++) {
            const int w0 = w;
            for (int a = l_off[l]; a < l_off[l + 1]; a++) {
                const int e = l_edge[a], c = ecol[e];
                if (c < 0) continue;
                int q = w++;                                   // insertion by column (a handful of entries per landmark), stable
                while (q > w0 && s_col[q - 1] > c) { s_col[q] = s_col[q - 1]; s_edge[q] = s_edge[q - 1]; q--; }
                if (q > w0 && s_col[q - 1] == c) dup = true;
                s_col[q] = c; s_edge[q] = e;
            }
            s_off[l + 1] = w;
        }
    }
    if (std::getenv("ORBX_LBA_PAIRS_GENERAL")) dup = true;        // (tests: both builders must give the same lists)
    if (!dup) {
        for (int l = 0; l < d.nL; l++)
            for (int a = s_off[l]; a < s_off[l + 1]; a++) {
                int* row = cnt.data() + (size_t)s_col[a] * d.nP;
                for (int b = a; b < s_off[l + 1]; b++) row[s_col[b]]++;
            }
    } else {
        for (int l = 0; l < d.nL; l++)
            for (int a = l_off[l]; a < l_off[l + 1]; a++) {
                const int i = ecol[l_edge[a]];
                if (i < 0) continue;
                int* row = cnt.data() + (size_t)i * d.nP;
                for (int b = l_off[l]; b < l_off[l + 1]; b++) {
                    const int j = ecol[l_edge[b]];
                    if (j >= i) row[j]++;
                }
            }
    }
    // every block of the upper triangle gets an entry (possibly with an empty pair list)
    d.nBlocks = d.nP * (d.nP + 1) / 2;
    std::vector<int> b_i(std::max(d.nBlocks, 1)), b_j(std::max(d.nBlocks, 1)), b_off(d.nBlocks + 1, 0), blk_of((size_t)d.nP * d.nP, -1);
    {
        int k = 0;
        for (int i = 0; i < d.nP; i++)
            for (int j = i; j < d.nP; j++, k++) {
                blk_of[(size_t)i * d.nP + j] = k;
                b_i[k] = i; b_j[k] = j;
                b_off[k + 1] = b_off[k] + cnt[(size_t)i * d.nP + j];
            }
    }
    std::vector<int2> pairs(std::max(b_off.back(), 1));
    std::vector<int> bc(b_off.begin(), b_off.end() - 1);
    if (!dup) {
        for (int l = 0; l < d.nL; l++)
            for (int a = s_off[l]; a < s_off[l + 1]; a++) {
                const int ea = s_edge[a];
                const int* brow = blk_of.data() + (size_t)s_col[a] * d.nP;
                for (int b = a; b < s_off[l + 1]; b++) { int2 pr; pr.x = ea; pr.y = s_edge[b]; pairs[bc[brow[s_col[b]]]++] = pr; }
            }
    } else {
        for (int l = 0; l < d.nL; l++)
            for (int a = l_off[l]; a < l_off[l + 1]; a++) {
                const int ea = l_edge[a], i = ecol[ea];
                if (i < 0) continue;
                const int* brow = blk_of.data() + (size_t)i * d.nP;
                for (int b = l_off[l]; b < l_off[l + 1]; b++) {
                    const int eb = l_edge[b], j = ecol[eb];
                    if (j < i) continue;
                    int2 pr; pr.x = ea; pr.y = eb;
                    pairs[bc[brow[j]]++] = pr;
                }
            }
    }
    const auto tb2 = std::chrono::steady_clock::now();
    std::vector<double> poses(7 * (size_t)p->n_poses);
    for (int i = 0; i < p->n_poses; i++) {
        for (int k = 0; k < 4; k++) poses[7 * i + k] = p->pose_q[4 * i + k];
        for (int k = 0; k < 3; k++) poses[7 * i + 4 + k] = p->pose_t[3 * i + k];
    }
#define LBA_TRY(x) do { r = (x); if (r) { lba_shard_destroy(s); return r; } } while (0)
    LBA_TRY(s->upload(&d.pose_col, pose_col)); LBA_TRY(s->upload(&d.col_pose, col_pose));
    LBA_TRY(s->upload_raw(&d.e_point, p->edge_point, (size_t)d.nE)); LBA_TRY(s->upload_raw(&d.e_pose, p->edge_pose, (size_t)d.nE));
    LBA_TRY(s->upload_raw(&d.e_obs, p->edge_obs, 3 * (size_t)d.nE));
    LBA_TRY(s->upload_raw(&d.e_w, p->edge_inv_sigma2, (size_t)d.nE)); LBA_TRY(s->upload_raw(&d.e_stereo, p->edge_stereo, (size_t)d.nE));
    LBA_TRY(s->upload(&d.l_off, l_off)); LBA_TRY(s->upload(&d.l_edge, l_edge)); LBA_TRY(s->upload(&d.p_off, p_off)); LBA_TRY(s->upload(&d.p_edge, p_edge));
    LBA_TRY(s->upload(&d.b_i, b_i)); LBA_TRY(s->upload(&d.b_j, b_j)); LBA_TRY(s->upload(&d.b_off, b_off)); LBA_TRY(s->upload(&d.b_pair, pairs));
    LBA_TRY(s->dalloc(&s->poses[0], 7 * (size_t)p->n_poses)); LBA_TRY(s->dalloc(&s->pts[0], 3 * (size_t)d.nL));
    s->upload_bytes = s->bytes_wanted;      // everything the host writes lies in front of this offset
    LBA_TRY(s->dalloc(&d.Hll, 9 * (size_t)d.nL)); LBA_TRY(s->dalloc(&d.bl, 3 * (size_t)d.nL));
    LBA_TRY(s->dalloc(&d.Hpp, 36 * (size_t)d.nP)); LBA_TRY(s->dalloc(&d.bp, 6 * (size_t)d.nP));
    LBA_TRY(s->dalloc(&d.W, lba::kBlk * (size_t)d.nE)); LBA_TRY(s->dalloc(&d.Z, lba::kBlk * (size_t)d.nE));
    LBA_TRY(s->dalloc(&d.Dinv, 9 * (size_t)d.nL)); LBA_TRY(s->dalloc(&d.db, 3 * (size_t)d.nL));
    LBA_TRY(s->dalloc(&d.err, 3 * (size_t)d.nE)); LBA_TRY(s->dalloc(&d.rho0, (size_t)d.nE));
    LBA_TRY(s->dalloc(&d.x, (size_t)d.n + 3 * (size_t)d.nL)); LBA_TRY(s->dalloc(&d.part, (size_t)d.nL + d.nP));
    LBA_TRY(s->dalloc(&d.chi_part, (size_t)d.nL)); LBA_TRY(s->dalloc(&d.ticket, 64));
    LBA_TRY(s->dalloc(&d.scal, 16));
    LBA_TRY(s->dalloc(&s->poses[1], 7 * (size_t)p->n_poses)); LBA_TRY(s->dalloc(&s->pts[1], 3 * (size_t)d.nL));
    s->reduce_len = (int64_t)d.n * d.n + 3 * (int64_t)d.n;
    LBA_TRY(s->dalloc(&s->reduce, (size_t)s->reduce_len));
    s->nblk = (d.n + lba::NB - 1) / lba::NB;
    LBA_TRY(s->dalloc(&s->Linv, (size_t)std::max(s->nblk, 1) * lba::NB * lba::NB));
    if (s->nblk <= lba::kFusedMaxBlocks) LBA_TRY(s->dalloc(&s->Lp, ((size_t)d.n + 1) * (size_t)std::max(d.n, 1)));
    // (the dynamic-LDS limits of the factorisation / substitution kernels: once per device and process, not per window)
    static std::atomic<unsigned long long> attr_done{0};
    const bool set_attr = !((attr_done.load() >> device) & 1ull);
    if (set_attr) {
        LBA_HIP(hipFuncSetAttribute((const void*)lba::k_chol_step, hipFuncAttributeMaxDynamicSharedMemorySize, lba::kStepLds));
        LBA_HIP(hipFuncSetAttribute((const void*)lba::k_chol_flow, hipFuncAttributeMaxDynamicSharedMemorySize, lba::kStepLds));
    }
    LBA_TRY(s->dalloc(&s->flow, (size_t)lba::kFlowFlags));
    LBA_HIP(hipMemsetAsync(s->flow, 0, lba::kFlowFlags * sizeof(unsigned), s->stream));
    {
        const size_t solve_lds = ((size_t)d.n + 64 + 16 * 64 + lba::NB * (lba::NB + 1)) * sizeof(double);
        if (solve_lds > 160 * 1024) LBA_TRY(fail(ORBX_ERR_CAPACITY, "%d reduced unknowns exceed the substitution kernel's LDS", d.n));
        if (set_attr) {     // (a limit, not an allocation: the largest system the check above lets through)
            const int lim = 160 * 1024;
            LBA_HIP(hipFuncSetAttribute((const void*)lba::k_chol_solve<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            LBA_HIP(hipFuncSetAttribute((const void*)lba::k_chol_solve<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            LBA_HIP(hipFuncSetAttribute((const void*)lba::k_chol_solve_update<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            LBA_HIP(hipFuncSetAttribute((const void*)lba::k_chol_solve_update<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            attr_done.fetch_or(1ull << device);
        }
    }
    LBA_TRY(s->dalloc(&s->Ldiag, (size_t)lba::NB * lba::NB));
    LBA_TRY(s->dalloc(&s->d_chi2, (size_t)d.nE)); LBA_TRY(s->dalloc(&s->d_depth, (size_t)d.nE));
    if (s->owns_hscal && hipHostMalloc((void**)&s->h_scal, 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { lba_shard_destroy(s); return fail(ORBX_ERR_HIP, "hipHostMalloc failed"); }
    if (hipHostGetDevicePointer((void**)&s->d_hmap, s->h_scal, 0) != hipSuccess) { lba_shard_destroy(s); return fail(ORBX_ERR_HIP, "hipHostGetDevicePointer failed"); }
    s->seq = *(const unsigned long long*)(s->h_scal + 8);      // a solver handle reuses the buffer across shards: continue its numbering
    if (s->owns_hscal) { std::memset(s->h_scal, 0, 16 * sizeof(double)); s->seq = 0; }
    if (s->owns_stream && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { lba_shard_destroy(s); return fail(ORBX_ERR_HIP, "stream create failed"); }
#undef LBA_TRY
    if ((r = s->put(s->poses[0], poses.data(), poses.size() * sizeof(double))) || (d.nL > 0 && (r = s->put(s->pts[0], p->points, 3 * (size_t)d.nL * sizeof(double)))) ||
        (r = s->flush_stage())) { lba_shard_destroy(s); return r; }
    LBA_HIP(hipMemsetAsync(d.err, 0, 3 * (size_t)std::max(d.nE, 1) * sizeof(double), s->stream));
    LBA_HIP(hipMemsetAsync(d.scal, 0, 16 * sizeof(double), s->stream));
    LBA_HIP(hipMemsetAsync(d.ticket, 0, 64 * sizeof(unsigned int), s->stream));
    d.cam.fx = p->fx; d.cam.fy = p->fy; d.cam.cx = p->cx; d.cam.cy = p->cy; d.cam.bf = p->bf;
    d.cam.huber_mono = p->huber_mono; d.cam.huber_stereo = p->huber_stereo;
    d.cam.dsqr_mono = p->huber_mono * p->huber_mono; d.cam.dsqr_stereo = p->huber_stereo * p->huber_stereo;    // RobustKernelHuber::setDelta
    hipLaunchKernelGGL(lba::k_normalize_poses, dim3((p->n_poses + 63) / 64), dim3(64), 0, s->stream, s->poses[0], p->n_poses);
    if (!owner) {       // a standalone shard can be reset to its initial estimates (lba_shard_reset); lba_solve never does that
        if ((r = s->dalloc(&s->poses0, 7 * (size_t)p->n_poses)) || (r = s->dalloc(&s->pts0, 3 * (size_t)d.nL))) { lba_shard_destroy(s); return r; }
        LBA_HIP(hipMemcpyAsync(s->poses0, s->poses[0], 7 * (size_t)p->n_poses * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        if (d.nL > 0) LBA_HIP(hipMemcpyAsync(s->pts0, s->pts[0], 3 * (size_t)d.nL * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
        LBA_HIP(hipStreamSynchronize(s->stream));
    }
    if (build_timing) {
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::fprintf(stderr, "[lba structure] CSR %.0f us, pair lists %.0f us, staging + enqueue %.0f us\n", us(tb0, tb1), us(tb1, tb2), us(tb2, std::chrono::steady_clock::now()));
    }
    *out = s;
    return ORBX_OK;
}

#ifdef LBA_STEP_TIMING
extern "C" int lba_debug_step_prof(unsigned long long* out8)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lba::d_step_prof), sizeof(z)) != hipSuccess) return ORBX_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(lba::d_step_prof), z, sizeof(z)) != hipSuccess) return ORBX_ERR_HIP;
    return ORBX_OK;
}
extern "C" int lba_debug_tile_prof(unsigned long long* out8)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lba::d_tile_prof), sizeof(z)) != hipSuccess) return ORBX_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(lba::d_tile_prof), z, sizeof(z)) != hipSuccess) return ORBX_ERR_HIP;
    return ORBX_OK;
}
#endif

extern "C" int lba_shard_create(int device, const LbaProblem* p, lba_shard** out) { return shard_create_impl(device, p, out, nullptr); }

extern "C" {

// restore the initial estimates (lets a benchmark re-run the optimisation without re-uploading the problem)
int lba_shard_reset(lba_shard* s)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    LBA_HIP(hipSetDevice(s->device));
    if (!s->poses0) return fail(ORBX_ERR_ARG, "this shard keeps no copy of its initial estimates");
    s->cur = 0;
    s->err_current = false;
    LBA_HIP(hipMemcpyAsync(s->poses[0], s->poses0, 7 * (size_t)s->d.nPoses * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    if (s->d.nL > 0) LBA_HIP(hipMemcpyAsync(s->pts[0], s->pts0, 3 * (size_t)s->d.nL * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    LBA_HIP(hipMemsetAsync(s->d.err, 0, 3 * (size_t)std::max(s->d.nE, 1) * sizeof(double), s->stream));
    return ORBX_OK;
}

void lba_shard_destroy(lba_shard* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); if (s->owns_stream) (void)hipStreamDestroy(s->stream); }
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->h_scal && s->owns_hscal) (void)hipHostFree(s->h_scal);
    for (hipEvent_t e : s->prof_ev) (void)hipEventDestroy(e);
    if (s->ev_fence) (void)hipEventDestroy(s->ev_fence);
    if (s->h_coll) (void)hipHostFree(s->h_coll);
    delete s;
}

int64_t lba_shard_reduce_len(const lba_shard* s) { return s ? s->reduce_len : 0; }
double* lba_shard_reduce_buffer(lba_shard* s) { return s ? s->reduce : nullptr; }

// local != 0: the caller promises that NO all-reduce happens between lba_shard_reduce() and lba_shard_finish() (world size 1):
// lambda is then added to diag(S) inside the Schur kernel and the stream is not synchronised after reduce().
int lba_shard_set_local(lba_shard* s, int local)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    s->lambda_in_reduce = local != 0;
    s->sync_after_reduce = local == 0;
    return ORBX_OK;
}

// Per-stage device time (HIP events on the shard's stream).  enable: start a fresh profile; read: milliseconds per stage summed
// over everything recorded since, stage_ms[7] = {linearise, Schur complement, factorisation, substitution, update + errors,
// reductions, host gaps}.  The events serialise nothing, but the profile is meant for measurement runs, not for production.
int lba_shard_profile_enable(lba_shard* s, int on)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    LBA_HIP(hipSetDevice(s->device));
    if (on && s->prof_ev.empty()) {
        s->prof_ev.resize(lba_shard::kProfMarks);
        s->prof_stage.assign(lba_shard::kProfMarks, 0);
        for (auto& e : s->prof_ev) LBA_HIP(hipEventCreate(&e));
    }
    s->profile = on != 0;
    s->prof_n = 0;
    return ORBX_OK;
}

int lba_shard_profile_read(lba_shard* s, float* stage_ms, int n_stages)
{
    if (!s || !stage_ms) return fail(ORBX_ERR_ARG, "NULL argument");
    LBA_HIP(hipSetDevice(s->device));
    LBA_HIP(hipStreamSynchronize(s->stream));
    for (int i = 0; i < n_stages; i++) stage_ms[i] = 0.f;
    for (int i = 0; i + 1 < s->prof_n; i++) {
        float ms = 0.f;
        LBA_HIP(hipEventElapsedTime(&ms, s->prof_ev[i], s->prof_ev[i + 1]));
        if (s->prof_stage[i] < n_stages) stage_ms[s->prof_stage[i]] += ms;
    }
    s->prof_n = 0;
    return ORBX_OK;
}

// Stream hand-over for the sharded global BA (no host synchronisation): fence_out makes `other` (the collective's stream) wait
// for everything enqueued on the shard's stream so far; fence_in makes the shard's stream wait for `other`.  With
// lba_shard_set_async_reduce(1) lba_shard_reduce() returns without synchronising and the caller brackets its all-reduce with
// the two fences.
int lba_shard_fence_out(lba_shard* s, void* other)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    LBA_HIP(hipSetDevice(s->device));
    if (!s->ev_fence) LBA_HIP(hipEventCreateWithFlags(&s->ev_fence, hipEventDisableTiming));
    LBA_HIP(hipEventRecord(s->ev_fence, s->stream));
    LBA_HIP(hipStreamWaitEvent((hipStream_t)other, s->ev_fence, 0));
    return ORBX_OK;
}

int lba_shard_fence_in(lba_shard* s, void* other)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    LBA_HIP(hipSetDevice(s->device));
    if (!s->ev_fence) LBA_HIP(hipEventCreateWithFlags(&s->ev_fence, hipEventDisableTiming));
    LBA_HIP(hipEventRecord(s->ev_fence, (hipStream_t)other));
    LBA_HIP(hipStreamWaitEvent(s->stream, s->ev_fence, 0));
    return ORBX_OK;
}

int lba_shard_set_async_reduce(lba_shard* s, int on)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    s->sync_after_reduce = on == 0 && !s->lambda_in_reduce;
    return ORBX_OK;
}

// Lets the caller own the reduce buffer (e.g. a torch.float64 CUDA tensor that torch.distributed all-reduces in place).
int lba_shard_set_reduce_buffer(lba_shard* s, double* device_buffer)
{
    if (!s || !device_buffer) return fail(ORBX_ERR_ARG, "NULL argument");
    s->reduce = device_buffer;
    return ORBX_OK;
}

static int read_scalars(lba_shard* s)
{
    // k_reduce wrote the scalars into the host-mapped buffer and then published its sequence number: poll for it (a trial is
    // a few hundred microseconds of kernels); after 20 ms fall back to a stream synchronisation, which also surfaces faults
    const volatile unsigned long long* flag = (const volatile unsigned long long*)(s->h_scal + 8);
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (*flag != s->seq) {
        if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
            LBA_HIP(hipStreamSynchronize(s->stream));
            if (*flag != s->seq) return fail(ORBX_ERR_INTERNAL, "reduction results did not arrive");
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return ORBX_OK;
}

// Optional: the lambda the first trial after the NEXT lba_shard_linearize will use (known from the second iteration on, or with a
// user lambda).  The linearisation then also performs the landmark side of the Schur complement, saving a dependent launch.
int lba_shard_hint_lambda(lba_shard* s, double lambda)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    s->hint_lambda = lambda;
    return ORBX_OK;
}

// computeActiveErrors + activeRobustChi2 + buildSystem on the accepted state
int lba_shard_linearize(lba_shard* s, double* chi2_local, double* max_diag_poses_local, double* max_diag_landmarks_local)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    LBA_HIP(hipSetDevice(s->device));
    const lba::Dev& d = s->d;
    const double* P = s->poses[s->cur];
    const double* X = s->pts[s->cur];
    // After an accepted trial the edge errors and their robust chi2 of the (new) estimate are already on the device
    // (lba_shard_finish computed them for the trial state): computeActiveErrors would reproduce them bit for bit, so only
    // the quadratic forms are rebuilt, with no host synchronisation.  The diagonal maxima are only refreshed on the
    // synchronising path (they are needed for lambda initialisation at the first iteration only).
    const bool reuse = s->err_current;
    s->mark(lba::kStageLinearize);
    if (!reuse && d.nE > 0) hipLaunchKernelGGL(lba::k_errors, dim3((d.nE + 255) / 256), dim3(256), 0, s->stream, d, P, X);
    // one launch for both sides; with a lambda hint (lba_shard_hint_lambda) the landmark workgroups also do their part of the Schur complement
    const double hint = s->hint_lambda;
    s->hint_lambda = -1.0;
    if (d.nP + d.nL > 0) hipLaunchKernelGGL(lba::k_lin_all, dim3(d.nP + (d.nL + 31) / 32), dim3(256), 0, s->stream, d, P, X, hint);
    s->schur_lambda = hint;
    s->mark(lba::kStageIdle);
    if (!reuse) {
        s->mark(lba::kStageReduce);
        hipLaunchKernelGGL(lba::k_reduce, dim3(1), dim3(1024), 0, s->stream, d, 0, s->d_hmap, ++s->seq);
        s->mark(lba::kStageIdle);
        LBA_HIP(hipGetLastError());
        int r = read_scalars(s);
        if (r) return r;
        s->chi_current = s->h_scal[0];
        s->mdp_cached = s->h_scal[1];
        s->mdl_cached = s->h_scal[2];
    } else {
        LBA_HIP(hipGetLastError());
    }
    s->err_valid = true;
    s->err_current = true;
    if (chi2_local) *chi2_local = s->chi_current;
    if (max_diag_poses_local) *max_diag_poses_local = s->mdp_cached;
    if (max_diag_landmarks_local) *max_diag_landmarks_local = s->mdl_cached;
    return ORBX_OK;
}

// partial Schur complement of this shard's landmarks into the reduce buffer (lambda enters through Hll only)
int lba_shard_reduce(lba_shard* s, double lambda)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    LBA_HIP(hipSetDevice(s->device));
    const lba::Dev& d = s->d;
    s->mark(lba::kStageSchur);
    if (d.nL > 0 && !(s->schur_lambda >= 0.0 && s->schur_lambda == lambda))
        hipLaunchKernelGGL(lba::k_schur_landmarks, dim3((d.nL + 7) / 8), dim3(64), 0, s->stream, d, lambda);
    s->schur_lambda = -1.0;                 // W / Dinv / Z now belong to this lambda only until the next trial changes it
    if (d.nBlocks + d.nP > 0)
        hipLaunchKernelGGL(lba::k_schur_blocks, dim3(d.nBlocks + d.nP), dim3(lba::kSchurThreads), 0, s->stream, d, s->S(),
                           s->lambda_in_reduce ? lambda : 0.0, s->bs(), s->bpf(), s->diag());
    s->lambda_added = s->lambda_in_reduce;
    s->mark(lba::kStageIdle);
    LBA_HIP(hipGetLastError());
    if (s->sync_after_reduce) LBA_HIP(hipStreamSynchronize(s->stream));      // the caller hands the buffer to RCCL on another stream
    return ORBX_OK;
}

// (after the caller's all-reduce) S += lambda I, solve, back-substitute, trial update, errors of the trial state.
// returns 1 when the linear solve succeeded, 0 when the reduced system was not positive definite.
int lba_shard_finish(lba_shard* s, double lambda, double* chi2_local_new, double* scale_poses, double* scale_landmarks_local)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    LBA_HIP(hipSetDevice(s->device));
    const lba::Dev& d = s->d;
    const int n = d.n;
    const double* P = s->poses[s->cur];
    const double* X = s->pts[s->cur];
    double* Pn = s->poses[1 - s->cur];
    double* Xn = s->pts[1 - s->cur];
    s->mark(lba::kStageFactor);
    if (n > 0) {
        if (!s->lambda_added) hipLaunchKernelGGL(lba::k_add_lambda, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->S(), n, lambda);
        const bool fused = s->nblk <= lba::kFusedMaxBlocks;
        static const bool step_launches = std::getenv("ORBX_LBA_STEPS") != nullptr;       // A/B knob: round 2's launch per block column
        if (fused && !step_launches) {
            // the whole factorisation as one launch: a workgroup per lower-triangle tile, flags between them (k_chol_flow)
            hipLaunchKernelGGL(lba::k_chol_flow, dim3(s->nblk * (s->nblk + 1) / 2), dim3(256), lba::kStepLds, s->stream, s->S(), s->Lp, n, s->nblk, s->Linv, d.scal,
                               s->flow, ++s->flow_epoch);
        } else if (fused) {
            // one launch per block column: panel + trailing update + the next diagonal factorisation (k_chol_step)
            hipLaunchKernelGGL(lba::k_chol_diag, dim3(1), dim3(256), 0, s->stream, (const double*)s->S(), n, 0, std::min(lba::NB, n), s->Linv, d.scal);
            for (int K = 0; K + 1 < s->nblk; K++) {
                const int T = s->nblk - 1 - K;
                hipLaunchKernelGGL(lba::k_chol_step, dim3(T * (T + 1) / 2), dim3(256), lba::kStepLds, s->stream, s->S(), s->Lp, n, K, s->nblk, s->Linv, d.scal);
            }
        }
        for (int K = 0; K < s->nblk && !fused; K++) {
            const int k0 = K * lba::NB, nb = std::min(lba::NB, n - k0);
            const int rows_below = (n + 1) - k0 - nb;       // includes the right-hand-side row n (always >= 1)
            hipLaunchKernelGGL(lba::k_chol_diag, dim3(1), dim3(256), 0, s->stream, (const double*)s->S(), n, k0, nb, s->Linv, d.scal);
            hipLaunchKernelGGL(lba::k_chol_panel, dim3((rows_below + lba::kPanelRows - 1) / lba::kPanelRows), dim3(1024), 0, s->stream,
                               s->S(), n, n + 1, k0, nb, (const double*)s->Linv, (const double*)d.scal);
            if (k0 + nb < n) {
                const int t = (rows_below + 31) / 32;
                hipLaunchKernelGGL(lba::k_chol_update, dim3(t, t), dim3(256), 0, s->stream, s->S(), n, n + 1, k0, nb, (const double*)d.scal);
            }
        }
    }
    s->mark(lba::kStageSolve);
    {
        const bool fused = n > 0 && s->nblk <= lba::kFusedMaxBlocks;
        const size_t solve_lds = ((size_t)n + 64 + 16 * 64 + lba::NB * (lba::NB + 1)) * sizeof(double);
        if (fused || n == 0)
            hipLaunchKernelGGL(lba::k_chol_solve_update<true>, dim3(1), dim3(1024), solve_lds, s->stream, (const double*)s->Lp, n, (const double*)s->Linv,
                               (const double*)(s->Lp ? s->Lp + (size_t)n * n : nullptr), (const double*)s->bs(), (const double*)d.scal, 1, d, lambda, (const double*)s->bpf(), P, Pn);
        else
            hipLaunchKernelGGL(lba::k_chol_solve_update<false>, dim3(1), dim3(1024), solve_lds, s->stream, (const double*)s->S(), n, (const double*)s->Linv,
                               (const double*)s->bs(), (const double*)s->bs(), (const double*)d.scal, 0, d, lambda, (const double*)s->bpf(), P, Pn);
    }
    s->mark(lba::kStageUpdate);
    hipLaunchKernelGGL(lba::k_update_errors, dim3(std::max((d.nL + lba::kUpdLandmarks - 1) / lba::kUpdLandmarks, 1)), dim3(lba::kUpdThreads), 0, s->stream, d, lambda, X, (const double*)Pn, Xn, s->d_hmap, ++s->seq);
    s->mark(lba::kStageIdle);
    LBA_HIP(hipGetLastError());
    int r = read_scalars(s);
    if (r) return r;
    s->chi_trial = s->h_scal[0];
    s->err_current = false;         // the error buffer now belongs to the trial state
    if (chi2_local_new) *chi2_local_new = s->h_scal[0];
    if (scale_poses) *scale_poses = s->h_scal[3];
    if (scale_landmarks_local) *scale_landmarks_local = s->h_scal[4];
    return s->h_scal[5] != 0.0 ? 0 : 1;
}

int lba_shard_accept(lba_shard* s, int accept)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    if (accept) {       // discardTop(): the trial state becomes the estimate; its errors / chi2 are the current ones
        s->cur = 1 - s->cur;
        s->err_current = true;
        s->chi_current = s->chi_trial;
    }                   // pop(): keep the old estimate; the error buffer stays that of the rejected trial (as in g2o)
    return ORBX_OK;
}

int lba_shard_download(lba_shard* s, double* pose_q, double* pose_t, double* points, double* chi2_per_edge, uint8_t* depth_positive)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    LBA_HIP(hipSetDevice(s->device));
    const lba::Dev& d = s->d;
    if (d.nE > 0)
        hipLaunchKernelGGL(lba::k_epilogue, dim3((d.nE + 255) / 256), dim3(256), 0, s->stream, d, (const double*)s->poses[s->cur], (const double*)s->pts[s->cur], s->d_chi2, s->d_depth);
    LBA_HIP(hipGetLastError());
    std::vector<double> poses(7 * (size_t)d.nPoses);
    LBA_HIP(hipMemcpyAsync(poses.data(), s->poses[s->cur], poses.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (points && d.nL > 0) LBA_HIP(hipMemcpyAsync(points, s->pts[s->cur], 3 * (size_t)d.nL * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (chi2_per_edge && d.nE > 0) LBA_HIP(hipMemcpyAsync(chi2_per_edge, s->d_chi2, (size_t)d.nE * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (depth_positive && d.nE > 0) LBA_HIP(hipMemcpyAsync(depth_positive, s->d_depth, (size_t)d.nE, hipMemcpyDeviceToHost, s->stream));
    LBA_HIP(hipStreamSynchronize(s->stream));
    for (int i = 0; i < d.nPoses; i++) {
        if (pose_q) for (int k = 0; k < 4; k++) pose_q[4 * i + k] = poses[7 * (size_t)i + k];
        if (pose_t) for (int k = 0; k < 3; k++) pose_t[3 * i + k] = poses[7 * (size_t)i + 4 + k];
    }
    return ORBX_OK;
}

// ---- Levenberg-Marquardt driver over a shard: SparseOptimizer::optimize (sparse_optimizer.cpp:354-419) driving
// OptimizationAlgorithmLevenberg::solve (optimization_algorithm_levenberg.cpp:61-169).  With an all-reduce callback this is the
// landmark-sharded global BA of SURVEY.md 8(e) for a C / C++ host: every rank calls it on its own shard and the callback is
// one ncclAllReduce on the stream it is given; every decision input is all-reduced, so all ranks
// walk the same path.  Without a callback (world size 1) it is lba_solve's loop. ----
int lba_shard_optimize(lba_shard* s, lba_allreduce_fn allreduce, void* user, int world_size, int max_iters, double lambda_init,
                       const volatile uint8_t* stop_flag, LbaStats* stats_out)
{
    if (!s) return fail(ORBX_ERR_ARG, "NULL shard");
    if (world_size < 1 || (world_size > 1 && !allreduce)) return fail(ORBX_ERR_ARG, "world size %d needs an all-reduce callback", world_size);
    LBA_HIP(hipSetDevice(s->device));
    const bool dist = allreduce != nullptr;         // a callback is always used, also by a communicator of one rank
    if (dist) {
        // the all-reduce sits between reduce() and finish() on the shard's own stream: lambda is added afterwards, no host wait
        s->lambda_in_reduce = false;
        s->sync_after_reduce = false;
        if (!s->d_coll) {
            int r0 = s->dalloc(&s->d_coll, 16);
            if (r0) return r0;
            if (hipHostMalloc((void**)&s->h_coll, 16 * sizeof(double), hipHostMallocDefault) != hipSuccess) return fail(ORBX_ERR_HIP, "hipHostMalloc failed");
        }
    } else {
        s->lambda_in_reduce = true;
        s->sync_after_reduce = false;
    }
    int r = ORBX_OK;
    // scalar pack through the caller's collective: host -> device scratch -> all-reduce on the shard's stream -> host
    auto reduce_scalars = [&](double* v, int n, int op) -> int {
        if (!dist) return ORBX_OK;
        for (int i = 0; i < n; i++) s->h_coll[i] = v[i];
        LBA_HIP(hipMemcpyAsync(s->d_coll, s->h_coll, n * sizeof(double), hipMemcpyHostToDevice, s->stream));
        if (allreduce(user, s->d_coll, n, op, (void*)s->stream)) return fail(ORBX_ERR_INTERNAL, "the all-reduce callback failed");
        LBA_HIP(hipMemcpyAsync(s->h_coll, s->d_coll, n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
        LBA_HIP(hipStreamSynchronize(s->stream));
        for (int i = 0; i < n; i++) v[i] = s->h_coll[i];
        return ORBX_OK;
    };
    auto reduce_system = [&]() -> int {
        if (!dist) return ORBX_OK;
        if (allreduce(user, s->reduce, s->reduce_len, LBA_REDUCE_SUM, (void*)s->stream)) return fail(ORBX_ERR_INTERNAL, "the all-reduce callback failed");
        return ORBX_OK;
    };
    // every rank must take the same decision: the flag is MAX-reduced
    auto terminate = [&](bool* stop) -> int {
        double f = (stop_flag && *stop_flag) ? 1.0 : 0.0;
        const int rr = reduce_scalars(&f, 1, LBA_REDUCE_MAX);
        *stop = f > 0.0;
        return rr;
    };
    LbaStats st;
    std::memset(&st, 0, sizeof(st));
    double lambda = -1, ni = 2;
    int nBad = 0;
    for (int it = 0; it < max_iters; it++) {
        bool stop = false;
        if ((r = terminate(&stop))) break;
        if (stop) { st.stop_reason = 3; break; }
        double currentChi = 0, mdp = 0, mdl = 0;
        if (it > 0) lba_shard_hint_lambda(s, lambda);
        else if (lambda_init > 0) lba_shard_hint_lambda(s, lambda_init);
        if ((r = lba_shard_linearize(s, &currentChi, &mdp, &mdl))) break;
        if ((r = reduce_scalars(&currentChi, 1, LBA_REDUCE_SUM))) break;
        const double iniChi = currentChi;
        if (it == 0) {
            st.chi2_initial = currentChi;
            if (lambda_init > 0) lambda = lambda_init;
            else {
                if (dist) {
                    // the pose diagonals are partial sums over the shards: one lambda-free exchange, then the maximum of the summed
                    // diagonal section; the landmark maximum is MAX-reduced (computeLambdaInit, levenberg.cpp:171-185)
                    if ((r = lba_shard_reduce(s, 0.0)) || (r = reduce_system())) break;
                    std::vector<double> dg((size_t)std::max(s->d.n, 1), 0.0);
                    if (s->d.n > 0) LBA_HIP(hipMemcpyAsync(dg.data(), s->diag(), (size_t)s->d.n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
                    LBA_HIP(hipStreamSynchronize(s->stream));
                    mdp = 0;
                    for (int k = 0; k < s->d.n; k++) mdp = std::max(mdp, std::fabs(dg[k]));
                    if ((r = reduce_scalars(&mdl, 1, LBA_REDUCE_MAX))) break;
                }
                lambda = 1e-5 * std::max(mdp, mdl);
            }
            ni = 2; nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        bool stopped = false;
        do {
            if ((r = lba_shard_reduce(s, lambda)) || (r = reduce_system())) break;
            double tempChi = 0, sp = 0, sl = 0;
            const int ok2 = lba_shard_finish(s, lambda, &tempChi, &sp, &sl);
            if (ok2 < 0) { r = ok2; break; }
            double pack[3] = {tempChi, sl, (double)ok2};
            if ((r = reduce_scalars(pack, 3, LBA_REDUCE_SUM))) break;
            tempChi = pack[0]; sl = pack[1];
            if (pack[2] < (dist ? world_size : 1) - 0.5) tempChi = std::numeric_limits<double>::max();
            rho = currentChi - tempChi;
            double scale = sp + sl;
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
                lba_shard_accept(s, 1);
            } else {
                lambda *= ni;
                ni *= 2;
                lba_shard_accept(s, 0);
            }
            qmax++;
            st.trials++;
            if ((r = terminate(&stopped))) break;
        } while (rho < 0 && qmax < 10 && !stopped);
        if (r) break;
        st.iterations++;
        if (it < 16) st.chi2_trace[it] = currentChi;
        st.chi2_final = currentChi;
        if (qmax == 10 || rho == 0) { st.stop_reason = 1; break; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) { st.stop_reason = 2; break; }
    }
    st.lambda = lambda;
    if (stats_out) *stats_out = st;
    return r;
}

// ---- single-GPU driver: optimizer.initializeOptimization(); optimizer.optimize(max_iters) ----
int lba_create(int device, lba_solver** out)
{
    if (!out) return fail(ORBX_ERR_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ORBX_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(ORBX_ERR_ARG, "device %d out of range", device);
    LBA_HIP(hipSetDevice(device));
    lba_solver* s = new lba_solver();
    s->device = device;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess || hipHostMalloc((void**)&s->h_scal, 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        delete s;
        return fail(ORBX_ERR_HIP, "stream / pinned buffer creation failed");
    }
    std::memset(s->h_scal, 0, 16 * sizeof(double));
    *out = s;
    return ORBX_OK;
}

void lba_destroy(lba_solver* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    if (s->arena) (void)hipFree(s->arena);
    if (s->stage) (void)hipHostFree(s->stage);
    if (s->h_scal) (void)hipHostFree(s->h_scal);
    delete s;
}

int lba_solve(lba_solver* sv, const LbaProblem* problem, const volatile uint8_t* stop_flag, int max_iters, double lambda_init,
              double* pose_q_out, double* pose_t_out, double* points_out,
              double* chi2_per_edge, uint8_t* depth_positive, LbaStats* stats_out)
{
    if (!sv) return fail(ORBX_ERR_ARG, "NULL solver");
    lba_shard* s = nullptr;
    static const bool timing = std::getenv("ORBX_LBA_TIMING") != nullptr;      // phase times of the call on stderr (tools/lba_prof.py)
    const auto t_start = std::chrono::steady_clock::now();
    int r = shard_create_impl(sv->device, problem, &s, sv);
    if (r) return r;
    const auto t_created = std::chrono::steady_clock::now();
    s->sync_after_reduce = false;
    s->lambda_in_reduce = true;
    const size_t wanted = s->bytes_wanted, wanted_stage = s->upload_bytes;
    LbaStats st;
    r = lba_shard_optimize(s, nullptr, nullptr, 1, max_iters, lambda_init, stop_flag, &st);
    const auto t_solved = std::chrono::steady_clock::now();
    if (!r) r = lba_shard_download(s, pose_q_out, pose_t_out, points_out, chi2_per_edge, depth_positive);
    const auto t_down = std::chrono::steady_clock::now();
    lba_shard_destroy(s);
    if (timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[lba_solve] structure + upload %.3f ms, %d iterations / %d trials %.3f ms, epilogue + download %.3f ms, destroy %.3f ms\n",
                     ms(t_start, t_created), st.iterations, st.trials, ms(t_created, t_solved), ms(t_solved, t_down), ms(t_down, std::chrono::steady_clock::now()));
    }
    solver_grow(sv, wanted, wanted_stage);
    if (stats_out) *stats_out = st;
    return r;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// lba_solve_batch: W independent windows (one map per client session, SURVEY.md 8(e): "independent maps shard round-robin") through
// ONE sequence of launches per Levenberg round, grid.y = window.  Every window walks exactly the path lba_solve would walk for it
// (same kernel bodies, same order of operations -> bit-identical results); the host keeps one LM state machine per window and a
// round is: [linearise the windows that start an iteration] + [one trial of every window that is not finished].
// ---------------------------------------------------------------------------------------------------------------
struct lba_batch {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<lba_solver*> slots;     // arena + pinned staging per window slot (they grow to the windows they have seen)
    lba::BWin* d_wins = nullptr;
    double* h_scal = nullptr;           // host-mapped, coherent: 16 doubles per slot
    double last_device_ms = 0.0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint8_t* h_out = nullptr;           // pinned staging of the results of all windows (one synchronisation per call)
    size_t h_out_cap = 0;
};

extern "C" {

int lba_batch_create(int device, lba_batch** out)
{
    if (!out) return fail(ORBX_ERR_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ORBX_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(ORBX_ERR_ARG, "device %d out of range", device);
    LBA_HIP(hipSetDevice(device));
    lba_batch* b = new lba_batch();
    b->device = device;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void**)&b->h_scal, lba::kMaxBatch * 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipMalloc((void**)&b->d_wins, lba::kMaxBatch * sizeof(lba::BWin)) != hipSuccess ||
        hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess) {
        lba_batch_destroy(b);
        return fail(ORBX_ERR_HIP, "batch handle creation failed");
    }
    std::memset(b->h_scal, 0, lba::kMaxBatch * 16 * sizeof(double));
    *out = b;
    return ORBX_OK;
}

void lba_batch_destroy(lba_batch* b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (lba_solver* sv : b->slots) {
        if (sv->arena) (void)hipFree(sv->arena);
        if (sv->stage) (void)hipHostFree(sv->stage);
        delete sv;
    }
    if (b->d_wins) (void)hipFree(b->d_wins);
    if (b->h_scal) (void)hipHostFree(b->h_scal);
    if (b->h_out) (void)hipHostFree(b->h_out);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

double lba_batch_last_device_ms(const lba_batch* b) { return b ? b->last_device_ms : 0.0; }

int lba_solve_batch(lba_batch* b, const LbaProblem* problems, const LbaOutputs* outputs, int n_windows,
                    const volatile uint8_t* const* stop_flags, int max_iters, double lambda_init, LbaStats* stats_out)
{
    if (!b || !problems || n_windows < 0) return fail(ORBX_ERR_ARG, "NULL argument");
    if (n_windows > lba::kMaxBatch) return fail(ORBX_ERR_CAPACITY, "at most %d windows per call", lba::kMaxBatch);
    if (n_windows == 0) return ORBX_OK;
    LBA_HIP(hipSetDevice(b->device));
    const int W = n_windows;
    while ((int)b->slots.size() < W) {
        lba_solver* sv = new lba_solver();
        sv->device = b->device; sv->stream = b->stream; sv->h_scal = b->h_scal + 16 * b->slots.size();
        b->slots.push_back(sv);
    }
    std::vector<lba_shard*> sh((size_t)W, nullptr);
    std::vector<size_t> wanted((size_t)W, 0), wanted_stage((size_t)W, 0);
    int r = ORBX_OK;
    auto cleanup = [&]() {
        for (int i = 0; i < W; i++) {
            if (sh[i]) lba_shard_destroy(sh[i]);
            solver_grow(b->slots[i], wanted[i], wanted_stage[i]);
        }
    };
    std::vector<lba::BWin> hw((size_t)W);
    int max_lin = 1, max_e = 1, max_lm = 1, max_sb = 1, max_nblk = 1, max_upd = 1, max_n = 0;
    static const bool timing = std::getenv("ORBX_LBA_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    {
        // structure build (CSR lists, pair lists) + upload of every window: host work, spread over threads (a window is ~0.3 ms)
        const int n_thr = std::max(1, std::min({W, (int)std::thread::hardware_concurrency(), 16}));
        std::vector<int> rcs((size_t)W, ORBX_OK);
        std::atomic<int> next(0);
        auto worker = [&]() {
            (void)hipSetDevice(b->device);
            for (int i = next.fetch_add(1); i < W; i = next.fetch_add(1)) rcs[i] = shard_create_impl(b->device, &problems[i], &sh[i], b->slots[i]);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < n_thr; t++) th.emplace_back(worker);
        worker();
        for (auto& t : th) t.join();
        for (int i = 0; i < W; i++) if (rcs[i] && !r) r = fail(rcs[i], "window %d could not be set up (code %d; the worker thread holds the detailed message)", i, rcs[i]);
    }
    const auto t_created = std::chrono::steady_clock::now();
    for (int i = 0; i < W && !r; i++) {
        lba_shard* s = sh[i];
        wanted[i] = s->bytes_wanted; wanted_stage[i] = s->upload_bytes;
        if (s->nblk > lba::kFusedMaxBlocks) { r = fail(ORBX_ERR_CAPACITY, "window %d has %d reduced unknowns: the batched path takes at most %d (use lba_solve)", i, s->d.n, lba::kFusedMaxBlocks * lba::NB); break; }
        lba::BWin& w = hw[i];
        w.d = s->d; w.poses[0] = s->poses[0]; w.poses[1] = s->poses[1]; w.pts[0] = s->pts[0]; w.pts[1] = s->pts[1];
        w.S = s->S(); w.bs = s->bs(); w.bpf = s->bpf(); w.diag = s->diag(); w.Lp = s->Lp; w.Linv = s->Linv; w.hmap = s->d_hmap; w.flow = s->flow; w.nblk = s->nblk;
        const lba::Dev& d = s->d;
        max_lin = std::max(max_lin, d.nP + (d.nL + 31) / 32); max_e = std::max(max_e, (d.nE + 255) / 256); max_lm = std::max(max_lm, (d.nL + 7) / 8);
        max_sb = std::max(max_sb, d.nBlocks + d.nP); max_nblk = std::max(max_nblk, s->nblk); max_upd = std::max(max_upd, (d.nL + lba::kUpdLandmarks - 1) / lba::kUpdLandmarks);
        max_n = std::max(max_n, d.n);
    }
    if (r) { cleanup(); return r; }
    const size_t solve_lds = ((size_t)max_n + 64 + 16 * 64 + lba::NB * (lba::NB + 1)) * sizeof(double);
    if (hipFuncSetAttribute((const void*)lba::k_chol_step_b, hipFuncAttributeMaxDynamicSharedMemorySize, lba::kStepLds) != hipSuccess ||
        hipFuncSetAttribute((const void*)lba::k_chol_flow_b, hipFuncAttributeMaxDynamicSharedMemorySize, lba::kStepLds) != hipSuccess ||
        hipFuncSetAttribute((const void*)lba::k_chol_solve_update_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(solve_lds, (size_t)65536)) != hipSuccess ||
        hipMemcpyAsync(b->d_wins, hw.data(), (size_t)W * sizeof(lba::BWin), hipMemcpyHostToDevice, b->stream) != hipSuccess) {
        cleanup();
        return fail(ORBX_ERR_HIP, "batch setup failed");
    }
    (void)hipEventRecord(b->ev0, b->stream);

    // ---- one Levenberg state machine per window (the control flow of lba_shard_optimize, cut at the points where it waits for the device) ----
    enum Phase { kStartIter, kAfterLin, kTrials, kDone };
    struct WS { Phase ph = kStartIter; double lambda = -1, ni = 2, cur_chi = 0, ini_chi = 0, rho = 0, trial_lambda = 0; int nBad = 0, it = 0, qmax = 0, cur = 0; bool err_current = false, first_trial = false; double hint = -1; LbaStats st; };
    std::vector<WS> ws((size_t)W);
    for (auto& x : ws) std::memset(&x.st, 0, sizeof(x.st));
    auto stopped = [&](int i) { return stop_flags && stop_flags[i] && *stop_flags[i]; };
    std::vector<unsigned long long> seq((size_t)W);
    for (int i = 0; i < W; i++) seq[i] = sh[i]->seq;
    lba::BDynAll dyn;
    std::memset(&dyn, 0, sizeof(dyn));
    auto finish_iteration = [&](int i) {        // after the trial loop of an iteration (levenberg.cpp:150-169, sparse_optimizer.cpp:395-414)
        WS& x = ws[i];
        x.st.iterations++;
        if (x.it < 16) x.st.chi2_trace[x.it] = x.cur_chi;
        x.st.chi2_final = x.cur_chi;
        if (x.qmax == 10 || x.rho == 0) { x.st.stop_reason = 1; x.ph = kDone; return; }
        if ((x.ini_chi - x.cur_chi) * 1e3 < x.ini_chi) x.nBad++; else x.nBad = 0;
        if (x.nBad >= 3) { x.st.stop_reason = 2; x.ph = kDone; return; }
        x.it++;
        x.ph = kStartIter;
    };
    for (;;) {
        bool any_lin = false, any_err = false, any_trial = false, any_lm = false;
        for (int i = 0; i < W; i++) {
            WS& x = ws[i];
            lba::BDyn& y = dyn.w[i];
            y.flags = 0;
            if (x.ph == kStartIter) {
                if (x.it >= max_iters) { x.ph = kDone; }
                else if (stopped(i)) { x.st.stop_reason = 3; x.ph = kDone; }
                else {
                    x.hint = x.it > 0 ? x.lambda : (lambda_init > 0 ? lambda_init : -1.0);
                    y.flags |= lba::kBwLin;
                    if (!x.err_current) { y.flags |= lba::kBwErrors | lba::kBwReduce0; x.ph = kAfterLin; }       // the host needs chi2 / the diagonals first
                    else { x.ini_chi = x.cur_chi; x.rho = 0; x.qmax = 0; x.first_trial = true; x.ph = kTrials; }
                }
            }
            if (x.ph == kTrials) {
                y.flags |= lba::kBwTrial;
                if (!(x.first_trial && x.hint >= 0.0 && x.hint == x.lambda)) y.flags |= lba::kBwSchurLm;
                x.trial_lambda = x.lambda;
            }
            y.lambda = x.lambda; y.hint = x.hint; y.cur = x.cur;
            if (y.flags & (lba::kBwReduce0 | lba::kBwTrial)) y.seq = ++seq[i];
            any_lin |= (y.flags & lba::kBwLin) != 0; any_err |= (y.flags & lba::kBwErrors) != 0;
            any_trial |= (y.flags & lba::kBwTrial) != 0; any_lm |= (y.flags & lba::kBwSchurLm) != 0;
        }
        if (!any_lin && !any_trial) break;
        hipStream_t st = b->stream;
        if (any_err) hipLaunchKernelGGL(lba::k_errors_b, dim3(max_e, W), dim3(256), 0, st, (const lba::BWin*)b->d_wins, dyn, 0);
        if (any_lin) hipLaunchKernelGGL(lba::k_lin_all_b, dim3(max_lin, W), dim3(256), 0, st, (const lba::BWin*)b->d_wins, dyn);
        if (any_err) hipLaunchKernelGGL(lba::k_reduce_b, dim3(1, W), dim3(1024), 0, st, (const lba::BWin*)b->d_wins, dyn, 0);
        if (any_trial) {
            if (any_lm) hipLaunchKernelGGL(lba::k_schur_landmarks_b, dim3(max_lm, W), dim3(64), 0, st, (const lba::BWin*)b->d_wins, dyn);
            hipLaunchKernelGGL(lba::k_schur_blocks_b, dim3(max_sb, W), dim3(lba::kSchurThreads), 0, st, (const lba::BWin*)b->d_wins, dyn);
            // the factorisation: one launch (a workgroup per tile and window, flags between them) while every workgroup of the launch
            // can be resident at once (it waits on others), else a launch per block column
            static const bool step_launches = std::getenv("ORBX_LBA_STEPS") != nullptr;
            const int tiles = max_nblk * (max_nblk + 1) / 2;
            if (!step_launches && tiles * W <= 240) {       // (100 KB of LDS per workgroup: one per CU)
                hipLaunchKernelGGL(lba::k_chol_flow_b, dim3(tiles, W), dim3(256), lba::kStepLds, st, (const lba::BWin*)b->d_wins, dyn);
            } else {
                hipLaunchKernelGGL(lba::k_chol_diag_b, dim3(1, W), dim3(256), 0, st, (const lba::BWin*)b->d_wins, dyn);
                for (int K = 0; K + 1 < max_nblk; K++) {
                    const int T = max_nblk - 1 - K;
                    hipLaunchKernelGGL(lba::k_chol_step_b, dim3(T * (T + 1) / 2, W), dim3(256), lba::kStepLds, st, (const lba::BWin*)b->d_wins, dyn, K);
                }
            }
            hipLaunchKernelGGL(lba::k_chol_solve_update_b, dim3(1, W), dim3(1024), solve_lds, st, (const lba::BWin*)b->d_wins, dyn);
            hipLaunchKernelGGL(lba::k_update_errors_b, dim3(max_upd, W), dim3(lba::kUpdThreads), 0, st, (const lba::BWin*)b->d_wins, dyn);
        }
        if (hipGetLastError() != hipSuccess) { r = fail(ORBX_ERR_HIP, "batched launch failed"); break; }
        // results of the round: every window that ran a reduction publishes its sequence number last
        for (int i = 0; i < W && !r; i++) {
            const int f = dyn.w[i].flags;
            if (!(f & (lba::kBwReduce0 | lba::kBwTrial))) continue;
            sh[i]->seq = seq[i];
            r = read_scalars(sh[i]);
        }
        if (r) break;
        for (int i = 0; i < W; i++) {
            WS& x = ws[i];
            const int f = dyn.w[i].flags;
            const double* h = sh[i]->h_scal;
            if (f & lba::kBwReduce0) {      // linearised without a trial: chi2 and the diagonal maxima are in
                x.cur_chi = h[0];
                x.err_current = true;
                x.ini_chi = x.cur_chi;
                if (x.it == 0) {
                    x.st.chi2_initial = x.cur_chi;
                    x.lambda = lambda_init > 0 ? lambda_init : 1e-5 * std::max(h[1], h[2]);      // computeLambdaInit (levenberg.cpp:171-185)
                    x.ni = 2; x.nBad = 0;
                }
                x.rho = 0; x.qmax = 0; x.first_trial = true;
                x.ph = kTrials;
                continue;
            }
            if (!(f & lba::kBwTrial)) continue;
            double tempChi = h[0];
            const double sp = h[3], sl = h[4];
            if (h[5] != 0.0) tempChi = std::numeric_limits<double>::max();       // the reduced system was not positive definite
            x.first_trial = false;
            x.err_current = false;
            x.rho = x.cur_chi - tempChi;
            double scale = sp + sl;
            scale += 1e-3;
            x.rho /= scale;
            if (x.rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * x.rho - 1), 3);
                alpha = std::min(alpha, 2. / 3.);
                x.lambda *= std::max(1. / 3., alpha);
                x.ni = 2;
                x.cur_chi = tempChi;
                x.cur = 1 - x.cur;              // discardTop(): the trial state becomes the estimate, its errors are the current ones
                x.err_current = true;
            } else {
                x.lambda *= x.ni;
                x.ni *= 2;
            }
            x.qmax++;
            x.st.trials++;
            const bool stop_now = stopped(i);
            if (!(x.rho < 0 && x.qmax < 10 && !stop_now)) finish_iteration(i);
        }
    }
    (void)hipEventRecord(b->ev1, b->stream);
    const auto t_solved = std::chrono::steady_clock::now();
#define BTRY(expr) do { if (!r && (expr) != hipSuccess) r = fail(ORBX_ERR_HIP, "%s failed", #expr); } while (0)
    // results of all windows: epilogue kernels and copies into ONE pinned buffer, one synchronisation, then the scatter
    {
        auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
        std::vector<size_t> off((size_t)W + 1, 0);
        for (int i = 0; i < W; i++) {
            const lba::Dev& d = sh[i]->d;
            off[i + 1] = off[i] + al(7 * (size_t)d.nPoses * 8) + al(3 * (size_t)d.nL * 8) + al((size_t)d.nE * 8) + al((size_t)d.nE);
        }
        if (outputs && off[W] > b->h_out_cap) {
            if (b->h_out) (void)hipHostFree(b->h_out);
            b->h_out = nullptr; b->h_out_cap = 0;
            const size_t cap = off[W] + off[W] / 4 + 4096;
            if (hipHostMalloc((void**)&b->h_out, cap) == hipSuccess) b->h_out_cap = cap;
            else r = fail(ORBX_ERR_HIP, "pinned result buffer allocation failed");
        }
        for (int i = 0; i < W && !r; i++) {
            lba_shard* s = sh[i];
            s->cur = ws[i].cur;
            ws[i].st.lambda = ws[i].lambda;
            if (stats_out) stats_out[i] = ws[i].st;
            if (!outputs) continue;
            const lba::Dev& d = s->d;
            uint8_t* o = b->h_out + off[i];
            if (d.nE > 0 && (outputs[i].chi2_per_edge || outputs[i].depth_positive))
                hipLaunchKernelGGL(lba::k_epilogue, dim3((d.nE + 255) / 256), dim3(256), 0, b->stream, d, (const double*)s->poses[s->cur], (const double*)s->pts[s->cur], s->d_chi2, s->d_depth);
            BTRY(hipMemcpyAsync(o, s->poses[s->cur], 7 * (size_t)d.nPoses * 8, hipMemcpyDeviceToHost, b->stream));
            o += al(7 * (size_t)d.nPoses * 8);
            if (outputs[i].points && d.nL > 0) BTRY(hipMemcpyAsync(o, s->pts[s->cur], 3 * (size_t)d.nL * 8, hipMemcpyDeviceToHost, b->stream));
            o += al(3 * (size_t)d.nL * 8);
            if (outputs[i].chi2_per_edge && d.nE > 0) BTRY(hipMemcpyAsync(o, s->d_chi2, (size_t)d.nE * 8, hipMemcpyDeviceToHost, b->stream));
            o += al((size_t)d.nE * 8);
            if (outputs[i].depth_positive && d.nE > 0) BTRY(hipMemcpyAsync(o, s->d_depth, (size_t)d.nE, hipMemcpyDeviceToHost, b->stream));
        }
        if (!r) {
            BTRY(hipStreamSynchronize(b->stream));
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess) b->last_device_ms = ms;
        }
        for (int i = 0; i < W && !r && outputs; i++) {
            const lba::Dev& d = sh[i]->d;
            const uint8_t* o = b->h_out + off[i];
            const double* poses = (const double*)o;
            for (int k = 0; k < d.nPoses; k++) {
                if (outputs[i].pose_q) for (int c = 0; c < 4; c++) outputs[i].pose_q[4 * k + c] = poses[7 * (size_t)k + c];
                if (outputs[i].pose_t) for (int c = 0; c < 3; c++) outputs[i].pose_t[3 * k + c] = poses[7 * (size_t)k + 4 + c];
            }
            o += al(7 * (size_t)d.nPoses * 8);
            if (outputs[i].points && d.nL > 0) std::memcpy(outputs[i].points, o, 3 * (size_t)d.nL * 8);
            o += al(3 * (size_t)d.nL * 8);
            if (outputs[i].chi2_per_edge && d.nE > 0) std::memcpy(outputs[i].chi2_per_edge, o, (size_t)d.nE * 8);
            o += al((size_t)d.nE * 8);
            if (outputs[i].depth_positive && d.nE > 0) std::memcpy(outputs[i].depth_positive, o, (size_t)d.nE);
        }
    }
#undef BTRY
    const auto t_down = std::chrono::steady_clock::now();
    cleanup();
    if (timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
        std::fprintf(stderr, "[lba_solve_batch] %d windows: structure + upload %.3f ms, Levenberg rounds %.3f ms, epilogue + download %.3f ms, destroy %.3f ms\n",
                     W, ms(t_start, t_created), ms(t_created, t_solved), ms(t_solved, t_down), ms(t_down, std::chrono::steady_clock::now()));
    }
    return r;
}

}  // extern "C"

#include "inertial_solver.inc"
