// pose_solver.hip -- gfx950 kernel + C ABI for Optimizer::PoseOptimization (reference src/Optimizer.cc:814-1115):
// motion-only bundle adjustment of one frame.  SURVEY.md 8(f) rank 1 ("next" row): it runs every frame right after
// matching (src/Tracking.cc:2889,3053,3115).
//
// MI355X mapping: the problem is tiny (one 6-dof pose, a few hundred unary edges, a 6x6 system) and strictly sequential in
// its control flow (4 rounds x <=10 LM iterations x <=10 trials), so a launch per step would be pure latency.  Instead ONE
// 256-thread workgroup runs the whole optimisation of a frame on the device -- edges in registers/L2, ordered block
// reductions for chi2 / H / b, the 6x6 LDL^T, SE3 exp and all Levenberg and outlier bookkeeping -- and a batch of frames
// is one launch with one workgroup per frame (frames are independent: they shard over workgroups and GPUs alike).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/orbslam3_hip.h"
#include "se3_device.h"

namespace orbx {
int fail(int code, const char* fmt, ...);
}
using orbx::fail;

#define POSE_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(ORBX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace poseopt {

using namespace se3;

struct ProblemDev {
    double q[4], t[3];
    int32_t n;
    const double* Xw; const double* obs; const double* w; const uint8_t* stereo;
    double fx, fy, cx, cy, bf, huber_mono, huber_stereo;
    double* err;            // [3n] scratch: edge._error as last computed
    uint8_t* outlier;       // [n] out
    uint8_t* active;        // [n] scratch: level 0
    PoseResult* result;
};

__device__ __forceinline__ double block_sum(double v, double* s_red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[wave] = v;
    __syncthreads();
    return ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

// 6x6 LDL^T solve (LinearSolverDense, solvers/linear_solver_dense.h:55-110) of (H + lambda I) x = b; Hu = packed upper
// triangle (row a, column c >= a at a*6 - a(a-1)/2 + c - a).  Returns false for a non-positive pivot.  Everything is
// indexed at compile time (registers only); one reciprocal per pivot instead of a division per entry.
__device__ __forceinline__ bool solve6(const double* Hu, double lambda, const double* b, double* x)
{
    double A[36], D[6], iD[6];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = r; c < 6; c++) { const double v = Hu[r * 6 - (r * (r - 1)) / 2 + (c - r)]; A[r * 6 + c] = v; A[c * 6 + r] = v; }
#pragma unroll
    for (int i = 0; i < 6; i++) A[i * 7] += lambda;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[j * 6 + k] * A[j * 6 + k] * D[k];
        ok = ok && (d > 0.0) && isfinite(d);
        D[j] = d;
        iD[j] = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double sv = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) sv -= A[i * 6 + k] * A[j * 6 + k] * D[k];
            A[i * 6 + j] = sv * iD[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double sv = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) sv -= A[i * 6 + k] * x[k];
        x[i] = sv;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] *= iD[i];
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double sv = x[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) sv -= A[k * 6 + i] * x[k];
        x[i] = sv;
    }
    return ok;
}

// one edge in registers
struct Edge {
    double X[3], o[3], w;
    int st;
};

template <bool STEREO>
__device__ __forceinline__ Edge load_edge(const ProblemDev& P, int e)
{
    Edge d;
    const double* X = P.Xw + 3 * (size_t)e;
    const double* o = P.obs + 3 * (size_t)e;
    d.X[0] = X[0]; d.X[1] = X[1]; d.X[2] = X[2];
    d.o[0] = o[0]; d.o[1] = o[1]; d.o[2] = STEREO ? o[2] : 0.0;
    d.w = P.w[e];
    d.st = STEREO ? (int)P.stereo[e] : 0;
    return d;
}

// EdgeSE3ProjectXYZOnlyPose::computeError (include/OptimizableTypes.h:46-50) and
// EdgeStereoSE3ProjectXYZOnlyPose::computeError (types_six_dof_expmap.h:203-207, cam_project .cpp:339-346)
template <bool STEREO>
__device__ __forceinline__ void edge_eval(const ProblemDev& P, const double* T, const Edge& d, double* Xc, double* r)
{
    pose_map(T, d.X, Xc);
    if (!STEREO || !d.st) {
        r[0] = d.o[0] - (P.fx * Xc[0] / Xc[2] + P.cx);
        r[1] = d.o[1] - (P.fy * Xc[1] / Xc[2] + P.cy);
        r[2] = 0;
    } else {
        const float invz = (float)(1.0 / Xc[2]);             // 1.0f/double rounded to float (types_six_dof_expmap.cpp:340)
        const double u = Xc[0] * (double)invz * P.fx + P.cx;
        const double v = Xc[1] * (double)invz * P.fy + P.cy;
        r[0] = d.o[0] - u; r[1] = d.o[1] - v; r[2] = d.o[2] - (u - P.bf * (double)invz);
    }
}

template <bool STEREO>
__device__ __forceinline__ double edge_chi2(const Edge& d, const double* r)
{
    double c = r[0] * (d.w * r[0]) + r[1] * (d.w * r[1]);
    if (STEREO && d.st) c += r[2] * (d.w * r[2]);
    return c;
}

struct Robust {
    bool on;
    double delta_m, delta_s, dsq_m, dsq_s;
};

// computeError + robustify + linearizeOplus + constructQuadraticForm of one edge, accumulated into acc[28]
template <bool STEREO>
__device__ __forceinline__ void edge_build(const ProblemDev& P, const double* T, const Edge& d, const Robust& rb, double* r, double* acc)
{
    constexpr int D = STEREO ? 3 : 2;
    double Xc[3], J[D * 6];
    edge_eval<STEREO>(P, T, d, Xc, r);
    const double chi = edge_chi2<STEREO>(d, r);
    const double delta = (STEREO && d.st) ? rb.delta_s : rb.delta_m, dsq = (STEREO && d.st) ? rb.dsq_s : rb.dsq_m;
    double rho0 = chi, rho1 = 1.0;
    if (rb.on && !(chi <= dsq)) { const double sq = sqrt(chi); rho0 = 2 * sq * delta - dsq; rho1 = delta / sq; }
    acc[27] += rho0;
    const double x = Xc[0], y = Xc[1], z = Xc[2];
    if (!STEREO || !d.st) {
        const double p00 = -(P.fx / z), p02 = P.fx * x / (z * z), p11 = -(P.fy / z), p12 = P.fy * y / (z * z);
        J[0] = p02 * y; J[1] = p00 * z + p02 * (-x); J[2] = p00 * (-y); J[3] = p00; J[4] = 0; J[5] = p02;
        J[6] = p11 * (-z) + p12 * y; J[7] = p12 * (-x); J[8] = p11 * x; J[9] = 0; J[10] = p11; J[11] = p12;
        if (STEREO) for (int k = 12; k < D * 6; k++) J[k] = 0;
    } else {
        const double invz = 1.0 / z, iz2 = invz * invz, fx = P.fx, fy = P.fy, bf = P.bf;
        J[0] = x * y * iz2 * fx; J[1] = -(1 + (x * x * iz2)) * fx; J[2] = y * invz * fx; J[3] = -invz * fx; J[4] = 0; J[5] = x * iz2 * fx;
        J[6] = (1 + y * y * iz2) * fy; J[7] = -x * y * iz2 * fy; J[8] = -x * invz * fy; J[9] = 0; J[10] = -invz * fy; J[11] = y * iz2 * fy;
        J[(D - 1) * 6 + 0] = J[0] - bf * y * iz2; J[(D - 1) * 6 + 1] = J[1] + bf * x * iz2; J[(D - 1) * 6 + 2] = J[2];
        J[(D - 1) * 6 + 3] = J[3]; J[(D - 1) * 6 + 4] = 0; J[(D - 1) * 6 + 5] = J[5] - bf * iz2;
    }
    // fixed trip counts (the third row of a mono edge is zero) so that everything stays in registers
    const double rw = rho1 * d.w;
    double wr[D];
#pragma unroll
    for (int q = 0; q < D; q++) wr[q] = d.w * r[q];
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int c = a; c < 6; c++) {
            double h = 0;
#pragma unroll
            for (int q = 0; q < D; q++) h += J[q * 6 + a] * rw * J[q * 6 + c];
            acc[a * 6 - (a * (a - 1)) / 2 + (c - a)] += h;
        }
        double sv = 0;
#pragma unroll
        for (int q = 0; q < D; q++) sv += J[q * 6 + a] * wr[q];
        acc[21 + a] -= rho1 * sv;
    }
}

// computeError + robust chi2 of one edge under the trial pose
template <bool STEREO>
__device__ __forceinline__ double edge_trial(const ProblemDev& P, const double* T, const Edge& d, const Robust& rb, double* r)
{
    double Xc[3];
    edge_eval<STEREO>(P, T, d, Xc, r);
    const double chi = edge_chi2<STEREO>(d, r);
    const double delta = (STEREO && d.st) ? rb.delta_s : rb.delta_m, dsq = (STEREO && d.st) ? rb.dsq_s : rb.dsq_m;
    if (rb.on && !(chi <= dsq)) return 2 * sqrt(chi) * delta - dsq;
    return chi;
}

// float chi2 against the 95 % thresholds (src/Optimizer.cc:1020-1034, 1049-1063)
template <bool STEREO>
__device__ __forceinline__ bool edge_is_outlier(const Edge& d, const double* r)
{
    const float chi2 = (float)edge_chi2<STEREO>(d, r);
    return chi2 > ((STEREO && d.st) ? 7.815f : 5.991f);
}

constexpr int kAccRow = 264;        // 256 partials + one pad per 32 (bank spread for the strided second stage)
// kRegEdges (template parameter R of the kernel): edges per thread held in registers -- 2 (n <= 512 never re-reads global memory) or,
// for frames with more edges (TrackWithMotionModel matches ~750 of 1000 features), 4; the per-thread summation order is the same.

// STEREO = the batch holds at least one stereo edge; the mono instantiation carries 2-row Jacobians only.
// The Levenberg state (pose, lambda, chi2, counters) and the 6x6 solve are REPLICATED in every thread: all lanes run the
// same scalar code on the same reduced values, so no broadcast barriers sit between the solve, the trial pass and the
// accept / reject decision -- the only synchronisation left is inside the two block reductions.
template <bool STEREO, int kRegEdges>
__global__ __launch_bounds__(256) void k_pose_opt(const ProblemDev* __restrict__ problems)
{
    __shared__ double s_acc[28][kAccRow];       // per-thread partials of H (21), b (6), chi2 (1), transposed
    __shared__ double s_out[28];
    __shared__ double s_red[4];
    const ProblemDev P = problems[blockIdx.x];
    const int tid = threadIdx.x;
    const int n = P.n;
    const int e_rest = tid + kRegEdges * 256;   // first edge of this thread that lives in global memory

    double T0[7] = {P.q[0], P.q[1], P.q[2], P.q[3], P.t[0], P.t[1], P.t[2]};
    quat_normalize(T0);                                 // SE3Quat(Quaterniond, Vector3d) (:829)
    double T[7];
    for (int k = 0; k < 7; k++) T[k] = T0[k];
    __shared__ int s_iters[4], s_trials[4];             // statistics only (thread 0)
    __shared__ double s_chi[4];
    if (tid < 4) { s_iters[tid] = 0; s_trials[tid] = 0; s_chi[tid] = 0; }
    // register-resident edges: data, level (active), outlier flag, last computed error
    Edge ce[kRegEdges];
    double cerr[kRegEdges][3];
    bool cvalid[kRegEdges], cact[kRegEdges], cout_[kRegEdges];
#pragma unroll
    for (int j = 0; j < kRegEdges; j++) {
        const int e = tid + j * 256;
        cvalid[j] = e < n;
        if (cvalid[j]) ce[j] = load_edge<STEREO>(P, e);
        else { ce[j].X[0] = 0; ce[j].X[1] = 0; ce[j].X[2] = 1; ce[j].o[0] = 0; ce[j].o[1] = 0; ce[j].o[2] = 0; ce[j].w = 0; ce[j].st = 0; }
        cact[j] = cvalid[j]; cout_[j] = false;
        cerr[j][0] = 0; cerr[j][1] = 0; cerr[j][2] = 0;
    }
    for (int e = e_rest; e < n; e += 256) { P.active[e] = 1; P.outlier[e] = 0; P.err[3 * (size_t)e] = 0; P.err[3 * (size_t)e + 1] = 0; P.err[3 * (size_t)e + 2] = 0; }
    Robust rb;
    rb.on = true;
    rb.delta_m = P.huber_mono; rb.delta_s = P.huber_stereo;
    rb.dsq_m = P.huber_mono * P.huber_mono; rb.dsq_s = P.huber_stereo * P.huber_stereo;
    int nBad = 0;
#ifdef POSE_TIMING
    long long tm[6] = {0, 0, 0, 0, 0, 0}, t_prev = clock64();
#define POSE_TICK(k) { const long long t_now = clock64(); tm[k] += t_now - t_prev; t_prev = t_now; }
#else
#define POSE_TICK(k)
#endif
    const int rounds = (n >= 3) ? 4 : 0;                // nInitialCorrespondences < 3 -> return 0 (:998-999)
#pragma unroll 1
    for (int round = 0; round < rounds; round++) {
        for (int k = 0; k < 7; k++) T[k] = T0[k];       // every round restarts from the frame pose (:1007-1008)
        // ---- optimizer.initializeOptimization(0); optimizer.optimize(10) ----
        double cnt = 0;
#pragma unroll
        for (int j = 0; j < kRegEdges; j++) cnt += cact[j] ? 1.0 : 0.0;
        for (int e = e_rest; e < n; e += 256) cnt += P.active[e];
        const int n_active = (int)block_sum(cnt, s_red);
        if (n_active > 0) {
            double lambda = 0, ni = 2;
            int nbad_lm = 0;
#pragma unroll 1
            for (int it = 0; it < 10; it++) {
                // computeActiveErrors + activeRobustChi2 + buildSystem on the current estimate
                double acc[28];
                for (int k = 0; k < 28; k++) acc[k] = 0;
#pragma unroll
                for (int j = 0; j < kRegEdges; j++)
                    if (cact[j]) edge_build<STEREO>(P, T, ce[j], rb, cerr[j], acc);
                for (int e = e_rest; e < n; e += 256) {
                    if (!P.active[e]) continue;
                    const Edge d = load_edge<STEREO>(P, e);
                    double r[3];
                    edge_build<STEREO>(P, T, d, rb, r, acc);
                    P.err[3 * (size_t)e] = r[0]; P.err[3 * (size_t)e + 1] = r[1]; P.err[3 * (size_t)e + 2] = r[2];
                }
                POSE_TICK(0)
                // two-stage ordered reduction through LDS: 28 values x 256 partials -> 8 partials of 32 -> 1
#pragma unroll
                for (int k = 0; k < 28; k++) s_acc[k][tid + (tid >> 5)] = acc[k];
                __syncthreads();
                if (tid < 224) {
                    const int k = tid >> 3, part = tid & 7;
                    const double* src = &s_acc[k][part * 33];
                    double v = 0;
#pragma unroll 8
                    for (int i = 0; i < 32; i++) v += src[i];
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
                    if (part == 0) s_out[k] = v;
                }
                __syncthreads();
                double Hu[21], b[6];
#pragma unroll
                for (int k = 0; k < 21; k++) Hu[k] = s_out[k];
#pragma unroll
                for (int k = 0; k < 6; k++) b[k] = s_out[21 + k];
                double cur = s_out[27];
                const double ini = cur;
                if (it == 0) {      // computeLambdaInit (levenberg.cpp:171-185)
                    double m = 0;
#pragma unroll
                    for (int j = 0; j < 6; j++) m = fmax(fabs(Hu[j * 6 - (j * (j - 1)) / 2]), m);
                    lambda = 1e-5 * m; ni = 2; nbad_lm = 0;
                }
                POSE_TICK(1)
                // ---- LM trial loop (levenberg.cpp:102-149) ----
                int qmax = 0;
                double rho = 0;
#pragma unroll 1
                do {
                    double x[6], Tt[7];
                    const bool ok2 = solve6(Hu, lambda, b, x);
                    if (ok2) pose_oplus<true>(T, x, Tt);
                    else {
                        for (int k = 0; k < 7; k++) Tt[k] = T[k];
                        for (int k = 0; k < 6; k++) x[k] = 0;
                    }
                    POSE_TICK(2)
                    double tchi = 0;
#pragma unroll
                    for (int j = 0; j < kRegEdges; j++)
                        if (cact[j]) tchi += edge_trial<STEREO>(P, Tt, ce[j], rb, cerr[j]);
                    for (int e = e_rest; e < n; e += 256) {
                        if (!P.active[e]) continue;
                        const Edge d = load_edge<STEREO>(P, e);
                        double r[3];
                        tchi += edge_trial<STEREO>(P, Tt, d, rb, r);
                        P.err[3 * (size_t)e] = r[0]; P.err[3 * (size_t)e + 1] = r[1]; P.err[3 * (size_t)e + 2] = r[2];
                    }
                    POSE_TICK(3)
                    double tempChi = block_sum(tchi, s_red);
                    if (!ok2) tempChi = 1.7976931348623157e308;
                    double scale = 0;
#pragma unroll
                    for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                    scale += 1e-3;
                    rho = (cur - tempChi) / scale;
                    if (rho > 0 && isfinite(tempChi)) {
                        const double c1 = 2 * rho - 1;
                        double alpha = 1. - c1 * c1 * c1;      // pow(2 rho - 1, 3) (levenberg.cpp:129), <= 2 ulp apart
                        alpha = fmin(alpha, 2. / 3.);
                        lambda *= fmax(1. / 3., alpha);
                        ni = 2;
                        cur = tempChi;
                        for (int k = 0; k < 7; k++) T[k] = Tt[k];     // discardTop()
                    } else {
                        lambda *= ni; ni *= 2;                          // pop(): the estimate stays
                    }
                    qmax++;
                    POSE_TICK(4)
                } while (rho < 0 && qmax < 10);
                // stop rules (:151-166)
                if (tid == 0) { s_iters[round]++; s_trials[round] += qmax; s_chi[round] = cur; }
                if (qmax == 10 || rho == 0) break;
                if ((ini - cur) * 1e3 < ini) nbad_lm++; else nbad_lm = 0;
                if (nbad_lm >= 3) break;
            }
        }
        // ---- inlier / outlier classification with float chi2 (:1016-1100) ----
        double bad = 0;
#pragma unroll
        for (int j = 0; j < kRegEdges; j++) {
            if (!cvalid[j]) continue;
            double Xc[3];
            if (cout_[j]) edge_eval<STEREO>(P, T, ce[j], Xc, cerr[j]);      // inactive edges did not follow the estimate: e->computeError()
            const bool o = edge_is_outlier<STEREO>(ce[j], cerr[j]);
            cout_[j] = o; cact[j] = !o; bad += o ? 1.0 : 0.0;
        }
        for (int e = e_rest; e < n; e += 256) {
            const Edge d = load_edge<STEREO>(P, e);
            double r[3];
            if (P.outlier[e]) {
                double Xc[3];
                edge_eval<STEREO>(P, T, d, Xc, r);
                P.err[3 * (size_t)e] = r[0]; P.err[3 * (size_t)e + 1] = r[1]; P.err[3 * (size_t)e + 2] = r[2];
            } else {
                r[0] = P.err[3 * (size_t)e]; r[1] = P.err[3 * (size_t)e + 1]; r[2] = P.err[3 * (size_t)e + 2];
            }
            const bool o = edge_is_outlier<STEREO>(d, r);
            P.outlier[e] = o ? 1 : 0; P.active[e] = o ? 0 : 1; bad += o ? 1.0 : 0.0;
        }
        nBad = (int)block_sum(bad, s_red);
        POSE_TICK(5)
        if (round == 2) rb.on = false;      // setRobustKernel(0) after the third round
        if (n < 10) break;                  // optimizer.edges().size() < 10
    }
#pragma unroll
    for (int j = 0; j < kRegEdges; j++)
        if (cvalid[j]) P.outlier[tid + j * 256] = cout_[j] ? 1 : 0;
    if (tid == 0) {
        PoseResult R;
        for (int k = 0; k < 4; k++) R.q[k] = T[k];
        for (int k = 0; k < 3; k++) R.t[k] = T[4 + k];
        R.n_bad = nBad;
        R.inliers = (n < 3) ? 0 : n - nBad;
        for (int k = 0; k < 4; k++) { R.iterations[k] = s_iters[k]; R.trials[k] = s_trials[k]; R.chi2[k] = s_chi[k]; }
#ifdef POSE_TIMING
        for (int k = 0; k < 4; k++) R.chi2[k] = (double)tm[k];
        R.t[0] = (double)tm[4]; R.t[1] = (double)tm[5];
#endif
        *P.result = R;
    }
}


// ---- device-resident entry (pose_optimize_batch_device): the frame's edges are gathered ON THE DEVICE from the arrays the extractor
// and the projection search left there, in feature order (the order Optimizer::PoseOptimization walks mvpMapPoints, :861-996) ----
struct GatherArgs {
    const OrbxKeyPoint* kps; const int32_t* n_kps; const float* u_right; const int32_t* assign; const float* mp_xyz; const double* pose;
    int32_t cap, mp_cap, n_levels;
    float inv_sigma2[16];
    double fx, fy, cx, cy, bf, huber_mono, huber_stereo;
    uint8_t* slots; size_t slot_bytes;              // per frame: [Xw | obs | w | err | idx | stereo | outl | act]
    size_t o_obs, o_w, o_err, o_idx, o_st, o_outl, o_act;
    ProblemDev* problems; PoseResult* results;
};

// one workgroup per frame: ordered compaction of the features that hold a map point (assign >= 0) into the solver's edge arrays;
// float -> double exactly as the reference casts them (obs << kpUn.pt.x ..., GetWorldPos().cast<double>(), mvInvLevelSigma2)
__global__ __launch_bounds__(256) void k_pose_gather(GatherArgs A)
{
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* slot = A.slots + (size_t)b * A.slot_bytes;
    double* Xw = (double*)slot; double* obs = (double*)(slot + A.o_obs); double* w = (double*)(slot + A.o_w);
    int32_t* idx = (int32_t*)(slot + A.o_idx); uint8_t* st = slot + A.o_st;
    const int n = min(max(A.n_kps[b], 0), A.cap);
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        int j = -1;
        if (i < n) { j = A.assign[(size_t)b * A.cap + i]; if (j >= A.mp_cap) j = -1; }
        const bool has = j >= 0;
        const unsigned long long bal = __ballot(has);
        const int in_wave = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int before = s_base;
        for (int q = 0; q < wave; q++) before += s_wave[q];
        if (has) {
            const int e = before + in_wave;
            const OrbxKeyPoint k = A.kps[(size_t)b * A.cap + i];
            const float* X = A.mp_xyz + 3 * ((size_t)b * A.mp_cap + j);
            const float ur = A.u_right ? A.u_right[(size_t)b * A.cap + i] : -1.0f;
            Xw[3 * (size_t)e] = (double)X[0]; Xw[3 * (size_t)e + 1] = (double)X[1]; Xw[3 * (size_t)e + 2] = (double)X[2];
            obs[3 * (size_t)e] = (double)k.x; obs[3 * (size_t)e + 1] = (double)k.y; obs[3 * (size_t)e + 2] = (double)ur;
            const int oc = min(max(k.octave, 0), A.n_levels - 1);
            w[e] = (double)A.inv_sigma2[oc];
            st[e] = (A.u_right && !(ur < 0.0f)) ? 1 : 0;          // mvuRight[i] < 0: monocular observation (:869)
            idx[e] = i;
        }
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (tid == 0) {
        ProblemDev P;
        for (int k = 0; k < 4; k++) P.q[k] = A.pose[7 * (size_t)b + k];
        for (int k = 0; k < 3; k++) P.t[k] = A.pose[7 * (size_t)b + 4 + k];
        P.n = s_base;
        P.Xw = Xw; P.obs = obs; P.w = w; P.stereo = st;
        P.fx = A.fx; P.fy = A.fy; P.cx = A.cx; P.cy = A.cy; P.bf = A.bf; P.huber_mono = A.huber_mono; P.huber_stereo = A.huber_stereo;
        P.err = (double*)(slot + A.o_err); P.outlier = slot + A.o_outl; P.active = slot + A.o_act;
        P.result = A.results + b;
        A.problems[b] = P;
    }
}

// results back into per-feature / per-frame arrays: mvbOutlier[i] (0 for a feature without a map point), the pose, the return value
__global__ __launch_bounds__(256) void k_pose_scatter(const ProblemDev* __restrict__ problems, const uint8_t* __restrict__ slots, size_t slot_bytes, size_t o_idx,
                                                      int cap, double* __restrict__ pose_out, int32_t* __restrict__ inliers, uint8_t* __restrict__ outlier,
                                                      PoseResult* __restrict__ results_out)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const ProblemDev& P = problems[b];
    const int32_t* idx = (const int32_t*)(slots + (size_t)b * slot_bytes + o_idx);
    if (outlier) {
        for (int i = tid; i < cap; i += 256) outlier[(size_t)b * cap + i] = 0;
        __syncthreads();
        for (int e = tid; e < P.n; e += 256) outlier[(size_t)b * cap + idx[e]] = P.outlier[e];
    }
    if (tid == 0) {
        const PoseResult R = *P.result;
        if (pose_out) { for (int k = 0; k < 4; k++) pose_out[7 * (size_t)b + k] = R.q[k]; for (int k = 0; k < 3; k++) pose_out[7 * (size_t)b + 4 + k] = R.t[k]; }
        if (inliers) inliers[b] = R.inliers;
        if (results_out) results_out[b] = R;
    }
}

}  // namespace poseopt

struct pose_solver {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint8_t* d_blob = nullptr;      // device image of h_blob + scratch
    uint8_t* h_blob = nullptr;      // pinned staging: [descriptors | inputs] up, [results | outlier flags] down
    size_t d_cap = 0, h_cap = 0;
    uint8_t* d_dev = nullptr;       // arena of the device-resident entry (edge slots, problem descriptors, results)
    size_t dev_cap = 0;
    float last_kernel_ms = 0;
};

namespace {
inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
inline void pose_launch(bool stereo, bool many_edges, int n, hipStream_t st, const poseopt::ProblemDev* p)
{
    if (stereo) {
        if (many_edges) hipLaunchKernelGGL((poseopt::k_pose_opt<true, 4>), dim3(n), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((poseopt::k_pose_opt<true, 2>), dim3(n), dim3(256), 0, st, p);
    } else {
        if (many_edges) hipLaunchKernelGGL((poseopt::k_pose_opt<false, 4>), dim3(n), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((poseopt::k_pose_opt<false, 2>), dim3(n), dim3(256), 0, st, p);
    }
}
}  // namespace

extern "C" {

int pose_create(int device, pose_solver** out)
{
    if (!out) return fail(ORBX_ERR_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ORBX_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(ORBX_ERR_ARG, "device %d out of range", device);
    POSE_HIP(hipSetDevice(device));
    pose_solver* s = new pose_solver();
    s->device = device;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&s->ev0) != hipSuccess ||
        hipEventCreate(&s->ev1) != hipSuccess) {
        pose_destroy(s);
        return fail(ORBX_ERR_HIP, "stream / event create failed");
    }
    *out = s;
    return ORBX_OK;
}

void pose_destroy(pose_solver* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->d_blob) (void)hipFree(s->d_blob);
    if (s->d_dev) (void)hipFree(s->d_dev);
    if (s->h_blob) (void)hipHostFree(s->h_blob);
    delete s;
}

float pose_last_kernel_ms(const pose_solver* s) { return s ? s->last_kernel_ms : 0.0f; }

int pose_optimize_batch(pose_solver* s, const PoseProblem* problems, int n_problems, PoseResult* results, uint8_t* const* outlier_out)
{
    if (!s || !problems || !results || n_problems < 1) return fail(ORBX_ERR_ARG, "bad arguments");
    POSE_HIP(hipSetDevice(s->device));
    // layout: [ProblemDev x N][per problem: Xw obs w stereo]  ||  [PoseResult x N][per problem: outlier]  ||  scratch
    struct Off { size_t Xw, obs, w, st, outl, err, act; };
    std::vector<Off> offs(n_problems);
    size_t pos = align16(sizeof(poseopt::ProblemDev) * (size_t)n_problems);
    for (int i = 0; i < n_problems; i++) {
        const PoseProblem& p = problems[i];
        if (p.n < 0 || (p.n > 0 && (!p.Xw || !p.obs || !p.inv_sigma2 || !p.stereo))) return fail(ORBX_ERR_ARG, "problem %d: NULL arrays", i);
        const size_t n = (size_t)p.n;
        Off& o = offs[i];
        o.Xw = pos; pos = align16(pos + 24 * n);
        o.obs = pos; pos = align16(pos + 24 * n);
        o.w = pos; pos = align16(pos + 8 * n);
        o.st = pos; pos = align16(pos + n);
    }
    const size_t up_bytes = pos;
    const size_t res_off = pos;
    pos = align16(pos + sizeof(PoseResult) * (size_t)n_problems);
    for (int i = 0; i < n_problems; i++) { offs[i].outl = pos; pos = align16(pos + (size_t)std::max(problems[i].n, 1)); }
    const size_t down_end = pos;
    for (int i = 0; i < n_problems; i++) {
        const size_t n = (size_t)std::max(problems[i].n, 1);
        offs[i].err = pos; pos = align16(pos + 24 * n);
        offs[i].act = pos; pos = align16(pos + n);
    }
    const size_t total = pos;
    if (down_end > s->h_cap) {
        if (s->h_blob) (void)hipHostFree(s->h_blob);
        s->h_blob = nullptr; s->h_cap = 0;
        const size_t cap = std::max(down_end * 2, (size_t)1 << 20);
        POSE_HIP(hipHostMalloc((void**)&s->h_blob, cap, hipHostMallocDefault));
        s->h_cap = cap;
    }
    if (total > s->d_cap) {
        if (s->d_blob) (void)hipFree(s->d_blob);
        s->d_blob = nullptr; s->d_cap = 0;
        const size_t cap = std::max(total * 2, (size_t)1 << 20);
        POSE_HIP(hipMalloc((void**)&s->d_blob, cap));
        s->d_cap = cap;
    }
    uint8_t* base = s->d_blob;
    poseopt::ProblemDev* descs = (poseopt::ProblemDev*)s->h_blob;
    bool any_stereo = false;
    for (int i = 0; i < n_problems; i++) {
        const PoseProblem& p = problems[i];
        const Off& o = offs[i];
        const size_t n = (size_t)p.n;
        for (size_t k = 0; k < n && !any_stereo; k++) any_stereo = p.stereo[k] != 0;
        if (n) {
            std::memcpy(s->h_blob + o.Xw, p.Xw, 24 * n); std::memcpy(s->h_blob + o.obs, p.obs, 24 * n);
            std::memcpy(s->h_blob + o.w, p.inv_sigma2, 8 * n); std::memcpy(s->h_blob + o.st, p.stereo, n);
        }
        poseopt::ProblemDev d;
        for (int k = 0; k < 4; k++) d.q[k] = p.q[k];
        for (int k = 0; k < 3; k++) d.t[k] = p.t[k];
        d.n = p.n;
        d.Xw = (const double*)(base + o.Xw); d.obs = (const double*)(base + o.obs); d.w = (const double*)(base + o.w); d.stereo = base + o.st;
        d.fx = p.fx; d.fy = p.fy; d.cx = p.cx; d.cy = p.cy; d.bf = p.bf; d.huber_mono = p.huber_mono; d.huber_stereo = p.huber_stereo;
        d.err = (double*)(base + o.err); d.outlier = base + o.outl; d.active = base + o.act;
        d.result = (PoseResult*)(base + res_off) + i;
        descs[i] = d;
    }
    POSE_HIP(hipMemcpyAsync(base, s->h_blob, up_bytes, hipMemcpyHostToDevice, s->stream));
    POSE_HIP(hipEventRecord(s->ev0, s->stream));
    int n_max = 0;
    for (int i = 0; i < n_problems; i++) n_max = std::max(n_max, problems[i].n);
    pose_launch(any_stereo, n_max > 512, n_problems, s->stream, (const poseopt::ProblemDev*)base);
    POSE_HIP(hipGetLastError());
    POSE_HIP(hipEventRecord(s->ev1, s->stream));
    POSE_HIP(hipMemcpyAsync(s->h_blob + res_off, base + res_off, down_end - res_off, hipMemcpyDeviceToHost, s->stream));
    POSE_HIP(hipStreamSynchronize(s->stream));
    (void)hipEventElapsedTime(&s->last_kernel_ms, s->ev0, s->ev1);
    std::memcpy(results, s->h_blob + res_off, sizeof(PoseResult) * (size_t)n_problems);
    if (outlier_out)
        for (int i = 0; i < n_problems; i++)
            if (outlier_out[i] && problems[i].n > 0) std::memcpy(outlier_out[i], s->h_blob + offs[i].outl, (size_t)problems[i].n);
    return ORBX_OK;
}

int pose_optimize_batch_device(pose_solver* s, const PoseDeviceFrames* f, int batch, double* d_pose_out, int32_t* d_inliers,
                               uint8_t* d_outlier, PoseResult* d_results, void* stream)
{
    if (!s || !f || batch < 1) return fail(ORBX_ERR_ARG, "bad arguments");
    if (!f->d_kps || !f->d_n || !f->d_assign || !f->d_mp_xyz || !f->d_pose || !f->inv_level_sigma2) return fail(ORBX_ERR_ARG, "NULL arrays");
    if (f->cap < 1 || f->mp_cap < 1 || f->n_levels < 1 || f->n_levels > 16) return fail(ORBX_ERR_ARG, "bad cap / mp_cap / n_levels");
    POSE_HIP(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t cap = (size_t)f->cap;
    poseopt::GatherArgs A;
    std::memset(&A, 0, sizeof(A));
    size_t pos = 0;
    pos = align16(pos + 24 * cap); A.o_obs = pos;
    pos = align16(pos + 24 * cap); A.o_w = pos;
    pos = align16(pos + 8 * cap); A.o_err = pos;
    pos = align16(pos + 24 * cap); A.o_idx = pos;
    pos = align16(pos + 4 * cap); A.o_st = pos;
    pos = align16(pos + cap); A.o_outl = pos;
    pos = align16(pos + cap); A.o_act = pos;
    pos = align16(pos + cap);
    A.slot_bytes = pos;
    const size_t prob_off = pos * (size_t)batch;
    const size_t res_off = align16(prob_off + sizeof(poseopt::ProblemDev) * (size_t)batch);
    const size_t total = align16(res_off + sizeof(PoseResult) * (size_t)batch);
    if (total > s->dev_cap) {           // (first call / larger batch: the only synchronising step)
        if (s->d_dev) { POSE_HIP(hipDeviceSynchronize()); (void)hipFree(s->d_dev); }
        s->d_dev = nullptr; s->dev_cap = 0;
        POSE_HIP(hipMalloc((void**)&s->d_dev, total));
        s->dev_cap = total;
    }
    A.kps = f->d_kps; A.n_kps = f->d_n; A.u_right = f->d_u_right; A.assign = f->d_assign; A.mp_xyz = f->d_mp_xyz; A.pose = f->d_pose;
    A.cap = f->cap; A.mp_cap = f->mp_cap; A.n_levels = f->n_levels;
    for (int l = 0; l < f->n_levels; l++) A.inv_sigma2[l] = f->inv_level_sigma2[l];
    A.fx = f->fx; A.fy = f->fy; A.cx = f->cx; A.cy = f->cy; A.bf = f->bf; A.huber_mono = f->huber_mono; A.huber_stereo = f->huber_stereo;
    A.slots = s->d_dev;
    A.problems = (poseopt::ProblemDev*)(s->d_dev + prob_off);
    A.results = (PoseResult*)(s->d_dev + res_off);
    hipLaunchKernelGGL(poseopt::k_pose_gather, dim3(batch), dim3(256), 0, st, A);
    pose_launch(f->d_u_right != nullptr, f->cap > 512, batch, st, (const poseopt::ProblemDev*)A.problems);
    hipLaunchKernelGGL(poseopt::k_pose_scatter, dim3(batch), dim3(256), 0, st, (const poseopt::ProblemDev*)A.problems, (const uint8_t*)s->d_dev, A.slot_bytes, A.o_idx,
                       f->cap, d_pose_out, d_inliers, d_outlier, d_results);
    POSE_HIP(hipGetLastError());
    return ORBX_OK;
}

int pose_optimize(pose_solver* s, const PoseProblem* problem, PoseResult* result, uint8_t* outlier)
{
    uint8_t* outs[1] = {outlier};
    return pose_optimize_batch(s, problem, 1, result, outs);
}

}  // extern "C"
