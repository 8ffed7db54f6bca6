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

__device__ __forceinline__ void edge_error(const ProblemDev& P, const double* T, int e, double* r)
{
    double Xc[3];
    pose_map(T, P.Xw + 3 * (size_t)e, Xc);
    const double* obs = P.obs + 3 * (size_t)e;
    if (!P.stereo[e]) {
        r[0] = obs[0] - (P.fx * Xc[0] / Xc[2] + P.cx);
        r[1] = obs[1] - (P.fy * Xc[1] / Xc[2] + P.cy);
        r[2] = 0;
    } else {
        const float invz = 1.0f / (float)Xc[2];              // float quirk (types_six_dof_expmap.cpp:339)
        const double u = Xc[0] * (double)invz * P.fx + P.cx;
        const double v = Xc[1] * (double)invz * P.fy + P.cy;
        r[0] = obs[0] - u; r[1] = obs[1] - v; r[2] = obs[2] - (u - P.bf * (double)invz);
    }
}

__device__ __forceinline__ double edge_chi2(const ProblemDev& P, int e, const double* r)
{
    const double w = P.w[e];
    double c = r[0] * (w * r[0]) + r[1] * (w * r[1]);
    if (P.stereo[e]) c += r[2] * (w * r[2]);
    return c;
}

__device__ __forceinline__ void huber(const ProblemDev& P, int e, bool robust, double chi, double& rho0, double& rho1)
{
    const double delta = P.stereo[e] ? P.huber_stereo : P.huber_mono;
    if (!robust || chi <= delta * delta) { rho0 = chi; rho1 = 1.0; }
    else { const double s = sqrt(chi); rho0 = 2 * s * delta - delta * delta; rho1 = delta / s; }
}

// 6x6 LDL^T solve (LinearSolverDense, solvers/linear_solver_dense.h:55-110); returns false for a non-positive pivot
__device__ inline bool solve6(const double* H, double lambda, const double* b, double* x)
{
    double A[36], D[6];
    for (int i = 0; i < 36; i++) A[i] = H[i];
    for (int i = 0; i < 6; i++) A[i * 7] += lambda;
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; k++) d -= A[j * 6 + k] * A[j * 6 + k] * D[k];
        if (!(d > 0.0) || !isfinite(d)) return false;
        D[j] = d;
        for (int i = j + 1; i < 6; i++) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; k++) s -= A[i * 6 + k] * A[j * 6 + k] * D[k];
            A[i * 6 + j] = s / d;
        }
    }
    for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= A[i * 6 + k] * x[k]; x[i] = s; }
    for (int i = 0; i < 6; i++) x[i] /= D[i];
    for (int i = 5; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < 6; k++) s -= A[k * 6 + i] * x[k]; x[i] = s; }
    return true;
}

__global__ __launch_bounds__(256) void k_pose_opt(const ProblemDev* __restrict__ problems)
{
    __shared__ double s_red[4];
    __shared__ double s_part[4][27];
    __shared__ double sT[7], sT0[7], sTt[7], sH[36], sb[6], sx[6];
    __shared__ double s_lambda, s_ni, s_cur, s_ini;
    __shared__ int s_flag, s_nbad_lm, s_qmax;
    const ProblemDev P = problems[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = P.n;

    if (tid == 0) {
        double T[7] = {P.q[0], P.q[1], P.q[2], P.q[3], P.t[0], P.t[1], P.t[2]};
        quat_normalize(T);                              // SE3Quat(Quaterniond, Vector3d) (:829)
        for (int k = 0; k < 7; k++) { sT0[k] = T[k]; sT[k] = T[k]; }
    }
    for (int e = tid; e < n; e += 256) { P.active[e] = 1; P.outlier[e] = 0; P.err[3 * (size_t)e] = 0; P.err[3 * (size_t)e + 1] = 0; P.err[3 * (size_t)e + 2] = 0; }
    __syncthreads();
    bool robust = true;
    int nBad = 0;
    const int rounds = (n >= 3) ? 4 : 0;                // nInitialCorrespondences < 3 -> return 0 (:998-999)
    for (int round = 0; round < rounds; round++) {
        if (tid < 7) sT[tid] = sT0[tid];                // every round restarts from the frame pose (:1007-1008)
        __syncthreads();
        // ---- optimizer.initializeOptimization(0); optimizer.optimize(10) ----
        double cnt = 0;
        for (int e = tid; e < n; e += 256) cnt += P.active[e];
        const int n_active = (int)block_sum(cnt, s_red);
        if (n_active > 0) {
            for (int it = 0; it < 10; it++) {
                // computeActiveErrors + activeRobustChi2 + buildSystem on the current estimate
                double acc[27];
                for (int k = 0; k < 27; k++) acc[k] = 0;
                double chi_sum = 0;
                for (int e = tid; e < n; e += 256) {
                    if (!P.active[e]) continue;
                    double r[3];
                    edge_error(P, sT, e, r);
                    P.err[3 * (size_t)e] = r[0]; P.err[3 * (size_t)e + 1] = r[1]; P.err[3 * (size_t)e + 2] = r[2];
                    double rho0, rho1;
                    huber(P, e, robust, edge_chi2(P, e, r), rho0, rho1);
                    chi_sum += rho0;
                    double Xc[3], J[18];
                    pose_map(sT, P.Xw + 3 * (size_t)e, Xc);
                    const double x = Xc[0], y = Xc[1], z = Xc[2];
                    const int st = P.stereo[e];
                    const int D = st ? 3 : 2;
                    if (!st) {
                        const double p00 = -(P.fx / z), p02 = P.fx * x / (z * z), p11 = -(P.fy / z), p12 = P.fy * y / (z * z);
                        J[0] = p02 * y; J[1] = p00 * z + p02 * (-x); J[2] = p00 * (-y); J[3] = p00; J[4] = 0; J[5] = p02;
                        J[6] = p11 * (-z) + p12 * y; J[7] = p12 * (-x); J[8] = p11 * x; J[9] = 0; J[10] = p11; J[11] = p12;
                        for (int k = 12; k < 18; k++) J[k] = 0;
                    } else {
                        const double invz = 1.0 / z, iz2 = invz * invz, fx = P.fx, fy = P.fy, bf = P.bf;
                        J[0] = x * y * iz2 * fx; J[1] = -(1 + (x * x * iz2)) * fx; J[2] = y * invz * fx; J[3] = -invz * fx; J[4] = 0; J[5] = x * iz2 * fx;
                        J[6] = (1 + y * y * iz2) * fy; J[7] = -x * y * iz2 * fy; J[8] = -x * invz * fy; J[9] = 0; J[10] = -invz * fy; J[11] = y * iz2 * fy;
                        J[12] = J[0] - bf * y * iz2; J[13] = J[1] + bf * x * iz2; J[14] = J[2]; J[15] = J[3]; J[16] = 0; J[17] = J[5] - bf * iz2;
                    }
                    const double w = P.w[e];
                    int idx = 0;
                    for (int a = 0; a < 6; a++)
                        for (int c = a; c < 6; c++, idx++) {
                            double h = 0;
                            for (int d = 0; d < D; d++) h += J[d * 6 + a] * (rho1 * w) * J[d * 6 + c];
                            acc[idx] += h;
                        }
                    for (int a = 0; a < 6; a++) {
                        double s = 0;
                        for (int d = 0; d < D; d++) s += J[d * 6 + a] * (w * r[d]);
                        acc[21 + a] -= rho1 * s;
                    }
                }
                const double currentChi0 = block_sum(chi_sum, s_red);
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
                        sH[a * 6 + c] = v; sH[c * 6 + a] = v;
                    } else sb[tid - 21] = v;
                }
                __syncthreads();
                if (tid == 0) {
                    s_cur = currentChi0; s_ini = currentChi0;
                    if (it == 0) {
                        double m = 0;
                        for (int j = 0; j < 6; j++) m = fmax(fabs(sH[j * 7]), m);
                        s_lambda = 1e-5 * m; s_ni = 2; s_nbad_lm = 0;       // computeLambdaInit (levenberg.cpp:171-185)
                    }
                    s_qmax = 0;
                }
                __syncthreads();
                // ---- LM trial loop (levenberg.cpp:102-149) ----
                bool again = true;
                double rho_last = 0;
                while (again) {
                    if (tid == 0) {
                        double x[6];
                        const bool ok2 = solve6(sH, s_lambda, sb, x);
                        for (int k = 0; k < 6; k++) sx[k] = ok2 ? x[k] : 0.0;
                        if (ok2) pose_oplus(sT, x, sTt); else for (int k = 0; k < 7; k++) sTt[k] = sT[k];
                        s_flag = ok2 ? 1 : 0;
                    }
                    __syncthreads();
                    double tchi = 0;
                    for (int e = tid; e < n; e += 256) {
                        if (!P.active[e]) continue;
                        double r[3];
                        edge_error(P, sTt, e, r);
                        P.err[3 * (size_t)e] = r[0]; P.err[3 * (size_t)e + 1] = r[1]; P.err[3 * (size_t)e + 2] = r[2];
                        double rho0, rho1;
                        huber(P, e, robust, edge_chi2(P, e, r), rho0, rho1);
                        tchi += rho0;
                    }
                    double tempChi = block_sum(tchi, s_red);
                    // every thread evaluates the same scalars (uniform control flow without another broadcast)
                    const bool ok2 = s_flag != 0;
                    if (!ok2) tempChi = 1.7976931348623157e308;
                    double scale = 0;
                    for (int j = 0; j < 6; j++) scale += sx[j] * (s_lambda * sx[j] + sb[j]);
                    scale += 1e-3;
                    const double rho = (s_cur - tempChi) / scale;
                    const bool good = rho > 0 && isfinite(tempChi);
                    const int qmax = s_qmax + 1;
                    __syncthreads();            // all reads of the shared LM state are done
                    if (tid == 0) {
                        if (good) {
                            double alpha = 1. - pow((2 * rho - 1), 3);
                            alpha = fmin(alpha, 2. / 3.);
                            s_lambda *= fmax(1. / 3., alpha);
                            s_ni = 2;
                            s_cur = tempChi;
                            for (int k = 0; k < 7; k++) sT[k] = sTt[k];     // discardTop()
                        } else {
                            s_lambda *= s_ni; s_ni *= 2;                    // pop(): keep sT
                        }
                        s_qmax = qmax;
                    }
                    __syncthreads();
                    rho_last = rho;
                    again = rho < 0 && qmax < 10;
                }
                // stop rules (:151-166), evaluated identically by every thread
                if (s_qmax == 10 || rho_last == 0) break;
                bool stop = false;
                if (tid == 0) {
                    if ((s_ini - s_cur) * 1e3 < s_ini) s_nbad_lm++; else s_nbad_lm = 0;
                }
                __syncthreads();
                stop = s_nbad_lm >= 3;
                __syncthreads();
                if (stop) break;
            }
        }
        // ---- inlier / outlier classification with float chi2 (:1016-1100) ----
        double bad = 0;
        for (int e = tid; e < n; e += 256) {
            double r[3];
            if (P.outlier[e]) {             // inactive edges did not follow the estimate: e->computeError()
                edge_error(P, sT, e, r);
                P.err[3 * (size_t)e] = r[0]; P.err[3 * (size_t)e + 1] = r[1]; P.err[3 * (size_t)e + 2] = r[2];
            } else {
                r[0] = P.err[3 * (size_t)e]; r[1] = P.err[3 * (size_t)e + 1]; r[2] = P.err[3 * (size_t)e + 2];
            }
            const float chi2 = (float)edge_chi2(P, e, r);
            const float thr = P.stereo[e] ? 7.815f : 5.991f;
            if (chi2 > thr) { P.outlier[e] = 1; P.active[e] = 0; bad += 1; }
            else { P.outlier[e] = 0; P.active[e] = 1; }
        }
        nBad = (int)block_sum(bad, s_red);
        if (round == 2) robust = false;     // setRobustKernel(0) after the third round
        if (n < 10) break;                  // optimizer.edges().size() < 10
    }
    if (tid == 0) {
        PoseResult R;
        for (int k = 0; k < 4; k++) R.q[k] = sT[k];
        for (int k = 0; k < 3; k++) R.t[k] = sT[4 + k];
        R.n_bad = nBad;
        R.inliers = (n < 3) ? 0 : n - nBad;
        *P.result = R;
    }
}

}  // namespace poseopt

struct pose_solver {
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t* d_blob = nullptr;
    size_t cap = 0;
    std::vector<uint8_t> host;
};

namespace {
struct PBlob {
    std::vector<uint8_t>& buf;
    explicit PBlob(std::vector<uint8_t>& b) : buf(b) { buf.clear(); }
    size_t put(const void* src, size_t bytes)
    {
        const size_t off = (buf.size() + 15) & ~(size_t)15;
        buf.resize(off + bytes);
        if (src && bytes) std::memcpy(buf.data() + off, src, bytes);
        return off;
    }
};
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
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) { delete s; return fail(ORBX_ERR_HIP, "stream create failed"); }
    *out = s;
    return ORBX_OK;
}

void pose_destroy(pose_solver* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    if (s->d_blob) (void)hipFree(s->d_blob);
    delete s;
}

int pose_optimize_batch(pose_solver* s, const PoseProblem* problems, int n_problems, PoseResult* results, uint8_t* const* outlier_out)
{
    if (!s || !problems || !results || n_problems < 1) return fail(ORBX_ERR_ARG, "bad arguments");
    POSE_HIP(hipSetDevice(s->device));
    PBlob blob(s->host);
    struct Off { size_t Xw, obs, w, st, err, outl, act, res; int n; };
    std::vector<Off> offs(n_problems);
    const size_t desc_off = blob.put(nullptr, sizeof(poseopt::ProblemDev) * n_problems);
    for (int i = 0; i < n_problems; i++) {
        const PoseProblem& p = problems[i];
        if (p.n < 0 || (p.n > 0 && (!p.Xw || !p.obs || !p.inv_sigma2 || !p.stereo))) return fail(ORBX_ERR_ARG, "problem %d: NULL arrays", i);
        Off& o = offs[i];
        o.n = p.n;
        o.Xw = blob.put(p.Xw, sizeof(double) * 3 * p.n); o.obs = blob.put(p.obs, sizeof(double) * 3 * p.n);
        o.w = blob.put(p.inv_sigma2, sizeof(double) * p.n); o.st = blob.put(p.stereo, p.n);
        o.err = blob.put(nullptr, sizeof(double) * 3 * std::max(p.n, 1));
        o.outl = blob.put(nullptr, std::max(p.n, 1)); o.act = blob.put(nullptr, std::max(p.n, 1));
        o.res = blob.put(nullptr, sizeof(PoseResult));
    }
    if (s->host.size() > s->cap) {
        if (s->d_blob) (void)hipFree(s->d_blob);
        s->d_blob = nullptr; s->cap = 0;
        const size_t cap = std::max(s->host.size() * 2, (size_t)1 << 20);
        POSE_HIP(hipMalloc((void**)&s->d_blob, cap));
        s->cap = cap;
    }
    uint8_t* base = s->d_blob;
    poseopt::ProblemDev* descs = (poseopt::ProblemDev*)(s->host.data() + desc_off);
    for (int i = 0; i < n_problems; i++) {
        const PoseProblem& p = problems[i];
        const Off& o = offs[i];
        poseopt::ProblemDev d;
        for (int k = 0; k < 4; k++) d.q[k] = p.q[k];
        for (int k = 0; k < 3; k++) d.t[k] = p.t[k];
        d.n = p.n;
        d.Xw = (const double*)(base + o.Xw); d.obs = (const double*)(base + o.obs); d.w = (const double*)(base + o.w); d.stereo = base + o.st;
        d.fx = p.fx; d.fy = p.fy; d.cx = p.cx; d.cy = p.cy; d.bf = p.bf; d.huber_mono = p.huber_mono; d.huber_stereo = p.huber_stereo;
        d.err = (double*)(base + o.err); d.outlier = base + o.outl; d.active = base + o.act; d.result = (PoseResult*)(base + o.res);
        descs[i] = d;
    }
    POSE_HIP(hipMemcpyAsync(base, s->host.data(), s->host.size(), hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(poseopt::k_pose_opt, dim3(n_problems), dim3(256), 0, s->stream, (const poseopt::ProblemDev*)(base + desc_off));
    POSE_HIP(hipGetLastError());
    for (int i = 0; i < n_problems; i++) {
        const Off& o = offs[i];
        POSE_HIP(hipMemcpyAsync(&results[i], base + o.res, sizeof(PoseResult), hipMemcpyDeviceToHost, s->stream));
        if (outlier_out && outlier_out[i] && o.n > 0) POSE_HIP(hipMemcpyAsync(outlier_out[i], base + o.outl, o.n, hipMemcpyDeviceToHost, s->stream));
    }
    POSE_HIP(hipStreamSynchronize(s->stream));
    return ORBX_OK;
}

int pose_optimize(pose_solver* s, const PoseProblem* problem, PoseResult* result, uint8_t* outlier)
{
    uint8_t* outs[1] = {outlier};
    return pose_optimize_batch(s, problem, 1, result, outs);
}

}  // extern "C"
