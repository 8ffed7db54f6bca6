// tracking_chain_demo.cpp -- Tracking::TrackWithMotionModel's device work (reference src/Tracking.cc:2975-3053) through the C ABI from
// plain C++ with the HIP runtime (no Python, no torch): a batch of synthetic streams, per stream a last frame and a current frame
// (the same texture 3 px to the right), everything device-resident and enqueued on ONE stream without a host visit in between:
//     orbx_extract_batch_device (current frames)  ->  orbm_search_by_projection_last_batch_device  ->  pose_optimize_batch_device
// Self-checks: most last-frame points are found again, and the optimised pose returns to the truth (identity) from a perturbed start.
// Build:  g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/tracking_chain_demo.cpp -Lorb_slam3-1_amd -lorbslam3_hip
//             -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/orb_slam3-1_amd -Wl,-rpath,/opt/rocm/lib -o tracking_chain_demo
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbslam3_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        const int rc_ = (call);                                                            \
        if (rc_ < 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, orbx_last_error()); return 1; } \
    } while (0)
#define HIP(call)                                                                          \
    do {                                                                                   \
        const hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } \
    } while (0)

static uint32_t rng_state = 2468u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static double urand() { return (double)(rnd() & 0xFFFFF) / (double)0x100000; }

template <typename T>
static T* dev_alloc(size_t n) { void* p = nullptr; if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr; (void)hipMemset(p, 0, n * sizeof(T)); return (T*)p; }

int main()
{
    if (orbx_device_count() < 1) { std::printf("no HIP device: the library has no CPU fallback\n"); return 77; }
    const int W = 640, H = 480, B = 8, SHIFT = 3;
    const double fx = 458.654f, fy = 457.296f, cx = 367.215f, cy = 248.375f;
    // per stream a blocky random texture (corners at the block boundaries); the current frame shows it SHIFT px further right
    std::vector<uint8_t> last((size_t)B * W * H), cur((size_t)B * W * H);
    for (int b = 0; b < B; b++) {
        std::vector<uint8_t> blocks((size_t)(W / 8 + 2) * (H / 8 + 2));
        for (auto& v : blocks) v = (uint8_t)(rnd() & 0xFF);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                last[((size_t)b * H + y) * W + x] = blocks[(size_t)(y / 8) * (W / 8 + 2) + x / 8];
                const int xs = x >= SHIFT ? x - SHIFT : x - SHIFT + W;
                cur[((size_t)b * H + y) * W + x] = blocks[(size_t)(y / 8) * (W / 8 + 2) + xs / 8];
            }
    }
    hipStream_t st = nullptr;
    HIP(hipStreamCreate(&st));
    orbx_extractor* ex = nullptr;
    orbm_matcher* m = nullptr;
    pose_solver* ps = nullptr;
    CHECK(orbx_create(1000, 1.2f, 8, 20, 7, 0, &ex));
    CHECK(orbm_create(0, &m));
    CHECK(pose_create(0, &ps));
    const int cap = orbx_max_keypoints(ex);
    const int L = orbx_levels(ex);
    std::vector<float> scale(L), inv_sigma2(L);
    CHECK(orbx_scale_tables(ex, scale.data(), nullptr, nullptr, inv_sigma2.data()));

    uint8_t* d_last = dev_alloc<uint8_t>(last.size());
    uint8_t* d_cur = dev_alloc<uint8_t>(cur.size());
    OrbxKeyPoint* l_kps = dev_alloc<OrbxKeyPoint>((size_t)B * cap);
    OrbxKeyPoint* c_kps = dev_alloc<OrbxKeyPoint>((size_t)B * cap);
    uint8_t* l_desc = dev_alloc<uint8_t>((size_t)B * cap * 32);
    uint8_t* c_desc = dev_alloc<uint8_t>((size_t)B * cap * 32);
    int32_t* l_n = dev_alloc<int32_t>(B);
    int32_t* c_n = dev_alloc<int32_t>(B);
    int32_t* d_mono = dev_alloc<int32_t>(B);
    int32_t* d_status = dev_alloc<int32_t>(B);
    if (!d_last || !d_cur || !l_kps || !c_kps || !l_desc || !c_desc || !l_n || !c_n || !d_mono || !d_status) { std::fprintf(stderr, "hipMalloc failed\n"); return 1; }
    HIP(hipMemcpy(d_last, last.data(), last.size(), hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_cur, cur.data(), cur.size(), hipMemcpyHostToDevice));

    // ---- the last frames: extracted once; their per-feature arrays stay resident (steady state of a tracker) ----
    CHECK(orbx_extract_batch_device(ex, d_last, B, W, H, W, (size_t)W * H, 0, 1000, l_kps, l_desc, cap, l_n, d_mono, d_status, st));
    HIP(hipStreamSynchronize(st));
    std::vector<OrbxKeyPoint> hk((size_t)B * cap);
    std::vector<int32_t> hn(B);
    HIP(hipMemcpy(hk.data(), l_kps, hk.size() * sizeof(OrbxKeyPoint), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(hn.data(), l_n, B * sizeof(int32_t), hipMemcpyDeviceToHost));
    // every last-frame feature holds a map point at a random depth; with the current camera at the identity it projects to the
    // feature's position moved by the shift
    std::vector<uint8_t> valid((size_t)B * cap, 0);
    std::vector<float> pu((size_t)B * cap, 0.f), pv((size_t)B * cap, 0.f), pang((size_t)B * cap, 0.f), xyz((size_t)B * cap * 3, 0.f);
    std::vector<int32_t> poct((size_t)B * cap, 0);
    for (int b = 0; b < B; b++)
        for (int i = 0; i < hn[b]; i++) {
            const size_t k = (size_t)b * cap + i;
            const OrbxKeyPoint& kp = hk[k];
            const float z = (float)(2.0 + 12.0 * urand());
            valid[k] = 1; pu[k] = kp.x + (float)SHIFT; pv[k] = kp.y; pang[k] = kp.angle; poct[k] = kp.octave;
            xyz[3 * k] = (float)((pu[k] - cx) / fx * z); xyz[3 * k + 1] = (float)((pv[k] - cy) / fy * z); xyz[3 * k + 2] = z;
        }
    uint8_t* d_valid = dev_alloc<uint8_t>(valid.size());
    float* d_pu = dev_alloc<float>(pu.size());
    float* d_pv = dev_alloc<float>(pv.size());
    float* d_pang = dev_alloc<float>(pang.size());
    int32_t* d_poct = dev_alloc<int32_t>(poct.size());
    float* d_xyz = dev_alloc<float>(xyz.size());
    // the motion model's prediction: a few centimetres / a fraction of a degree off the truth (identity); qx qy qz qw tx ty tz
    std::vector<double> pose0((size_t)B * 7, 0.0);
    for (int b = 0; b < B; b++) {
        for (int k = 0; k < 3; k++) { pose0[7 * b + k] = 0.01 * (urand() - 0.5); pose0[7 * b + 4 + k] = 0.06 * (urand() - 0.5); }
        pose0[7 * b + 3] = 1.0;
    }
    double* d_pose0 = dev_alloc<double>(pose0.size());
    double* d_pose = dev_alloc<double>((size_t)B * 7);
    int32_t* d_assign = dev_alloc<int32_t>((size_t)B * cap);
    uint8_t* d_occ = dev_alloc<uint8_t>((size_t)B * cap);
    int32_t* d_nm = dev_alloc<int32_t>(B);
    int32_t* d_inl = dev_alloc<int32_t>(B);
    uint8_t* d_outl = dev_alloc<uint8_t>((size_t)B * cap);
    if (!d_valid || !d_pu || !d_pv || !d_pang || !d_poct || !d_xyz || !d_pose0 || !d_pose || !d_assign || !d_occ || !d_nm || !d_inl || !d_outl) { std::fprintf(stderr, "hipMalloc failed\n"); return 1; }
    HIP(hipMemcpy(d_valid, valid.data(), valid.size(), hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_pu, pu.data(), pu.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_pv, pv.data(), pv.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_pang, pang.data(), pang.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_poct, poct.data(), poct.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_xyz, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_pose0, pose0.data(), pose0.size() * 8, hipMemcpyHostToDevice));
    HIP(hipMemset(d_assign, 0xFF, (size_t)B * cap * sizeof(int32_t)));        // -1: CurrentFrame.mvpMapPoints cleared (:2986)

    // ---- the chain: three calls enqueued back to back on one stream ----
    CHECK(orbx_extract_batch_device(ex, d_cur, B, W, H, W, (size_t)W * H, 0, 1000, c_kps, c_desc, cap, c_n, d_mono, d_status, st));
    OrbmDeviceFrames cf{};
    cf.d_kps = c_kps; cf.d_desc = c_desc; cf.d_n = c_n; cf.cap = cap;
    cf.min_x = 0.f; cf.min_y = 0.f; cf.max_x = (float)W; cf.max_y = (float)H; cf.grid_cols = 64; cf.grid_rows = 48;
    cf.scale_factors = scale.data(); cf.n_levels = L;
    OrbmDeviceLastPoints lp{};
    lp.d_valid = d_valid; lp.d_u = d_pu; lp.d_v = d_pv; lp.d_octave = d_poct; lp.d_angle = d_pang; lp.d_desc = l_desc; lp.d_n = l_n; lp.cap = cap;
    lp.d_has_obs = nullptr;
    CHECK(orbm_search_by_projection_last_batch_device(m, &cf, &lp, B, 15.0f, 1, d_assign, d_occ, d_nm, st));
    PoseDeviceFrames pf{};
    pf.d_kps = c_kps; pf.d_n = c_n; pf.d_u_right = nullptr; pf.cap = cap;
    pf.d_assign = d_assign; pf.d_mp_xyz = d_xyz; pf.mp_cap = cap; pf.d_pose = d_pose0;
    pf.inv_level_sigma2 = inv_sigma2.data(); pf.n_levels = L;
    pf.fx = fx; pf.fy = fy; pf.cx = cx; pf.cy = cy; pf.bf = 0.0; pf.huber_mono = std::sqrt(5.991f); pf.huber_stereo = std::sqrt(7.815f);
    CHECK(pose_optimize_batch_device(ps, &pf, B, d_pose, d_inl, d_outl, nullptr, st));
    HIP(hipStreamSynchronize(st));

    std::vector<int32_t> nm(B), inl(B), cn(B), status(B);
    std::vector<double> pose((size_t)B * 7);
    HIP(hipMemcpy(nm.data(), d_nm, B * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(inl.data(), d_inl, B * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(cn.data(), c_n, B * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(status.data(), d_status, B * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(pose.data(), d_pose, pose.size() * 8, hipMemcpyDeviceToHost));
    bool ok = true;
    for (int b = 0; b < B; b++) {
        double terr = 0, qerr = 0;
        for (int k = 0; k < 3; k++) { terr = std::fmax(terr, std::fabs(pose[7 * b + 4 + k])); qerr = std::fmax(qerr, std::fabs(pose[7 * b + k])); }
        std::printf("stream %d: %d / %d key points (last / current), %d projection matches, %d inliers, |t| %.4f m, |q_xyz| %.5f after optimisation\n",
                    b, hn[b], cn[b], nm[b], inl[b], terr, qerr);
        ok = ok && status[b] == 0 && nm[b] > hn[b] / 3 && inl[b] > nm[b] / 2 && terr < 0.02 && qerr < 0.005;
    }
    pose_destroy(ps); orbm_destroy(m); orbx_destroy(ex);
    (void)hipStreamDestroy(st);
    if (!ok) { std::printf("tracking chain demo FAILED\n"); return 1; }
    std::printf("tracking chain demo OK\n");
    return 0;
}
