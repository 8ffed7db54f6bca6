// edge_packet.hip -- gfx950 kernels + C ABI for the fork's edge-SLAM wire format ("edge wire
// format"): the data format on either side of the hot path when features travel between an edge client and the server.
//   client side  SlamPktVI(id, timestamp, kps, descriptors, imus)   reference include/Socket/slampkt_vi.h:127-167
//                getHead()                                            :185-193
//   server side  SlamPktVI(buffer, packet_size)                       :85-125, consumed by src/Socket/client.cc:132-143 which
//                builds Frame(keypoints, descriptors, ...) (src/Frame.cc:384) with the extractor bypassed.
// Layout of one packet: 16-byte info block {int32 frame id (native LE), int64 time stamp (native LE, at byte 4), u16 number of
// key points BIG-endian, u16 number of IMU samples BIG-endian}; 36 bytes per key point {u16 x BE, u16 y BE, 32 descriptor
// bytes}, x / y = (unsigned short)pt.x / pt.y (truncation); 32 bytes per IMU sample {int64 ts, 3 x f32 gyro, 3 x f32 acce}.
//
// MI355X mapping.  Pure byte shuffling, HBM-bound (72 B moved per key point): one workgroup per frame, every record area is
// walked as a dword array so that both the packet side and the array side are touched with coalesced 4-byte accesses (a key
// point record is 9 dwords, an IMU record 8, the info block 4 -- every record start is 4-byte aligned because 16, 36 and 32
// are).  Packing consumes orbx_extract_batch_device's outputs in place, unpacking writes key points / descriptors in the
// layout the matcher, the vocabulary transform and AssignFeaturesToGrid kernels read -- no host round trip on either side.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../../include/orbslam3_hip.h"

namespace orbx {
int fail(int code, const char* fmt, ...);
}
using orbx::fail;

#define ORBE_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(ORBX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace orbe {

constexpr int kInfoLen = 16, kPtLen = 36, kImuLen = 32;     // slampkt_vi.h:19-21
constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t be16_pair(uint32_t x, uint32_t y)
{
    // bytes in memory order: x >> 8, x & 0xff, y >> 8, y & 0xff  (slampkt_vi.h:146-149)
    return ((x >> 8) & 0xffu) | ((x & 0xffu) << 8) | (((y >> 8) & 0xffu) << 16) | ((y & 0xffu) << 24);
}

// (unsigned short)float of the reference: truncation toward zero, then the low 16 bits (in-range values only are defined)
__device__ __forceinline__ uint32_t to_u16(float v) { return (uint32_t)(int32_t)v & 0xffffu; }

__global__ __launch_bounds__(kThreads) void k_pack_packets(const OrbxKeyPoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                          const int32_t* __restrict__ n_pts, int cap,
                                                          const int32_t* __restrict__ frame_id, const int64_t* __restrict__ timestamp,
                                                          const OrbeImuSample* __restrict__ imu, const int32_t* __restrict__ imu_off,
                                                          uint8_t* __restrict__ payload, int stride,
                                                          int32_t* __restrict__ len, uint8_t* __restrict__ head, int32_t* __restrict__ status)
{
    const int b = blockIdx.x;
    const int n = min(max(n_pts[b], 0), cap);
    const int i0 = imu ? imu_off[b] : 0;
    const int m = imu ? imu_off[b + 1] - i0 : 0;
    const int total = kInfoLen + n * kPtLen + m * kImuLen;           // total_len_ (:130)
    const bool fits = total <= stride && n <= 65535 && m >= 0 && m <= 65535;
    if (threadIdx.x == 0) {
        len[b] = total;
        // getHead(): two big-endian bytes of the packet size, not representable past 65536 (:185-192)
        status[b] = !fits ? ORBX_ERR_CAPACITY : (total > 65536 ? ORBX_ERR_ARG : ORBX_OK);
        if (head) {
            head[2 * b] = (uint8_t)(((uint32_t)total & 0xffffu) >> 8);
            head[2 * b + 1] = (uint8_t)((uint32_t)total & 0xffu);
        }
    }
    if (!fits) return;
    uint32_t* out = (uint32_t*)(payload + (size_t)b * stride);
    if (threadIdx.x == 0) {
        const uint64_t ts = (uint64_t)timestamp[b];
        out[0] = (uint32_t)frame_id[b];                               // int2byte(frame_id_, 0)
        out[1] = (uint32_t)ts;                                        // long2byte(time_stamp_, 4)
        out[2] = (uint32_t)(ts >> 32);
        out[3] = be16_pair((uint32_t)n, (uint32_t)m);                 // :140-143
    }
    const OrbxKeyPoint* kp = kps + (size_t)b * cap;
    const uint32_t* dsc = (const uint32_t*)(desc + (size_t)b * cap * 32);
    uint32_t* pts = out + kInfoLen / 4;
    for (int j = threadIdx.x; j < n * 9; j += kThreads) {
        const int p = j / 9, k = j - p * 9;
        pts[j] = k == 0 ? be16_pair(to_u16(kp[p].x), to_u16(kp[p].y)) : dsc[p * 8 + (k - 1)];
    }
    const uint32_t* src = (const uint32_t*)(imu + i0);
    uint32_t* dst = pts + n * 9;
    for (int j = threadIdx.x; j < m * 8; j += kThreads) dst[j] = src[j];     // same little-endian field order (:152-161)
}

__global__ __launch_bounds__(kThreads) void k_unpack_packets(const uint8_t* __restrict__ payload, int stride, const int32_t* __restrict__ len,
                                                            int cap, int imu_cap,
                                                            OrbxKeyPoint* __restrict__ kps, uint8_t* __restrict__ desc, int32_t* __restrict__ n_pts,
                                                            int32_t* __restrict__ frame_id, int64_t* __restrict__ timestamp,
                                                            OrbeImuSample* __restrict__ imu, int32_t* __restrict__ n_imu, int32_t* __restrict__ status)
{
    const int b = blockIdx.x;
    const uint32_t* in = (const uint32_t*)(payload + (size_t)b * stride);
    const int size = len[b];
    int n = 0, m = 0, st = ORBX_OK;
    if (size < kInfoLen || size > stride) st = ORBX_ERR_ARG;
    else {
        const uint32_t w = in[3];                                     // :92-96
        n = (int)(((w & 0xffu) << 8) | ((w >> 8) & 0xffu));
        m = (int)((((w >> 16) & 0xffu) << 8) | (w >> 24));
        // the reference reads whatever the counts say; a packet shorter than its own counts is rejected here instead
        if (kInfoLen + n * kPtLen + m * kImuLen > size) st = ORBX_ERR_ARG;
        else if (n > cap || m > imu_cap) st = ORBX_ERR_CAPACITY;
    }
    if (st != ORBX_OK) n = m = 0;
    if (threadIdx.x == 0) {
        status[b] = st;
        n_pts[b] = n;
        n_imu[b] = m;
        const bool hdr = size >= kInfoLen && size <= stride;
        frame_id[b] = hdr ? (int32_t)in[0] : 0;                       // byte2int(0)
        timestamp[b] = hdr ? (int64_t)((uint64_t)in[1] | ((uint64_t)in[2] << 32)) : 0;     // byte2long(4)
    }
    const uint32_t* pts = in + kInfoLen / 4;
    uint32_t* dsc = (uint32_t*)(desc + (size_t)b * cap * 32);
    for (int j = threadIdx.x; j < n * 8; j += kThreads) {
        const int p = j >> 3, k = j & 7;
        dsc[j] = pts[p * 9 + 1 + k];                                  // :102-104
    }
    OrbxKeyPoint* kp = kps + (size_t)b * cap;
    for (int p = threadIdx.x; p < n; p += kThreads) {
        const uint32_t w = pts[p * 9];                                // :99-101, KeyPoint(x, y, 1)
        OrbxKeyPoint o;
        o.x = (float)(((w & 0xffu) << 8) | ((w >> 8) & 0xffu));
        o.y = (float)((((w >> 16) & 0xffu) << 8) | (w >> 24));
        o.size = 1.0f; o.angle = -1.0f; o.response = 0.0f; o.octave = 0; o.class_id = -1;
        kp[p] = o;
    }
    if (imu) {
        const uint32_t* src = pts + n * 9;
        uint32_t* dst = (uint32_t*)(imu + (size_t)b * imu_cap);
        for (int j = threadIdx.x; j < m * 8; j += kThreads) dst[j] = src[j];     // :108-121
    }
}

// Frame::UndistortKeyPoints (reference src/Frame.cc:834-867) for the key points a server has just unpacked: cv::undistortPoints(mat, mat,
// K, mDistCoef, cv::Mat(), mK) = five fixed-point iterations of the radial / tangential model in double (OpenCV 4.x
// cvUndistortPointsInternal with TermCriteria(MAX_ITER, 5, 0.01), icdist < 0 guard), re-projected with the new camera matrix and
// stored as float; every other KeyPoint field is copied (mvKeysUn[i] = mvKeys[i] with a new pt).  k[0] == 0 means "no distortion":
// a plain copy, as at :836-840.  One thread per key point.
__global__ __launch_bounds__(kThreads) void k_undistort(const OrbxKeyPoint* __restrict__ kin, const int32_t* __restrict__ n_pts, int cap, int total,
                                                       OrbeCamera cam, OrbxKeyPoint* __restrict__ kout)
{
    const int g = blockIdx.x * kThreads + threadIdx.x;
    if (g >= total) return;
    const int b = g / cap, i = g - b * cap;
    if (i >= min(max(n_pts[b], 0), cap)) return;
    OrbxKeyPoint kp = kin[g];
    if (cam.k[0] != 0.0f) {
        const double fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy, ifx = 1. / fx, ify = 1. / fy;
        const double k0 = cam.k[0], k1 = cam.k[1], p1 = cam.k[2], p2 = cam.k[3], k4 = cam.k[4];
        const double u = kp.x, v = kp.y;
        double x = (u - cx) * ifx, y = (v - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = 1. / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);      // k[5..7] = 0: the numerator of OpenCV's rational model is 1
            if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
            const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        kp.x = (float)((double)cam.fx_new * x + (double)cam.cx_new);
        kp.y = (float)((double)cam.fy_new * y + (double)cam.cy_new);
    }
    kout[g] = kp;
}

}  // namespace orbe

struct orbe_codec {
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t* d_buf = nullptr;
    size_t d_bytes = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= d_bytes) return ORBX_OK;
        if (d_buf) (void)hipFree(d_buf);
        d_buf = nullptr; d_bytes = 0;
        ORBE_HIP(hipMalloc((void**)&d_buf, bytes));
        d_bytes = bytes;
        return ORBX_OK;
    }
};

namespace {
size_t al(size_t v) { return (v + 255) & ~(size_t)255; }
bool aligned4(const void* p) { return ((uintptr_t)p & 3u) == 0; }
}  // namespace

extern "C" {

int orbe_packet_bytes(int n_pts, int n_imu)
{
    if (n_pts < 0 || n_imu < 0) return ORBX_ERR_ARG;
    return orbe::kInfoLen + n_pts * orbe::kPtLen + n_imu * orbe::kImuLen;
}

int orbe_create(int device, orbe_codec** out)
{
    if (!out) return fail(ORBX_ERR_ARG, "null out");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return fail(ORBX_ERR_NO_DEVICE, "no HIP device: the packet codec runs on the GPU only");
    if (device < 0 || device >= count) return fail(ORBX_ERR_ARG, "device %d out of range", device);
    ORBE_HIP(hipSetDevice(device));
    orbe_codec* c = new orbe_codec;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(ORBX_ERR_HIP, "stream"); }
    *out = c;
    return ORBX_OK;
}

void orbe_destroy(orbe_codec* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int orbe_pack_batch_device(orbe_codec* c, const OrbxKeyPoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int batch, int cap,
                           const int32_t* d_frame_id, const int64_t* d_timestamp, const OrbeImuSample* d_imu, const int32_t* d_imu_off,
                           uint8_t* d_payload, int stride, int32_t* d_len, uint8_t* d_head, int32_t* d_status, void* stream)
{
    if (!c || batch < 1 || cap < 1 || !d_kps || !d_desc || !d_n || !d_frame_id || !d_timestamp || !d_payload || !d_len || !d_status)
        return fail(ORBX_ERR_ARG, "bad arguments");
    if ((d_imu == nullptr) != (d_imu_off == nullptr)) return fail(ORBX_ERR_ARG, "d_imu and d_imu_off go together");
    if (stride < orbe::kInfoLen || (stride & 3) || !aligned4(d_payload) || !aligned4(d_desc))
        return fail(ORBX_ERR_ARG, "packet stride must be a multiple of 4 (>= 16) and the buffers 4-byte aligned");
    ORBE_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(orbe::k_pack_packets, dim3(batch), dim3(orbe::kThreads), 0, (hipStream_t)stream, d_kps, d_desc, d_n, cap, d_frame_id,
                       d_timestamp, d_imu, d_imu_off, d_payload, stride, d_len, d_head, d_status);
    ORBE_HIP(hipGetLastError());
    return ORBX_OK;
}

int orbe_unpack_batch_device(orbe_codec* c, const uint8_t* d_payload, int stride, const int32_t* d_len, int batch, int cap, int imu_cap,
                             OrbxKeyPoint* d_kps, uint8_t* d_desc, int32_t* d_n, int32_t* d_frame_id, int64_t* d_timestamp,
                             OrbeImuSample* d_imu, int32_t* d_n_imu, int32_t* d_status, void* stream)
{
    if (!c || batch < 1 || cap < 1 || imu_cap < 0 || !d_payload || !d_len || !d_kps || !d_desc || !d_n || !d_frame_id || !d_timestamp || !d_n_imu || !d_status)
        return fail(ORBX_ERR_ARG, "bad arguments");
    if (imu_cap > 0 && !d_imu) return fail(ORBX_ERR_ARG, "imu_cap > 0 needs d_imu");
    if (stride < orbe::kInfoLen || (stride & 3) || !aligned4(d_payload) || !aligned4(d_desc))
        return fail(ORBX_ERR_ARG, "packet stride must be a multiple of 4 (>= 16) and the buffers 4-byte aligned");
    ORBE_HIP(hipSetDevice(c->device));
    hipLaunchKernelGGL(orbe::k_unpack_packets, dim3(batch), dim3(orbe::kThreads), 0, (hipStream_t)stream, d_payload, stride, d_len, cap,
                       imu_cap > 0 ? imu_cap : 1, d_kps, d_desc, d_n, d_frame_id, d_timestamp, imu_cap > 0 ? d_imu : nullptr, d_n_imu, d_status);
    ORBE_HIP(hipGetLastError());
    return ORBX_OK;
}

// Host-buffer forms (what the header-only shim calls): one staging blob up, one kernel, one blob down.
int orbe_pack_batch(orbe_codec* c, const OrbxKeyPoint* kps, const uint8_t* desc, const int32_t* n, int batch, int cap,
                    const int32_t* frame_id, const int64_t* timestamp, const OrbeImuSample* imu, const int32_t* imu_off,
                    uint8_t* payload, int stride, int32_t* len, uint8_t* head, int32_t* status)
{
    if (!c || batch < 1 || cap < 1 || !kps || !desc || !n || !frame_id || !timestamp || !payload || !len || !status)
        return fail(ORBX_ERR_ARG, "bad arguments");
    if ((imu == nullptr) != (imu_off == nullptr)) return fail(ORBX_ERR_ARG, "imu and imu_off go together");
    if (stride < orbe::kInfoLen || (stride & 3)) return fail(ORBX_ERR_ARG, "packet stride must be a multiple of 4 (>= 16)");
    const int n_imu = imu ? imu_off[batch] : 0;
    if (n_imu < 0) return fail(ORBX_ERR_ARG, "bad imu offsets");
    ORBE_HIP(hipSetDevice(c->device));
    const size_t B = (size_t)batch;
    const size_t o_kps = 0, o_desc = al(o_kps + B * cap * sizeof(OrbxKeyPoint)), o_n = al(o_desc + B * cap * 32), o_id = al(o_n + 4 * B),
                 o_ts = al(o_id + 4 * B), o_imu = al(o_ts + 8 * B), o_ioff = al(o_imu + (size_t)n_imu * sizeof(OrbeImuSample)),
                 o_in_end = al(o_ioff + 4 * (B + 1));
    const size_t o_pay = o_in_end, o_len = al(o_pay + B * stride), o_head = al(o_len + 4 * B), o_st = al(o_head + 2 * B), total = al(o_st + 4 * B);
    int r = c->ensure(total);
    if (r) return r;
    uint8_t* d = c->d_buf;
    hipStream_t s = c->stream;
    ORBE_HIP(hipMemcpyAsync(d + o_kps, kps, B * cap * sizeof(OrbxKeyPoint), hipMemcpyHostToDevice, s));
    ORBE_HIP(hipMemcpyAsync(d + o_desc, desc, B * cap * 32, hipMemcpyHostToDevice, s));
    ORBE_HIP(hipMemcpyAsync(d + o_n, n, 4 * B, hipMemcpyHostToDevice, s));
    ORBE_HIP(hipMemcpyAsync(d + o_id, frame_id, 4 * B, hipMemcpyHostToDevice, s));
    ORBE_HIP(hipMemcpyAsync(d + o_ts, timestamp, 8 * B, hipMemcpyHostToDevice, s));
    if (imu) {
        if (n_imu) ORBE_HIP(hipMemcpyAsync(d + o_imu, imu, (size_t)n_imu * sizeof(OrbeImuSample), hipMemcpyHostToDevice, s));
        ORBE_HIP(hipMemcpyAsync(d + o_ioff, imu_off, 4 * (B + 1), hipMemcpyHostToDevice, s));
    }
    ORBE_HIP(hipMemsetAsync(d + o_pay, 0, B * stride, s));
    r = orbe_pack_batch_device(c, (const OrbxKeyPoint*)(d + o_kps), d + o_desc, (const int32_t*)(d + o_n), batch, cap, (const int32_t*)(d + o_id),
                               (const int64_t*)(d + o_ts), imu ? (const OrbeImuSample*)(d + o_imu) : nullptr,
                               imu ? (const int32_t*)(d + o_ioff) : nullptr, d + o_pay, stride, (int32_t*)(d + o_len), d + o_head,
                               (int32_t*)(d + o_st), s);
    if (r) return r;
    ORBE_HIP(hipMemcpyAsync(payload, d + o_pay, B * stride, hipMemcpyDeviceToHost, s));
    ORBE_HIP(hipMemcpyAsync(len, d + o_len, 4 * B, hipMemcpyDeviceToHost, s));
    if (head) ORBE_HIP(hipMemcpyAsync(head, d + o_head, 2 * B, hipMemcpyDeviceToHost, s));
    ORBE_HIP(hipMemcpyAsync(status, d + o_st, 4 * B, hipMemcpyDeviceToHost, s));
    ORBE_HIP(hipStreamSynchronize(s));
    return ORBX_OK;
}

int orbe_unpack_batch(orbe_codec* c, const uint8_t* payload, int stride, const int32_t* len, int batch, int cap, int imu_cap,
                      OrbxKeyPoint* kps, uint8_t* desc, int32_t* n, int32_t* frame_id, int64_t* timestamp,
                      OrbeImuSample* imu, int32_t* n_imu, int32_t* status)
{
    if (!c || batch < 1 || cap < 1 || imu_cap < 0 || !payload || !len || !kps || !desc || !n || !frame_id || !timestamp || !n_imu || !status)
        return fail(ORBX_ERR_ARG, "bad arguments");
    if (imu_cap > 0 && !imu) return fail(ORBX_ERR_ARG, "imu_cap > 0 needs imu");
    if (stride < orbe::kInfoLen || (stride & 3)) return fail(ORBX_ERR_ARG, "packet stride must be a multiple of 4 (>= 16)");
    ORBE_HIP(hipSetDevice(c->device));
    const size_t B = (size_t)batch;
    const size_t o_pay = 0, o_len = al(o_pay + B * stride), o_kps = al(o_len + 4 * B), o_desc = al(o_kps + B * cap * sizeof(OrbxKeyPoint)),
                 o_n = al(o_desc + B * cap * 32), o_id = al(o_n + 4 * B), o_ts = al(o_id + 4 * B),
                 o_imu = al(o_ts + 8 * B), o_ni = al(o_imu + B * (size_t)imu_cap * sizeof(OrbeImuSample)), o_st = al(o_ni + 4 * B),
                 total = al(o_st + 4 * B);
    int r = c->ensure(total);
    if (r) return r;
    uint8_t* d = c->d_buf;
    hipStream_t s = c->stream;
    ORBE_HIP(hipMemcpyAsync(d + o_pay, payload, B * stride, hipMemcpyHostToDevice, s));
    ORBE_HIP(hipMemcpyAsync(d + o_len, len, 4 * B, hipMemcpyHostToDevice, s));
    ORBE_HIP(hipMemsetAsync(d + o_kps, 0, o_n - o_kps, s));
    if (imu_cap > 0) ORBE_HIP(hipMemsetAsync(d + o_imu, 0, o_ni - o_imu, s));
    r = orbe_unpack_batch_device(c, d + o_pay, stride, (const int32_t*)(d + o_len), batch, cap, imu_cap, (OrbxKeyPoint*)(d + o_kps), d + o_desc,
                                 (int32_t*)(d + o_n), (int32_t*)(d + o_id), (int64_t*)(d + o_ts),
                                 imu_cap > 0 ? (OrbeImuSample*)(d + o_imu) : nullptr, (int32_t*)(d + o_ni), (int32_t*)(d + o_st), s);
    if (r) return r;
    ORBE_HIP(hipMemcpyAsync(n, d + o_n, 4 * B, hipMemcpyDeviceToHost, s));
    ORBE_HIP(hipMemcpyAsync(n_imu, d + o_ni, 4 * B, hipMemcpyDeviceToHost, s));
    ORBE_HIP(hipMemcpyAsync(status, d + o_st, 4 * B, hipMemcpyDeviceToHost, s));
    ORBE_HIP(hipMemcpyAsync(frame_id, d + o_id, 4 * B, hipMemcpyDeviceToHost, s));
    ORBE_HIP(hipMemcpyAsync(timestamp, d + o_ts, 8 * B, hipMemcpyDeviceToHost, s));
    ORBE_HIP(hipMemcpyAsync(kps, d + o_kps, B * cap * sizeof(OrbxKeyPoint), hipMemcpyDeviceToHost, s));      // rows past n[b] are zero
    ORBE_HIP(hipMemcpyAsync(desc, d + o_desc, B * cap * 32, hipMemcpyDeviceToHost, s));
    if (imu_cap > 0) ORBE_HIP(hipMemcpyAsync(imu, d + o_imu, B * imu_cap * sizeof(OrbeImuSample), hipMemcpyDeviceToHost, s));
    ORBE_HIP(hipStreamSynchronize(s));
    return ORBX_OK;
}

int orbe_undistort_batch_device(orbe_codec* c, const OrbxKeyPoint* d_kps, const int32_t* d_n, int batch, int cap, const OrbeCamera* cam,
                                OrbxKeyPoint* d_kps_un, void* stream)
{
    if (!c || !d_kps || !d_n || !cam || !d_kps_un || batch < 1 || cap < 1) return fail(ORBX_ERR_ARG, "bad arguments");
    if (!(cam->fx != 0.0f) || !(cam->fy != 0.0f)) return fail(ORBX_ERR_ARG, "bad camera matrix");
    ORBE_HIP(hipSetDevice(c->device));
    const int total = batch * cap;
    hipLaunchKernelGGL(orbe::k_undistort, dim3((total + orbe::kThreads - 1) / orbe::kThreads), dim3(orbe::kThreads), 0, (hipStream_t)stream, d_kps, d_n, cap, total,
                       *cam, d_kps_un);
    ORBE_HIP(hipGetLastError());
    return ORBX_OK;
}

}  // extern "C"
