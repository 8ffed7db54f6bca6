/*
 * edge_packet_oracle.cpp -- CPU ORACLE (test infrastructure, not the product) for the fork's edge-SLAM wire format,
 * class SlamPktVI of the reference, include/Socket/slampkt_vi.h.
 *   edge_oracle_pack    follows SlamPktVI(id, timestamp, kps, descriptors, imus)  :127-167 and getHead() :185-193
 *   edge_oracle_unpack  follows SlamPktVI(buffer, packet_size)                    :85-125
 * byte for byte (int2byte / long2byte / float2byte are memcpy of the native representation, :30-83; the point and IMU
 * counts and the point coordinates are big-endian 16-bit).  PARITY UNPINNED: the reference holds no packet fixture; the
 * hand-computed known-answer packet in tests/test_edge_packet_oracle.py pins the layout to the header's comments (:19-21).
 *
 * The reference reads as many records as the counts in the info block say, whatever packet_size is; this restatement
 * returns -1 for such a packet instead of reading past the buffer.
 */
#include <cstdint>
#include <cstring>

#include "oracle.h"

namespace {
const int kInfoLen = 16, kPtLen = 36, kImuLen = 32, kDescLen = 32;     /* :19-22 */
}

extern "C" {

int edge_oracle_pack(int32_t frame_id, int64_t timestamp, const OracleKeyPoint* kps, const uint8_t* desc, int n_pts,
                     const OracleImuSample* imu, int n_imu, uint8_t* payload, int capacity, uint8_t head[2])
{
    const int total = kInfoLen + n_pts * kPtLen + n_imu * kImuLen;     /* :130 */
    if (head) {                                                           /* getHead(): undefined (nullptr) past 65536 */
        head[0] = (unsigned char)((unsigned short)total >> 8);
        head[1] = (unsigned char)((unsigned short)total & 0xff);
    }
    if (total > capacity) return -total;
    std::memcpy(payload + 0, &frame_id, 4);                               /* int2byte(frame_id_, 0) */
    std::memcpy(payload + 4, &timestamp, 8);                              /* long2byte(time_stamp_, 4) */
    payload[12] = (unsigned char)((unsigned short)n_pts >> 8);
    payload[13] = (unsigned char)((unsigned short)n_pts & 0xff);
    payload[14] = (unsigned char)((unsigned short)n_imu >> 8);
    payload[15] = (unsigned char)((unsigned short)n_imu & 0xff);
    for (int i = 0; i < n_pts; i++) {                                     /* :145-153 */
        unsigned char* p = payload + i * kPtLen + kInfoLen;
        p[0] = (unsigned char)((unsigned short)kps[i].x >> 8);
        p[1] = (unsigned char)((unsigned short)kps[i].x & 0xff);
        p[2] = (unsigned char)((unsigned short)kps[i].y >> 8);
        p[3] = (unsigned char)((unsigned short)kps[i].y & 0xff);
        for (int j = 0; j < kDescLen; j++) p[4 + j] = desc[i * kDescLen + j];
    }
    const int imu_start = kInfoLen + n_pts * kPtLen;                      /* :155-166 */
    for (int i = 0; i < n_imu; i++) {
        unsigned char* p = payload + i * kImuLen + imu_start;
        std::memcpy(p, &imu[i].ts, 8);
        for (int j = 0; j < 3; j++) std::memcpy(p + 8 + j * 4, &imu[i].gyro[j], 4);
        for (int j = 0; j < 3; j++) std::memcpy(p + 20 + j * 4, &imu[i].acce[j], 4);
    }
    return total;
}

int edge_oracle_unpack(const uint8_t* payload, int packet_size, int32_t* frame_id, int64_t* timestamp,
                       OracleKeyPoint* kps, uint8_t* desc, int cap_pts, int* n_pts_out,
                       OracleImuSample* imu, int cap_imu, int* n_imu_out)
{
    *n_pts_out = *n_imu_out = 0;
    if (packet_size < kInfoLen) return -1;
    std::memcpy(frame_id, payload + 0, 4);                                /* byte2int(0) */
    std::memcpy(timestamp, payload + 4, 8);                               /* byte2long(4) */
    const int n_pts = (int)(unsigned short)(((unsigned short)payload[12]) * 256 + (unsigned short)payload[13]);
    const int n_imu = (int)(unsigned short)(((unsigned short)payload[14]) * 256 + (unsigned short)payload[15]);
    if (kInfoLen + n_pts * kPtLen + n_imu * kImuLen > packet_size) return -1;
    if (n_pts > cap_pts || n_imu > cap_imu) return -2;
    for (int i = 0; i < n_pts; i++) {                                     /* :98-106 */
        const unsigned char* p = payload + i * kPtLen + kInfoLen;
        const unsigned short x = (unsigned short)(((unsigned short)p[0]) * 256 + (unsigned short)p[1]);
        const unsigned short y = (unsigned short)(((unsigned short)p[2]) * 256 + (unsigned short)p[3]);
        OracleKeyPoint k;                                                 /* cv::KeyPoint(x, y, 1) and its defaults */
        k.x = (float)x; k.y = (float)y; k.size = 1.0f; k.angle = -1.0f; k.response = 0.0f; k.octave = 0; k.class_id = -1;
        kps[i] = k;
        for (int j = 0; j < kDescLen; j++) desc[i * kDescLen + j] = p[4 + j];
    }
    const int imu_start = kInfoLen + n_pts * kPtLen;                      /* :108-121 */
    for (int i = 0; i < n_imu; i++) {
        const unsigned char* p = payload + i * kImuLen + imu_start;
        std::memcpy(&imu[i].ts, p, 8);
        for (int j = 0; j < 3; j++) std::memcpy(&imu[i].gyro[j], p + 8 + j * 4, 4);
        for (int j = 0; j < 3; j++) std::memcpy(&imu[i].acce[j], p + 20 + j * 4, 4);
    }
    *n_pts_out = n_pts;
    *n_imu_out = n_imu;
    return 0;
}

}  // extern "C"

/* Frame::UndistortKeyPoints (reference src/Frame.cc:834-867): cv::undistortPoints(mat, mat, K, mDistCoef, cv::Mat(), mK) restated from
 * OpenCV 4.x cvUndistortPointsInternal [UPSTREAM, not in /root/reference; PARITY UNPINNED]: double arithmetic, five iterations
 * (TermCriteria(MAX_ITER, 5, 0.01)), the icdist < 0 guard, no tilt, R = I, P = mK; results stored as float. */
extern "C" void edge_oracle_undistort(const OracleKeyPoint* kin, int n, const float K[4], const float dist[5], const float Knew[4], OracleKeyPoint* kout)
{
    for (int i = 0; i < n; i++) {
        OracleKeyPoint kp = kin[i];
        if (dist[0] != 0.0f) {
            const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], ifx = 1. / fx, ify = 1. / fy;
            const double k[5] = {dist[0], dist[1], dist[2], dist[3], dist[4]};
            const double u = kp.x, v = kp.y;
            double x = (u - cx) * ifx, y = (v - cy) * ify;
            const double x0 = x, y0 = y;
            for (int j = 0; j < 5; j++) {
                const double r2 = x * x + y * y;
                const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
                if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
                const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
                const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
                x = (x0 - deltaX) * icdist;
                y = (y0 - deltaY) * icdist;
            }
            kp.x = (float)((double)Knew[0] * x + (double)Knew[2]);
            kp.y = (float)((double)Knew[1] * y + (double)Knew[3]);
        }
        kout[i] = kp;
    }
}
