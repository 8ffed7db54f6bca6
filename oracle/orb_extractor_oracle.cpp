/*
 * ORACLE (test infrastructure, not product) -- CPU restatement of
 * ORBextractor::operator() and the OpenCV primitives it calls.
 *
 * Follows (read as text; nothing copied):
 *   /root/reference/src/ORBextractor.cc:71-146   IC_Angle, computeOrbDescriptor
 *   /root/reference/src/ORBextractor.cc:409-469  constructor tables
 *   /root/reference/src/ORBextractor.cc:480-779  ExtractorNode::DivideNode, DistributeOctTree
 *   /root/reference/src/ORBextractor.cc:781-896  ComputeKeyPointsOctTree
 *   /root/reference/src/ORBextractor.cc:1077-1195 computeDescriptors, operator(), ComputePyramid
 * Third-party arithmetic NOT in /root/reference (OpenCV >=3.4, README pins 4.4.0;
 * SURVEY.md Appendix A): cv::FAST (FAST-9/16 + cornerScore + 3x3 NMS),
 * cv::resize INTER_LINEAR 8UC1 (11-bit fixed point), cv::GaussianBlur 7x7 8UC1
 * (4.x Q8.8 fixed-point path, error-diffused taps), cv::fastAtan2, cvRound.
 * PARITY UNPINNED at that boundary: the reference has no tests/golden vectors.
 *
 * Single-threaded.  Build with -ffp-contract=off so float expressions are
 * evaluated without FMA contraction (the HIP kernels do the same).
 */
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <list>
#include <vector>

namespace {

typedef OracleKeyPoint KP;

const int kPatchSize = 31;      // ORBextractor.cc:71
const int kHalfPatch = 15;      // :72
const int kEdge = 19;           // :73 EDGE_THRESHOLD

const signed char kPattern[1024] = {
#include "orb_pattern_31.inc"
};

// ---- cvRound / cvFloor / cvCeil (OpenCV core/fast_math.hpp): round-half-even ----
inline int cv_round(double v) { return (int)std::nearbyint(v); }     // default rounding mode = to nearest even
inline int cv_roundf(float v) { return (int)std::nearbyintf(v); }
inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

// ---- cv::fastAtan2 (OpenCV core/mathfuncs_core: atan_f32), degrees in [0,360) ----
float fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.141592653589793238462643383279502884);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---- cos/sin of a float angle, computed in double and rounded to float ----
// The reference evaluates (float)cos(angle), (float)sin(angle) with a float
// argument (ORBextractor.cc:111-112).  libm results are not reproducible on a
// GPU, so oracle and kernels both use this fixed double-precision algorithm
// (Cody-Waite reduction by pi/2 + Taylor polynomials, no FMA): |err| ~1e-16
// before the final rounding to float, i.e. the correctly rounded float except
// in astronomically rare ties.  Valid for |x| < ~1e5; the extractor feeds [0, 2pi].
void sincos_f(float angle, float* c_out, float* s_out)
{
    const double x = (double)angle;
    const double kd = std::nearbyint(x * 0.63661977236758134308);
    const int k = (int)kd;
    const double r = (x - kd * 1.57079632679489655800e+00) - kd * 6.12323399573676603587e-17;
    const double z = r * r;
    // sin(r) = r * (1 - z/3! + z^2/5! - ... + z^8/17!)
    double sp = 2.81145725434552075980e-15;            //  1/17!
    sp = sp * z + -7.64716373181981647590e-13;         // -1/15!
    sp = sp * z + 1.60590438368216145994e-10;          //  1/13!
    sp = sp * z + -2.50521083854417187751e-08;         // -1/11!
    sp = sp * z + 2.75573192239858906526e-06;          //  1/9!
    sp = sp * z + -1.98412698412698412698e-04;         // -1/7!
    sp = sp * z + 8.33333333333333333333e-03;          //  1/5!
    sp = sp * z + -1.66666666666666666667e-01;         // -1/3!
    const double s = r + r * (sp * z);
    // cos(r) = 1 - z/2! + z^2/4! - ... + z^9/18!
    double cp = -1.56192069685862264433e-16;           // -1/18!
    cp = cp * z + 4.77947733238738529744e-14;          //  1/16!
    cp = cp * z + -1.14707455977297247139e-11;         // -1/14!
    cp = cp * z + 2.08767569878680989792e-09;          //  1/12!
    cp = cp * z + -2.75573192239858906526e-07;         // -1/10!
    cp = cp * z + 2.48015873015873015873e-05;          //  1/8!
    cp = cp * z + -1.38888888888888888889e-03;         // -1/6!
    cp = cp * z + 4.16666666666666666667e-02;          //  1/4!
    cp = cp * z + -5.00000000000000000000e-01;         // -1/2!
    const double c = 1.0 + cp * z;
    double cc, ss;
    switch (k & 3) {
        case 0: cc = c; ss = s; break;
        case 1: cc = -s; ss = c; break;
        case 2: cc = -c; ss = -s; break;
        default: cc = s; ss = -c; break;
    }
    *c_out = (float)cc;
    *s_out = (float)ss;
}

// ---- cv::FAST(img, kps, threshold, nonmax, TYPE_9_16) restated (Appendix A.1) ----
// Ring offsets (dx,dy) k=0..15, wrapped to 25 entries.
const int kRingDx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int kRingDy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

int corner_score16(const uint8_t* p, const int* off, int threshold)
{
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[off[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++) a = std::min(a, d[k + j]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        for (int j = 6; j <= 8; j++) b = std::max(b, d[k + j]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

void fast9_16(const uint8_t* img, int cols, int rows, int stride, int threshold, bool nonmax, std::vector<KP>& out)
{
    out.clear();
    if (cols < 7 || rows < 7) return;
    int off[25];
    for (int k = 0; k < 25; k++) off[k] = kRingDy[k % 16] * stride + kRingDx[k % 16];
    threshold = std::min(std::max(threshold, 0), 255);
    uint8_t threshold_tab[512];
    for (int i = -255; i <= 255; i++) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    // three rolling rows of scores + corner positions, exactly like the upstream scan
    std::vector<uint8_t> bufmem((size_t)cols * 3, 0);
    std::vector<int> cpmem((size_t)(cols + 1) * 3, 0);
    uint8_t* buf[3] = {&bufmem[0], &bufmem[cols], &bufmem[2 * cols]};
    int* cpbuf[3] = {&cpmem[0], &cpmem[cols + 1], &cpmem[2 * (cols + 1)]};

    for (int i = 3; i < rows - 2; i++) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3] + 1;
        std::memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; j++, ptr++) {
                const int v = ptr[0];
                // upstream quick reject: any 9-arc contains one pixel of each opposite pair (k, k+8)
                const uint8_t* tab = &threshold_tab[0] - v + 255;
                int d = tab[ptr[off[0]]] | tab[ptr[off[8]]];
                if (d == 0) continue;
                d &= tab[ptr[off[2]]] | tab[ptr[off[10]]];
                d &= tab[ptr[off[4]]] | tab[ptr[off[12]]];
                d &= tab[ptr[off[6]]] | tab[ptr[off[14]]];
                if (d == 0) continue;
                d &= tab[ptr[off[1]]] | tab[ptr[off[9]]];
                d &= tab[ptr[off[3]]] | tab[ptr[off[11]]];
                d &= tab[ptr[off[5]]] | tab[ptr[off[13]]];
                d &= tab[ptr[off[7]]] | tab[ptr[off[15]]];
                // 9 contiguous ring pixels all darker than v-t or all brighter than v+t (25-long wrapped ring)
                bool corner = false;
                if (d & 1) {
                    const int vt = v - threshold;
                    int count = 0;
                    for (int k = 0; k < 25; k++) {
                        if (ptr[off[k]] < vt) { if (++count > 8) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (!corner && (d & 2)) {
                    const int vt = v + threshold;
                    int count = 0;
                    for (int k = 0; k < 25; k++) {
                        if (ptr[off[k]] > vt) { if (++count > 8) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (corner) {
                    cornerpos[ncorners++] = j;
                    if (nonmax) curr[j] = (uint8_t)corner_score16(ptr, off, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3] + 1;
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            const int j = cornerpos[k];
            const int score = prev[j];
            if (!nonmax ||
                (score > prev[j + 1] && score > prev[j - 1] &&
                 score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                 score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
                KP kp;
                kp.x = (float)j; kp.y = (float)(i - 1); kp.size = 7.f; kp.angle = -1.f;
                kp.response = (float)score; kp.octave = 0; kp.class_id = -1;
                out.push_back(kp);
            }
        }
    }
}

// ---- cv::resize(..., INTER_LINEAR) for 8UC1, generic fixed-point path (Appendix A.2) ----
inline short sat_short_from_float(float v)
{
    int iv = cv_roundf(v);
    return (short)std::min(std::max(iv, -32768), 32767);
}

void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride)
{
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha((size_t)dw * 2), ibeta((size_t)dh * 2);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_short_from_float((1.f - fx) * 2048);
        ialpha[dx * 2 + 1] = sat_short_from_float(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = sat_short_from_float((1.f - fy) * 2048);
        ibeta[dy * 2 + 1] = sat_short_from_float(fy * 2048);
    }
    std::vector<int> row0(dw), row1(dw);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);
        int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            const int sx = xofs[dx];
            const int sx1 = std::min(sx + 1, sw - 1);     // weight is 0 when clamped
            const int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1];
            row0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            row1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) {
            int v = (((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)std::min(std::max(v, 0), 255);
        }
    }
}

// ---- cv::GaussianBlur(src, dst, Size(k,k), sigma, sigma, BORDER_REFLECT_101), 8UC1, OpenCV 4.x ----
// Fixed-point path: taps in Q8.8 (ufixedpoint16) with error diffusion and sum forced to 256
// (getGaussianKernelFixedPoint_ED), horizontal pass u8*Q8.8 -> Q8.8, vertical Q8.8*Q8.8 -> Q16.16,
// rounded (v + 0x8000) >> 16 (Appendix A.3).
void gauss_taps_q8(int n, double sigma, int* taps)
{
    std::vector<double> k(n);
    const double scale2x = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        k[i] = std::exp(scale2x * x * x);
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] *= sum;
    const int n2 = n / 2;
    double err = 0;
    long long acc = 0;
    for (int i = 0; i < n2; i++) {
        double adj = k[i] * 256.0 + err;
        long long v0 = cv_round(adj);
        err = adj - (double)v0;
        taps[i] = (int)v0;
        taps[n - 1 - i] = (int)v0;
        acc += v0;
    }
    taps[n2] = (int)(256 - 2 * acc);
}

inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

void gaussian7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    int t[7];
    gauss_taps_q8(7, 2.0, t);
    std::vector<uint16_t> hbuf((size_t)w * h);
    std::vector<uint8_t> prow((size_t)w + 6);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        for (int x = -3; x < w + 3; x++) prow[x + 3] = S[reflect101(x, w)];
        uint16_t* H = &hbuf[(size_t)y * w];
        for (int x = 0; x < w; x++) {
            const uint8_t* p = &prow[x];
            unsigned acc = (unsigned)t[0] * (p[0] + p[6]) + (unsigned)t[1] * (p[1] + p[5]) +
                           (unsigned)t[2] * (p[2] + p[4]) + (unsigned)t[3] * p[3];
            H[x] = (uint16_t)std::min(acc, 65535u);   // Q8.8, cannot saturate: taps sum to 256
        }
    }
    for (int y = 0; y < h; y++) {
        const uint16_t* R[7];
        for (int k = 0; k < 7; k++) R[k] = &hbuf[(size_t)reflect101(y + k - 3, h) * w];
        uint8_t* D = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            uint32_t acc = (uint32_t)t[0] * ((uint32_t)R[0][x] + R[6][x]) + (uint32_t)t[1] * ((uint32_t)R[1][x] + R[5][x]) +
                           (uint32_t)t[2] * ((uint32_t)R[2][x] + R[4][x]) + (uint32_t)t[3] * R[3][x];
            D[x] = (uint8_t)std::min((acc + 0x8000u) >> 16, 255u);
        }
    }
}

// ---- the extractor ----
struct Node {
    std::vector<KP> keys;
    int ulx, uly, urx, ury, blx, bly, brx, bry;
    std::list<Node>::iterator lit;
    bool no_more;
    Node() : ulx(0), uly(0), urx(0), ury(0), blx(0), bly(0), brx(0), bry(0), no_more(false) {}
};

// ExtractorNode::DivideNode (ORBextractor.cc:480-536)
void divide_node(const Node& p, Node c[4])
{
    const int halfX = (int)std::ceil((float)(p.urx - p.ulx) / 2);
    const int halfY = (int)std::ceil((float)(p.bry - p.uly) / 2);
    c[0].ulx = p.ulx;          c[0].uly = p.uly;
    c[0].urx = p.ulx + halfX;  c[0].ury = p.uly;
    c[0].blx = p.ulx;          c[0].bly = p.uly + halfY;
    c[0].brx = p.ulx + halfX;  c[0].bry = p.uly + halfY;
    c[1].ulx = c[0].urx; c[1].uly = c[0].ury;
    c[1].urx = p.urx;    c[1].ury = p.ury;
    c[1].blx = c[0].brx; c[1].bly = c[0].bry;
    c[1].brx = p.urx;    c[1].bry = p.uly + halfY;
    c[2].ulx = c[0].blx; c[2].uly = c[0].bly;
    c[2].urx = c[0].brx; c[2].ury = c[0].bry;
    c[2].blx = p.blx;    c[2].bly = p.bly;
    c[2].brx = c[0].brx; c[2].bry = p.bly;
    c[3].ulx = c[2].urx; c[3].uly = c[2].ury;
    c[3].urx = c[1].brx; c[3].ury = c[1].bry;
    c[3].blx = c[2].brx; c[3].bly = c[2].bry;
    c[3].brx = p.brx;    c[3].bry = p.bry;
    for (int i = 0; i < 4; i++) c[i].keys.reserve(p.keys.size());
    for (size_t i = 0; i < p.keys.size(); i++) {
        const KP& kp = p.keys[i];
        if (kp.x < c[0].urx) {
            if (kp.y < c[0].bry) c[0].keys.push_back(kp);
            else c[2].keys.push_back(kp);
        } else if (kp.y < c[0].bry) c[1].keys.push_back(kp);
        else c[3].keys.push_back(kp);
    }
    for (int i = 0; i < 4; i++) if (c[i].keys.size() == 1) c[i].no_more = true;
}

typedef std::pair<int, Node*> SizeNode;

// compareNodes (:538-553)
bool node_less(SizeNode& a, SizeNode& b)
{
    if (a.first < b.first) return true;
    if (a.first > b.first) return false;
    return a.second->ulx < b.second->ulx;
}

struct Extractor {
    int nfeatures, nlevels, ini_th, min_th;
    double scale_factor_d;      // member is double in the reference (include/ORBextractor.h:92)
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> nfeat, umax;
    // last call
    std::vector<int> lw, lh;
    std::vector<std::vector<uint8_t> > pyr, blurred;
    std::vector<std::vector<KP> > cand, lkeys;

    Extractor(int nf, float sf, int nl, int ini, int mn)
        : nfeatures(nf), nlevels(nl), ini_th(ini), min_th(mn), scale_factor_d((double)sf)
    {
        // :413-429 (float products through the double member)
        scale.resize(nl); sigma2.resize(nl); inv_scale.resize(nl); inv_sigma2.resize(nl);
        scale[0] = 1.0f; sigma2[0] = 1.0f;
        for (int i = 1; i < nl; i++) {
            scale[i] = (float)(scale[i - 1] * scale_factor_d);
            sigma2[i] = scale[i] * scale[i];
        }
        for (int i = 0; i < nl; i++) {
            inv_scale[i] = 1.0f / scale[i];
            inv_sigma2[i] = 1.0f / sigma2[i];
        }
        // :433-445
        nfeat.resize(nl);
        float factor = (float)(1.0f / scale_factor_d);
        float per_scale = nf * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
        int sum = 0;
        for (int l = 0; l < nl - 1; l++) {
            nfeat[l] = cv_roundf(per_scale);
            sum += nfeat[l];
            per_scale *= factor;
        }
        nfeat[nl - 1] = std::max(nf - sum, 0);
        // :453-468 umax
        umax.assign(kHalfPatch + 1, 0);
        int v, v0;
        const int vmax = cv_floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
        const int vmin = cv_ceil(kHalfPatch * std::sqrt(2.f) / 2);
        const double hp2 = kHalfPatch * kHalfPatch;
        for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt(hp2 - v * v));
        for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
        lw.assign(nl, 0); lh.assign(nl, 0);
        pyr.resize(nl); blurred.resize(nl); cand.resize(nl); lkeys.resize(nl);
    }

    // ComputePyramid (:1170-1195).  Borders (copyMakeBorder) are not materialised: the mono
    // path never reads them (keypoints lie in [19, W-20], patch radius <= 18).
    void compute_pyramid(const uint8_t* img, int w, int h, int stride)
    {
        for (int l = 0; l < nlevels; l++) {
            const float s = inv_scale[l];
            lw[l] = cv_roundf((float)w * s);
            lh[l] = cv_roundf((float)h * s);
            pyr[l].assign((size_t)std::max(lw[l], 0) * std::max(lh[l], 0), 0);
            if (lw[l] <= 0 || lh[l] <= 0) continue;
            if (l == 0) {
                for (int y = 0; y < h; y++) std::memcpy(&pyr[0][(size_t)y * w], img + (size_t)y * stride, w);
            } else {
                resize_linear_u8(&pyr[l - 1][0], lw[l - 1], lh[l - 1], lw[l - 1], &pyr[l][0], lw[l], lh[l], lw[l]);
            }
        }
    }

    // DistributeOctTree (:555-779)
    std::vector<KP> distribute(const std::vector<KP>& in, int minX, int maxX, int minY, int maxY, int N)
    {
        const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
        const float hX = (float)(maxX - minX) / nIni;
        std::list<Node> nodes;
        std::vector<Node*> ini(std::max(nIni, 0));
        for (int i = 0; i < nIni; i++) {
            Node ni;
            ni.ulx = (int)(hX * (float)i);       ni.uly = 0;
            ni.urx = (int)(hX * (float)(i + 1)); ni.ury = 0;
            ni.blx = ni.ulx; ni.bly = maxY - minY;
            ni.brx = ni.urx; ni.bry = maxY - minY;
            ni.keys.reserve(in.size());
            nodes.push_back(ni);
            ini[i] = &nodes.back();
        }
        if (nIni > 0)
            for (size_t i = 0; i < in.size(); i++) ini[(int)(in[i].x / hX)]->keys.push_back(in[i]);

        std::list<Node>::iterator lit = nodes.begin();
        while (lit != nodes.end()) {
            if (lit->keys.size() == 1) { lit->no_more = true; ++lit; }
            else if (lit->keys.empty()) lit = nodes.erase(lit);
            else ++lit;
        }

        bool finish = false;
        std::vector<SizeNode> expandable;
        expandable.reserve(nodes.size() * 4);

        // push the non-empty children to the FRONT of the list in order 1..4 (:639-676)
        auto push_children = [&](Node c[4], int* n_to_expand) {
            for (int i = 0; i < 4; i++) {
                if (c[i].keys.empty()) continue;
                nodes.push_front(c[i]);
                if (c[i].keys.size() > 1) {
                    if (n_to_expand) (*n_to_expand)++;
                    expandable.push_back(std::make_pair((int)c[i].keys.size(), &nodes.front()));
                    nodes.front().lit = nodes.begin();
                }
            }
        };

        while (!finish) {
            int prev_size = (int)nodes.size();
            lit = nodes.begin();
            int n_to_expand = 0;
            expandable.clear();
            while (lit != nodes.end()) {
                if (lit->no_more) { ++lit; continue; }
                Node c[4];
                divide_node(*lit, c);
                push_children(c, &n_to_expand);
                lit = nodes.erase(lit);
            }
            if ((int)nodes.size() >= N || (int)nodes.size() == prev_size) {
                finish = true;
            } else if ((int)nodes.size() + n_to_expand * 3 > N) {
                while (!finish) {
                    prev_size = (int)nodes.size();
                    std::vector<SizeNode> prev = expandable;
                    expandable.clear();
                    std::sort(prev.begin(), prev.end(), node_less);     // unstable libstdc++ introsort (:700)
                    for (int j = (int)prev.size() - 1; j >= 0; j--) {
                        Node c[4];
                        divide_node(*prev[j].second, c);
                        push_children(c, 0);
                        nodes.erase(prev[j].second->lit);
                        if ((int)nodes.size() >= N) break;
                    }
                    if ((int)nodes.size() >= N || (int)nodes.size() == prev_size) finish = true;
                }
            }
        }

        // best response per node, first wins ties (:758-776)
        std::vector<KP> result;
        result.reserve(nfeatures);
        for (lit = nodes.begin(); lit != nodes.end(); ++lit) {
            const std::vector<KP>& k = lit->keys;
            const KP* best = &k[0];
            float max_resp = best->response;
            for (size_t i = 1; i < k.size(); i++)
                if (k[i].response > max_resp) { best = &k[i]; max_resp = k[i].response; }
            result.push_back(*best);
        }
        return result;
    }

    // IC_Angle (:76-103)
    float ic_angle(const uint8_t* img, int stride, float px, float py) const
    {
        int m_01 = 0, m_10 = 0;
        const uint8_t* center = img + (size_t)cv_roundf(py) * stride + cv_roundf(px);
        for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m_10 += u * center[u];
        for (int v = 1; v <= kHalfPatch; ++v) {
            int v_sum = 0;
            const int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        return fast_atan2((float)m_01, (float)m_10);
    }

    // ComputeKeyPointsOctTree (:781-896)
    void compute_keypoints()
    {
        const float W = 35;
        for (int level = 0; level < nlevels; ++level) {
            cand[level].clear();
            lkeys[level].clear();
            const int cols = lw[level], rows = lh[level];
            const int minBorderX = kEdge - 3, minBorderY = minBorderX;
            const int maxBorderX = cols - kEdge + 3, maxBorderY = rows - kEdge + 3;
            const float width = (float)(maxBorderX - minBorderX);
            const float height = (float)(maxBorderY - minBorderY);
            const int nCols = (int)(width / W), nRows = (int)(height / W);
            if (nCols <= 0 || nRows <= 0) continue;     // image too small for a single cell (reference would divide by zero)
            const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
            const uint8_t* img = &pyr[level][0];
            std::vector<KP>& to_distribute = cand[level];
            to_distribute.reserve(nfeatures * 10);
            std::vector<KP> cell;
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBorderY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = (float)maxBorderY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBorderX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - 6) continue;
                    if (maxX > maxBorderX) maxX = (float)maxBorderX;
                    const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
                    const uint8_t* sub = img + (size_t)y0 * cols + x0;
                    fast9_16(sub, x1 - x0, y1 - y0, cols, ini_th, true, cell);
                    if (cell.empty()) fast9_16(sub, x1 - x0, y1 - y0, cols, min_th, true, cell);
                    for (size_t k = 0; k < cell.size(); k++) {
                        cell[k].x += j * wCell;
                        cell[k].y += i * hCell;
                        to_distribute.push_back(cell[k]);
                    }
                }
            }
            std::vector<KP>& keys = lkeys[level];
            keys = distribute(to_distribute, minBorderX, maxBorderX, minBorderY, maxBorderY, nfeat[level]);
            const int scaledPatchSize = (int)(kPatchSize * scale[level]);
            for (size_t i = 0; i < keys.size(); i++) {
                keys[i].x += minBorderX;
                keys[i].y += minBorderY;
                keys[i].octave = level;
                keys[i].size = (float)scaledPatchSize;
            }
        }
        for (int level = 0; level < nlevels; ++level) {
            std::vector<KP>& keys = lkeys[level];
            for (size_t i = 0; i < keys.size(); i++)
                keys[i].angle = ic_angle(&pyr[level][0], lw[level], keys[i].x, keys[i].y);
        }
    }

    // computeOrbDescriptor (:107-146)
    void descriptor(const KP& kpt, const uint8_t* img, int step, uint8_t* desc) const
    {
        const float factorPI = (float)(3.141592653589793238462643383279502884 / 180.f);
        const float angle = (float)kpt.angle * factorPI;
        float a, b;
        sincos_f(angle, &a, &b);
        const uint8_t* center = img + (size_t)cv_roundf(kpt.y) * step + cv_roundf(kpt.x);
        const signed char* pat = kPattern;
        for (int i = 0; i < 32; ++i, pat += 32) {
            int val = 0;
            for (int k = 0; k < 8; k++) {
                const float x0 = pat[4 * k], y0 = pat[4 * k + 1], x1 = pat[4 * k + 2], y1 = pat[4 * k + 3];
                const int t0 = center[cv_roundf(x0 * b + y0 * a) * step + cv_roundf(x0 * a - y0 * b)];
                const int t1 = center[cv_roundf(x1 * b + y1 * a) * step + cv_roundf(x1 * a - y1 * b)];
                val |= (t0 < t1) << k;
            }
            desc[i] = (uint8_t)val;
        }
    }

    // operator() (:1086-1168)
    int extract(const uint8_t* img, int w, int h, int stride, int lap0, int lap1, KP* kps, uint8_t* desc, int cap, int* n_out)
    {
        *n_out = 0;
        if (!img || w <= 0 || h <= 0) return -1;
        compute_pyramid(img, w, h, stride);
        compute_keypoints();
        int nkeypoints = 0;
        for (int l = 0; l < nlevels; l++) nkeypoints += (int)lkeys[l].size();
        *n_out = nkeypoints;
        if (nkeypoints > cap) return -2;
        int monoIndex = 0, stereoIndex = nkeypoints - 1;
        uint8_t d[32];
        for (int level = 0; level < nlevels; ++level) {
            blurred[level].clear();
            std::vector<KP>& keys = lkeys[level];
            if (keys.empty()) continue;
            blurred[level].resize((size_t)lw[level] * lh[level]);
            gaussian7_u8(&pyr[level][0], lw[level], lh[level], lw[level], &blurred[level][0], lw[level]);
            const float sc = scale[level];
            for (size_t i = 0; i < keys.size(); i++) {
                descriptor(keys[i], &blurred[level][0], lw[level], d);
                KP kp = keys[i];
                if (level != 0) { kp.x *= sc; kp.y *= sc; }
                int dstIdx;
                if (kp.x >= lap0 && kp.x <= lap1) dstIdx = stereoIndex--;
                else dstIdx = monoIndex++;
                kps[dstIdx] = kp;
                std::memcpy(desc + (size_t)dstIdx * 32, d, 32);
            }
        }
        return monoIndex;
    }
};

}  // namespace

extern "C" {

void* orb_oracle_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
{
    if (nlevels < 1 || nlevels > 32) return 0;
    return new Extractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
}
void orb_oracle_destroy(void* h) { delete (Extractor*)h; }

int orb_oracle_extract(void* h, const uint8_t* img, int w, int hgt, int stride, int lap0, int lap1,
                       OracleKeyPoint* kps, uint8_t* desc, int cap, int* n_out)
{
    return ((Extractor*)h)->extract(img, w, hgt, stride, lap0, lap1, kps, desc, cap, n_out);
}

void orb_oracle_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* nfeat, int* umax16)
{
    Extractor* e = (Extractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        if (scale) scale[i] = e->scale[i];
        if (inv_scale) inv_scale[i] = e->inv_scale[i];
        if (sigma2) sigma2[i] = e->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = e->inv_sigma2[i];
        if (nfeat) nfeat[i] = e->nfeat[i];
    }
    if (umax16) for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}

int orb_oracle_level_size(void* h, int level, int* w, int* hgt)
{
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= e->nlevels) return -1;
    *w = e->lw[level]; *hgt = e->lh[level];
    return 0;
}
int orb_oracle_level_image(void* h, int level, uint8_t* out)
{
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= e->nlevels) return -1;
    std::memcpy(out, e->pyr[level].data(), e->pyr[level].size());
    return 0;
}
int orb_oracle_level_blurred(void* h, int level, uint8_t* out)
{
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= e->nlevels || e->blurred[level].empty()) return -1;
    std::memcpy(out, e->blurred[level].data(), e->blurred[level].size());
    return 0;
}
static int copy_kps(const std::vector<KP>& v, OracleKeyPoint* out, int cap)
{
    int n = (int)v.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = v[i];
    return n;
}
int orb_oracle_level_candidates(void* h, int level, OracleKeyPoint* out, int cap)
{
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= e->nlevels) return -1;
    return copy_kps(e->cand[level], out, cap);
}
int orb_oracle_level_keypoints(void* h, int level, OracleKeyPoint* out, int cap)
{
    Extractor* e = (Extractor*)h;
    if (level < 0 || level >= e->nlevels) return -1;
    return copy_kps(e->lkeys[level], out, cap);
}

int orb_oracle_fast(const uint8_t* img, int w, int hgt, int stride, int threshold, int nonmax, OracleKeyPoint* out, int cap)
{
    std::vector<KP> v;
    fast9_16(img, w, hgt, stride, threshold, nonmax != 0, v);
    return copy_kps(v, out, cap);
}
void orb_oracle_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride)
{
    resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orb_oracle_gaussian7(const uint8_t* src, int w, int hgt, int sstride, uint8_t* dst, int dstride)
{
    gaussian7_u8(src, w, hgt, sstride, dst, dstride);
}
void orb_oracle_gauss_taps(int ksize, double sigma, int* taps_q8) { gauss_taps_q8(ksize, sigma, taps_q8); }
float orb_oracle_fast_atan2(float y, float x) { return fast_atan2(y, x); }
void orb_oracle_sincos(float angle_rad, float* c, float* s) { sincos_f(angle_rad, c, s); }
int orb_oracle_cvround(double v) { return cv_round(v); }

void orb_oracle_sort_nodes(int* count, int* ulx, int* tag, int n)
{
    std::vector<Node> pool(n);
    std::vector<SizeNode> v(n);
    for (int i = 0; i < n; i++) {
        pool[i].ulx = ulx[i];
        pool[i].uly = tag[i];
        v[i] = std::make_pair(count[i], &pool[i]);
    }
    std::sort(v.begin(), v.end(), node_less);
    for (int i = 0; i < n; i++) {
        count[i] = v[i].first;
        ulx[i] = v[i].second->ulx;
        tag[i] = v[i].second->uly;
    }
}

// Frame::ComputeStereoMatches (reference src/Frame.cc:931-1101), SURVEY.md 8(f) rank 4: row-band candidates, best Hamming
// match, 11x11 SAD sliding window on the pyramid level of the left key point, parabola sub-pixel fit, median-based outlier
// cut.  hL / hR are the oracle extractors that produced kpsL / kpsR (their pyramids of the LAST extract call are
// mpORBextractorLeft/Right->mvImagePyramid).  kps are mvKeys / mvKeysRight (level-0 coordinates).  Outputs mvuRight, mvDepth.
int orb_oracle_stereo_matches(void* hL, void* hR, const OracleKeyPoint* kpsL, const uint8_t* descL, int N,
                              const OracleKeyPoint* kpsR, const uint8_t* descR, int Nr, float mb, float mbf,
                              float* uRight, float* depth)
{
    Extractor* eL = (Extractor*)hL;
    Extractor* eR = (Extractor*)hR;
    const int TH_HIGH = 100, TH_LOW = 50;
    for (int i = 0; i < N; i++) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = eL->lh[0];
    std::vector<std::vector<size_t> > vRowIndices(nRows);
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = kpsR[iR].y;
        const float r = 2.0f * eL->scale[kpsR[iR].octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);        // (the reference indexes unchecked)
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int> > vDistIdx;
    for (int iL = 0; iL < N; iL++) {
        const OracleKeyPoint& kpL = kpsL[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t* dL = descL + (size_t)iL * 32;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const OracleKeyPoint& kpR = kpsR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orbm_oracle_hamming(dL, descR + iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = kpsR[bestIdxR].x;
            const float scaleFactor = eL->inv_scale[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const int lw = eL->lw[kpL.octave], lhh = eL->lh[kpL.octave];
            const std::vector<uint8_t>& IL = eL->pyr[kpL.octave];
            const std::vector<uint8_t>& IRi = eR->pyr[kpL.octave];
            int bestDistS = INT32_MAX, bestincR = 0;
            float vDists[2 * 5 + 1];
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= eR->lw[kpL.octave]) continue;
            const int y0 = (int)(scaledvL - w), xl0 = (int)(scaleduL - w);
            if (y0 < 0 || y0 + 2 * w + 1 > lhh || xl0 < 0 || xl0 + 2 * w + 1 > lw) return -2;      // cv::Mat::rowRange would assert
            for (int incR = -L; incR <= +L; incR++) {
                const int xr0 = (int)(scaleduR0 + incR - w);
                if (xr0 < 0 || xr0 + 2 * w + 1 > eR->lw[kpL.octave]) return -2;
                int sad = 0;                                            // cv::norm(IL, IR, NORM_L1) on 8UC1 = integer sum
                for (int yy = 0; yy < 2 * w + 1; yy++)
                    for (int xx = 0; xx < 2 * w + 1; xx++)
                        sad += std::abs((int)IL[(size_t)(y0 + yy) * lw + xl0 + xx] - (int)IRi[(size_t)(y0 + yy) * eR->lw[kpL.octave] + xr0 + xx]);
                const float dist = (float)sad;
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = eL->scale[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                depth[iL] = mbf / disparity;
                uRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
            }
        }
    }
    if (vDistIdx.empty()) return 0;                                      // (the reference would read vDistIdx[0])
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        uRight[vDistIdx[i].second] = -1;
        depth[vDistIdx[i].second] = -1;
    }
    return (int)vDistIdx.size();
}

}  // extern "C"
