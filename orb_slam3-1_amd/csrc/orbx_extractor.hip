// orbx_extractor.hip -- host driver + C ABI of the gfx950 ORB extractor.
//
// Mirrors ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:44-109, src/ORBextractor.cc):
// constructor tables on the host (E0), then one fixed sequence of kernel launches per batch of
// frames (orbx_kernels.hip).  No CPU compute path exists: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "orbx_kernels.hip"

namespace orbx {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define ORBX_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(ORBX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static inline int cv_round_host(double v) { return (int)std::nearbyint(v); }
static inline int cv_floor_host(double v) { int i = (int)v; return i - (i > v); }
static inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

// getGaussianKernelFixedPoint_ED (OpenCV 4.x): Q8.8 taps, error diffusion, centre = 256 - 2*sum(others)
static void gauss_taps_q8(int n, double sigma, int* taps)
{
    std::vector<double> k(n);
    const double scale2x = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) { const double x = i - (n - 1) * 0.5; k[i] = std::exp(scale2x * x * x); sum += k[i]; }
    sum = 1. / sum;
    double err = 0;
    long long acc = 0;
    for (int i = 0; i < n / 2; i++) {
        const double adj = k[i] * sum * 256.0 + err;
        const long long v0 = cv_round_host(adj);
        err = adj - (double)v0;
        taps[i] = taps[n - 1 - i] = (int)v0;
        acc += v0;
    }
    taps[n / 2] = (int)(256 - 2 * acc);
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int ensure(size_t count)
    {
        if (count <= n) return ORBX_OK;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        ORBX_HIP(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
        return ORBX_OK;
    }
    int upload(const std::vector<T>& v)
    {
        int r = ensure(v.size() ? v.size() : 1);
        if (r) return r;
        if (!v.empty()) ORBX_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return ORBX_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace orbx

using namespace orbx;

struct orbx_extractor {
    int device = 0;
    int nfeatures = 0, nlevels = 0, ini_th = 0, min_th = 0;
    float scale_factor_f = 0;
    double scale_factor_d = 0;           // the reference member is a double (include/ORBextractor.h:92)
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> nfeat;
    int taps[7];
    int max_kp = 0;

    // geometry for the current image size
    int geo_w = 0, geo_h = 0;
    std::vector<LevelDesc> levels;
    std::vector<CellDesc> cells;
    std::vector<TileDesc> tiles;
    size_t pyr_frame_bytes = 0;
    int cand_frame_entries = 0, sel_frame_entries = 0;
    std::vector<StripDesc> strips;       // FAST work items: runs of adjacent cells of one cell row
    FastLds fast_layout;
    int fast_corner_cap = 0;             // 0 = kCornerCap; tests shrink it to drive the overflow path (orbx_debug_set_fast_corner_cap)
    size_t fast_lds = 0, oct_lds = 0;
    int oct_pool = 0, oct_lds_keys = 0, oct_small_keys = 0;
    bool oct_nodes_hbm = false;          // the octree's node pool does not fit LDS: it lives in d_oct_nodes
    size_t oct_node_stride = 0;
    size_t oct_small_lds = 0;
    int oct_dyn_keys = 0;                // k_octree_dyn (batches up to oct_dyn_max_batch frames whose levels' worst-case key buffers do not fit LDS): keys that do
    size_t oct_dyn_lds = 0;
    bool oct_keys_forced = false;

    DevBuf<LevelDesc> d_levels;
    DevBuf<CellDesc> d_cells;
    DevBuf<TileDesc> d_tiles;
    DevBuf<StripDesc> d_strips;
    // resize tables per destination level: per quad of columns the first source column, 4 v_perm selectors, 4 coefficient pairs
    std::vector<DevBuf<int> > d_qsx0, d_yofs;
    std::vector<DevBuf<uint4> > d_qsel, d_qalpha;
    std::vector<DevBuf<short> > d_ibeta;
    std::vector<DevBuf<int> > d_xofs;            // per-column tables of the levels that take k_resize_generic (resize_generic[l])
    std::vector<DevBuf<short> > d_ialpha;
    std::vector<uint8_t> resize_generic;

    // per-batch scratch
    int batch_cap = 0, last_batch = 0;
    DevBuf<uint8_t> d_pyr, d_blur;
    DevBuf<uint32_t> d_cand, d_scratch, d_sel;
    DevBuf<uint8_t> d_oct_nodes;
    DevBuf<int> d_cell_count, d_sel_count, d_kp_dst;
    DevBuf<OrbxKeyPoint> d_lvl_kps;
    // staging for the host-pointer API
    DevBuf<OrbxKeyPoint> d_kps;
    DevBuf<uint8_t> d_desc;
    DevBuf<int> d_n, d_mono, d_status;
    DevBuf<int> d_sad;                   // ComputeStereoMatches: SAD of the accepted matches
    DevBuf<uint8_t> d_stereo_io;         // host-buffer variant of the stereo matcher
    int out_cap = 0;
    hipStream_t stream = nullptr;
    // per-stage HIP-event profiling (off by default)
    static constexpr int kProfEvents = 8;
    bool profile = false;
    hipEvent_t prof_ev[kProfEvents] = {};
    int prof_marks = 0;
    hipStream_t prof_stream = nullptr;
    // side stream for the blur (runs beside the latency-bound octree)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // second stream for the octree of the upper pyramid levels (their node pools are small: own launch, own LDS size)
    hipStream_t oct_stream = nullptr;
    hipEvent_t ev_oct_join = nullptr, ev_fast0 = nullptr, ev_resize = nullptr;
    hipEvent_t ev_tail = nullptr;        // the resize tail's levels are written (the blur on the launch stream reads them)
    int dbg_tail_delay_us = 0;           // tests: a spin kernel in front of the resize tail (orbx_debug_set_tail_delay)
    int last_octree_variant = 0;         // which k_octree instantiation / schedule the last enqueue used (orbx_debug_last_schedule)
    int oct_split = 0, oct_pool_hi = 0;  // levels [oct_split, nlevels) go to the second launch with a pool of oct_pool_hi nodes (0: one launch)
    size_t oct_lds_hi = 0;
    // streams for the ranges a large batch is cut into (enqueue)
    int split_parts = 1;                 // ORBX_SPLIT (measurement knob, see enqueue)
    bool serial_schedule = false;        // ORBX_SERIAL=1: every kernel of a batch on the launch stream, one launch per stage (per-kernel profiles)
    int resize_tail_first = 4;           // first pyramid level of the fused resize tail (ORBX_RESIZE_TAIL; 0: a launch per level)
    bool oct_dyn_off = false;            // ORBX_OCT_DYN=0: small batches keep the scratch instantiation (measurement / tests)
    int oct_dyn_max_batch = 64;          // largest batch that takes k_octree_dyn (ORBX_OCT_DYN_MAXB; measured: one frame 0.163 -> 0.136 ms,
                                         // 64 frames 0.412 -> 0.352 ms per call, 256 frames 0.99 -> 1.12 ms: the LDS is the blur's there; LDS keys for the early level-0 launch alone: 0.993 -> 1.008)
    int oct_dyn_keys_beside = 6144;      // LDS keys per workgroup when the blur runs beside the octree (batches of 32 frames and more)
    bool resize_beside = true;           // the whole resize chain on the side stream beside FAST on level 0 (ORBX_RESIZE_BESIDE=0: in front of it)
    std::vector<hipStream_t> aux_streams;
    hipEvent_t ev_parts_fork = nullptr;
    std::vector<hipEvent_t> ev_parts_join;

    int setup_geometry(int w, int h);
    int ensure_batch(int batch);
    int enqueue(const uint8_t* d_imgs, bool level0_ready, int batch, int row_stride, size_t frame_stride,
                int lap0, int lap1, OrbxKeyPoint* o_kps, uint8_t* o_desc, int cap, int* o_n, int* o_mono, int* o_status,
                hipStream_t st);
};

// E0: constructor tables (reference src/ORBextractor.cc:409-445)
static void build_tables(orbx_extractor* e)
{
    const int nl = e->nlevels;
    e->scale.assign(nl, 1.0f); e->sigma2.assign(nl, 1.0f); e->inv_scale.resize(nl); e->inv_sigma2.resize(nl);
    for (int i = 1; i < nl; i++) {
        e->scale[i] = (float)(e->scale[i - 1] * e->scale_factor_d);
        e->sigma2[i] = e->scale[i] * e->scale[i];
    }
    for (int i = 0; i < nl; i++) {
        e->inv_scale[i] = 1.0f / e->scale[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    e->nfeat.resize(nl);
    const float factor = (float)(1.0f / e->scale_factor_d);
    float per_scale = e->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        e->nfeat[l] = (int)std::nearbyintf(per_scale);
        sum += e->nfeat[l];
        per_scale *= factor;
    }
    e->nfeat[nl - 1] = std::max(e->nfeatures - sum, 0);
    e->max_kp = 0;
    // per level the octree returns at most N + 3 nodes -- except that its FIRST pass divides every root without looking at N
    // (src/ORBextractor.cc:606-672: the size test follows the pass), so a level with a tiny budget still returns up to 4 nodes per root
    // (nIni = round(width / height) roots; 8 roots = images up to 8.5 : 1 are covered by this bound, wider ones report ORBX_ERR_CAPACITY)
    for (int l = 0; l < nl; l++) e->max_kp += std::max(e->nfeat[l] + 3, 32);
    gauss_taps_q8(7, 2.0, e->taps);
}

int orbx_extractor::setup_geometry(int w, int h)
{
    if (w == geo_w && h == geo_h) return ORBX_OK;
    levels.assign(nlevels, LevelDesc());
    cells.clear();
    tiles.clear();
    strips.clear();
    size_t off = 0;
    int cand_off = 0, sel_off = 0, max_tw = 0, max_th = 0, max_nfeat = 0;
    std::vector<int> node_need(nlevels, 0);
    d_qsx0.resize(nlevels); d_yofs.resize(nlevels); d_qsel.resize(nlevels); d_qalpha.resize(nlevels); d_ibeta.resize(nlevels);
    d_xofs.resize(nlevels); d_ialpha.resize(nlevels); resize_generic.assign(nlevels, 0);
    for (int l = 0; l < nlevels; l++) {
        LevelDesc& L = levels[l];
        L.w = (int)std::nearbyintf((float)w * inv_scale[l]);      // ComputePyramid :1175
        L.h = (int)std::nearbyintf((float)h * inv_scale[l]);
        if (L.w < 1 || L.h < 1) return fail(ORBX_ERR_ARG, "pyramid level %d of a %dx%d image is empty", l, w, h);
        if (L.w > 4096 + 32 || L.h > 4096 + 32) return fail(ORBX_ERR_ARG, "image %dx%d too large (keys are 12-bit)", w, h);
        L.stride = round_up(L.w, 64);
        L.off = (int64_t)off;
        off += (size_t)L.stride * L.h;
        off = (off + 255) & ~(size_t)255;
        L.nfeat = nfeat[l];
        L.scale = scale[l];
        L.patch_size = (int)(kPatch * scale[l]);
        {   // (see build_tables: N + 3, or what the first pass makes of the roots when the level's budget is tiny)
            const int bw = (L.w - kEdge + 3) - (kEdge - 3), bh = (L.h - kEdge + 3) - (kEdge - 3);
            const int n_ini = bh > 0 ? (int)std::roundf((float)bw / (float)bh) : 0;
            L.sel_off = sel_off; L.sel_cap = std::max({nfeat[l] + 3, 32, 4 * std::max(n_ini, 0)});
            node_need[l] = std::max(nfeat[l], 4 * std::max(n_ini, 0));         // nodes the level's list can hold at once (+ 16 below)
        }
        sel_off += (L.sel_cap + 1) & ~1;        // even: the two key points of a wave of k_orient_desc share a level
        max_nfeat = std::max(max_nfeat, node_need[l]);
        // FAST cells (ComputeKeyPointsOctTree :787-822)
        L.cell_begin = (int)cells.size();
        L.cand_off = cand_off;
        const int minBX = kEdge - 3, minBY = minBX, maxBX = L.w - kEdge + 3, maxBY = L.h - kEdge + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const int nCols = (int)(width / 35.f), nRows = (int)(height / 35.f);
        if (nCols > 0 && nRows > 0) {
            const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBY - 3) continue;
                if (maxY > maxBY) maxY = (float)maxBY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBX - 6) continue;
                    if (maxX > maxBX) maxX = (float)maxBX;
                    CellDesc c;
                    c.level = (int16_t)l;
                    c.x0 = (int16_t)iniX; c.x1 = (int16_t)maxX; c.y0 = (int16_t)iniY; c.y1 = (int16_t)maxY;
                    const int iw = c.x1 - c.x0 - 6, ih = c.y1 - c.y0 - 6;
                    if (iw <= 0 || ih <= 0) continue;       // cv::FAST finds nothing in a sub-image thinner than 7 px
                    c.slot_off = cand_off;
                    c.slot_cap = ((iw + 1) / 2) * ((ih + 1) / 2);     // 3x3 NMS: no two 8-adjacent survivors
                    cand_off += c.slot_cap;
                    max_tw = std::max(max_tw, c.x1 - c.x0);
                    max_th = std::max(max_th, c.y1 - c.y0);
                    cells.push_back(c);
                }
            }
        }
        L.cell_count = (int)cells.size() - L.cell_begin;
        L.cand_cap = cand_off - L.cand_off;
        // strips: the cells of a cell row are adjacent (interiors tile the row); cut every row into runs of at most
        // kStripMaxCells cells and about kStripWidth columns, evenly
        for (int c0 = L.cell_begin; c0 < (int)cells.size();) {
            int c1 = c0 + 1;
            while (c1 < (int)cells.size() && cells[c1].y0 == cells[c0].y0) c1++;
            const int n_row = c1 - c0;
            const int w_cell = std::max(cells[c0].x1 - cells[c0].x0 - 6, 1);
            static const int strip_width = getenv("ORBX_STRIP_WIDTH") ? std::max(48, std::min(atoi(getenv("ORBX_STRIP_WIDTH")), 250)) : kStripWidth;
            const int per = std::max(1, std::min(kStripMaxCells, (strip_width - 6) / w_cell));
            const int n_strips_row = (n_row + per - 1) / per;
            for (int k = 0; k < n_strips_row; k++) {
                const int a = c0 + (int)((long long)n_row * k / n_strips_row), b = c0 + (int)((long long)n_row * (k + 1) / n_strips_row);
                StripDesc sd;
                sd.level = (int16_t)l; sd.cell0 = (int16_t)a; sd.ncell = (int16_t)(b - a);
                sd.x0 = cells[a].x0; sd.x1 = cells[b - 1].x1; sd.y0 = cells[a].y0; sd.y1 = cells[a].y1; sd.pad = 0;
                strips.push_back(sd);
            }
            c0 = c1;
        }
        // blur tiles
        for (int y0 = 0; y0 < L.h; y0 += kBlurTH)
            for (int x0 = 0; x0 < L.w; x0 += kBlurTW) {
                TileDesc t; t.level = (int16_t)l; t.x0 = (int16_t)x0; t.y0 = (int16_t)y0; t.pad = 0;
                tiles.push_back(t);
            }
        // resize tables (cv::resize INTER_LINEAR 8UC1); level l is made from level l-1
        if (l > 0) {
            const LevelDesc& P = levels[l - 1];
            const double scale_x = 1. / ((double)L.w / P.w), scale_y = 1. / ((double)L.h / P.h);
            const int wpad = round_up(L.w, 4);          // k_resize reads 4 entries per 16-B load
            std::vector<int> xofs(wpad, 0), yofs(L.h);
            std::vector<short> ia((size_t)wpad * 2, 0), ib((size_t)L.h * 2);
            auto sat = [](float v) { int iv = (int)std::nearbyintf(v); return (short)std::min(std::max(iv, -32768), 32767); };
            for (int dx = 0; dx < L.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = cv_floor_host(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= P.w - 1) { fx = 0; sx = P.w - 1; }
                xofs[dx] = sx;
                ia[2 * dx] = sat((1.f - fx) * 2048);
                ia[2 * dx + 1] = sat(fx * 2048);
            }
            for (int dy = 0; dy < L.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = cv_floor_host(fy);
                fy -= sy;
                yofs[dy] = sy;
                ib[2 * dy] = sat((1.f - fy) * 2048);
                ib[2 * dy + 1] = sat(fy * 2048);
            }
            // per quad of destination columns (k_resize): sx0, selectors of (S[sx], S[sx+1]) inside the 8 bytes from sx0 on,
            // coefficient pairs; columns past the level's width get coefficient 0 (they land in the row padding as 0)
            const int nq = wpad / 4;
            std::vector<int> qsx0(nq);
            std::vector<uint4> qsel(nq), qal(nq);
            for (int q = 0; q < nq; q++) {
                qsx0[q] = xofs[4 * q];
                uint32_t se[4], al[4];
                for (int k = 0; k < 4; k++) {
                    const int dx = 4 * q + k;
                    if (dx < L.w) {
                        const int o = xofs[dx] - qsx0[q];
                        if (o < 0) return fail(ORBX_ERR_INTERNAL, "resize table: source offset %d", o);
                        if (o > 6) { resize_generic[l] = 1; se[k] = 0x0C0C0C0Cu; al[k] = 0u; continue; }     // scale factors >= 2: the level takes k_resize_generic
                        se[k] = (uint32_t)o | 0x0C000C00u | ((uint32_t)(o + 1) << 16);
                        al[k] = (uint32_t)(uint16_t)ia[2 * dx] | ((uint32_t)(uint16_t)ia[2 * dx + 1] << 16);
                    } else { se[k] = 0x0C0C0C0Cu; al[k] = 0u; }
                }
                qsel[q] = make_uint4(se[0], se[1], se[2], se[3]);
                qal[q] = make_uint4(al[0], al[1], al[2], al[3]);
            }
            int r;
            if (resize_generic[l] && ((r = d_xofs[l].upload(xofs)) || (r = d_ialpha[l].upload(ia)))) return r;
            if ((r = d_qsx0[l].upload(qsx0)) || (r = d_yofs[l].upload(yofs)) || (r = d_qsel[l].upload(qsel)) || (r = d_qalpha[l].upload(qal)) ||
                (r = d_ibeta[l].upload(ib))) return r;
        }
    }
    pyr_frame_bytes = off;
    cand_frame_entries = std::max(cand_off, 1);
    sel_frame_entries = sel_off;
    // FAST kernel LDS: sized for the largest strip (orbx_fast_strips.inc)
    {
        if (cells.size() > 32767) return fail(ORBX_ERR_ARG, "%zu FAST cells exceed the strip table's 16-bit cell index", cells.size());
        int max_nch = 1, max_ngx = 1, max_sth = 7, max_ent = 1, max_ncell = 1, max_int = 1;
        for (const StripDesc& sd : strips) {
            const int g0 = (sd.x0 + 3) >> 2, g1 = (sd.x1 - 4) >> 2, ngx = g1 - g0 + 1, sth = sd.y1 - sd.y0, ih = sth - 6;
            if (ngx > 64 || ih > 127) return fail(ORBX_ERR_ARG, "FAST strip of %dx%d px exceeds the 64-quad x 127-row work item index", sd.x1 - sd.x0, sth);
            const int xa = (4 * (g0 - 1)) & ~15, qoff = 4 * g0 - xa;
            max_nch = std::max(max_nch, (qoff + 4 * ngx + 4 + 15) >> 4);
            for (int j = 0; j < sd.ncell; j++) { const CellDesc& c = cells[sd.cell0 + j]; max_int = std::max(max_int, (c.x1 - c.x0 - 6) * ih); }
            max_ngx = std::max(max_ngx, ngx); max_sth = std::max(max_sth, sth);
            max_ent = std::max(max_ent, ((ih + 3) / 4) * ngx);
            max_ncell = std::max(max_ncell, (int)sd.ncell);
        }
        if (max_int > 8191) return fail(ORBX_ERR_ARG, "FAST cell interior of %d px exceeds the 8191-px bitmask", max_int);
        FastLds& Z = fast_layout;
        auto al16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
        size_t o = 0;
        Z.tile_pitch = 16 * max_nch;
        o = al16((size_t)max_sth * Z.tile_pitch);
        Z.map_off = (int)o; Z.map_pitch = 4 * max_ngx + 8;
        o = al16(o + (size_t)(max_sth - 6 + 2) * Z.map_pitch);
        Z.colcell_off = (int)o; o = al16(o + Z.map_pitch);
        Z.ent_cap = (max_ent + 7) & ~7;
        Z.ent_off = (int)o; o = al16(o + 2 * 4 * (size_t)Z.ent_cap);
        Z.px_off = (int)o; o = al16(o + 2 * 4 * (size_t)kPxCap);
        Z.corner_cap = fast_corner_cap > 0 ? fast_corner_cap : kCornerCap;
        Z.corner_off = (int)o; o = al16(o + 2 * 4 * (size_t)Z.corner_cap);
        Z.wpc = (max_int + 31) / 32;
        Z.pre_off = (int)o; o = al16(o + 4 * (size_t)max_ncell * Z.wpc);
        Z.bits_off = (int)o; o = al16(o + 4 * (size_t)max_ncell * Z.wpc);
        Z.total = (int)o;
        fast_lds = o;
        if (fast_lds > 150 * 1024) return fail(ORBX_ERR_ARG, "FAST strip of %d quads x %d rows does not fit LDS", max_ngx, max_sth);
    }
    // octree kernel LDS
    oct_pool = max_nfeat + 16;
    if (oct_pool > 32000) return fail(ORBX_ERR_ARG, "nfeatures per level %d too large for the device octree (16-bit node ids)", max_nfeat);
    const char* env = getenv("ORBX_OCT_LDS_KEYS");
    oct_lds_keys = env ? atoi(env) : 0;      // measured on MI355X: L2-resident HBM scratch + more resident waves beats LDS keys
    oct_keys_forced = env != nullptr;        // ... for large batches; small batches are latency-bound and take LDS keys (enqueue())
    auto pool_bytes = [](int pool) { return (size_t)pool * (2 * sizeof(SortNode) + 8 + 16 + 2 * kOctLogFactor) + (size_t)((pool + 15) & ~15); };
    const size_t node_bytes = pool_bytes(oct_pool);
    // Large batches run the octree beside the blur, and a workgroup's LDS is sized by its launch: with one launch every level
    // would hold the node pool of the largest one (8 levels x 14 KB per frame and CU leave the blur no LDS to run in).  The upper
    // levels -- fewer features each -- therefore get a launch of their own with a pool of their size; the split that minimises
    // the LDS per frame is taken.
    oct_split = 0; oct_pool_hi = 0; oct_lds_hi = 0;
    {
        size_t best = (size_t)nlevels * node_bytes;
        for (int sp = 1; sp < nlevels; sp++) {
            int hi = 0;
            for (int l = sp; l < nlevels; l++) hi = std::max(hi, node_need[l]);
            const size_t cost = (size_t)sp * node_bytes + (size_t)(nlevels - sp) * pool_bytes(hi + 16);
            if (cost < best) { best = cost; oct_split = sp; oct_pool_hi = hi + 16; oct_lds_hi = pool_bytes(hi + 16) + 16; }
        }
    }
    // the node pool of a level lives in LDS (about 2500 nodes); a level with more features takes the instantiation whose pool
    // lives in an HBM scratch (ensure_batch allocates it)
    oct_nodes_hbm = node_bytes > 150 * 1024;
    oct_node_stride = (node_bytes + 255) & ~(size_t)255;
    if (oct_nodes_hbm) oct_lds_keys = 0;
    if (!oct_nodes_hbm && node_bytes + 8 * (size_t)oct_lds_keys > 150 * 1024) oct_lds_keys = (int)((150 * 1024 - node_bytes) / 8);
    if (oct_lds_keys < 0) oct_lds_keys = 0;
    oct_lds = node_bytes + 8 * (size_t)oct_lds_keys + 16;
    // second configuration for small batches: both key buffers of a level (up to 4096 candidates) live in LDS
    int max_cand_cap = 0;
    for (const LevelDesc& lv : levels) max_cand_cap = std::max(max_cand_cap, lv.cand_cap);
    oct_small_keys = max_cand_cap;
    if (oct_nodes_hbm || node_bytes + 8 * (size_t)oct_small_keys > 150 * 1024) oct_small_keys = 0;      // does not fit: small batches use the scratch path too
    oct_small_lds = node_bytes + 8 * (size_t)oct_small_keys + 16;
    if (oct_lds_keys > 0 && oct_lds_keys < max_cand_cap) oct_lds_keys = 0;              // the LDS instantiation needs room for a whole level
    // third configuration, for the smallest batches when a level's worst case does NOT fit: as many keys as LDS holds beside the
    // node pool; the kernel takes the LDS body for every (level, frame) whose actual candidates fit (k_octree_dyn)
    oct_dyn_keys = 0; oct_dyn_lds = 0;
    if (!oct_nodes_hbm && oct_small_keys == 0 && node_bytes + 8 * 2048 <= 150 * 1024) {
        oct_dyn_keys = (int)std::min<size_t>((size_t)max_cand_cap, (150 * 1024 - node_bytes) / 8);
        if (const char* envd = getenv("ORBX_OCT_DYN_KEYS")) oct_dyn_keys = std::max(64, std::min(oct_dyn_keys, atoi(envd)));     // (tests: force the fallback body)
        oct_dyn_lds = node_bytes + 8 * (size_t)oct_dyn_keys + 16;
    }
    if (!oct_nodes_hbm) {
        ORBX_HIP(hipFuncSetAttribute((const void*)k_octree<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)node_bytes + 64));
        ORBX_HIP(hipFuncSetAttribute((const void*)k_octree<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(oct_lds, oct_small_lds)));
        if (oct_dyn_keys > 0) ORBX_HIP(hipFuncSetAttribute((const void*)k_octree_dyn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oct_dyn_lds));
    } else {
        oct_lds = 64; oct_small_lds = 64;
    }
    ORBX_HIP(hipFuncSetAttribute((const void*)k_fast_strips, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast_lds));
    int r;
    if ((r = d_levels.upload(levels)) || (r = d_cells.upload(cells)) || (r = d_tiles.upload(tiles)) || (r = d_strips.upload(strips))) return r;
    geo_w = w; geo_h = h;
    batch_cap = 0;      // scratch must be re-sized for the new geometry
    return ORBX_OK;
}

int orbx_extractor::ensure_batch(int batch)
{
    if (batch <= batch_cap) return ORBX_OK;
    int r;
    const size_t B = (size_t)batch;
    // + 64: the 16-byte patch chunks of k_orient_desc may start up to 45 bytes right of a key point's last patch column
    if ((r = d_pyr.ensure(B * pyr_frame_bytes + 64)) || (r = d_blur.ensure(B * pyr_frame_bytes + 64)) ||
        (r = d_cand.ensure(B * cand_frame_entries)) || (r = d_scratch.ensure(B * 2 * cand_frame_entries)) ||
        (r = d_sel.ensure(B * sel_frame_entries)) || (r = d_cell_count.ensure(B * std::max<size_t>(cells.size(), 1))) ||
        (r = d_sel_count.ensure(B * nlevels)) || (r = d_kp_dst.ensure(B * sel_frame_entries)) ||
        (r = d_lvl_kps.ensure(B * sel_frame_entries)) || (r = d_n.ensure(B)) || (r = d_mono.ensure(B)) || (r = d_status.ensure(B)))
        return r;
    if (oct_nodes_hbm && (r = d_oct_nodes.ensure(B * nlevels * oct_node_stride))) return r;
    // the pyramid row padding is read by dword loads at row ends; keep it defined
    ORBX_HIP(hipMemset(d_pyr.p, 0, B * pyr_frame_bytes + 64));
    ORBX_HIP(hipMemset(d_blur.p, 0, B * pyr_frame_bytes + 64));
    batch_cap = batch;
    return ORBX_OK;
}

namespace orbx {
// test aid: keeps a stream busy for `ticks` of the 100 MHz wall clock (orbx_debug_set_tail_delay)
__global__ void k_debug_spin(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
// copies B frames of arbitrary row stride into pyramid level 0 (dword stores, byte loads)
__global__ __launch_bounds__(256) void k_copy_level0(const uint8_t* __restrict__ src, int row_stride, size_t src_frame_stride,
                                                     uint8_t* __restrict__ pyr, size_t frame_stride, LevelDesc L)
{
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (x4 >= L.w) return;
    const uint8_t* s = src + (size_t)blockIdx.z * src_frame_stride + (size_t)y * row_stride + x4;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) if (x4 + k < L.w) v |= (uint32_t)s[k] << (8 * k);
    *(uint32_t*)(pyr + (size_t)blockIdx.z * frame_stride + L.off + (size_t)y * L.stride + x4) = v;
}
}  // namespace orbx

int orbx_extractor::enqueue(const uint8_t* d_imgs, bool level0_ready, int batch, int row_stride, size_t frame_stride,
                            int lap0, int lap1, OrbxKeyPoint* o_kps, uint8_t* o_desc, int cap, int* o_n, int* o_mono, int* o_status,
                            hipStream_t st)
{
    const int B = batch;
    ORBX_HIP(hipMemsetAsync(o_status, 0, sizeof(int) * B, st));
    // optional per-stage timing with HIP events on the launch stream (bench.py roofline leg)
    int mark_i = 0;
    auto mark = [&]() { if (profile && mark_i < kProfEvents) (void)hipEventRecord(prof_ev[mark_i++], st); };
#define ORBX_LAUNCHED(name) do { const hipError_t le_ = hipGetLastError(); if (le_ != hipSuccess) return fail(ORBX_ERR_HIP, "launch of %s: %s", name, hipGetErrorString(le_)); } while (0)
    // Level 0 is the caller's image.  When its rows can be read with aligned 16-byte loads it is used in place: the resize to
    // level 1, FAST and the blur read it where it lies, and the blur -- which touches every pixel anyway -- leaves the copy in
    // the pyramid that the later readers of level 0 (descriptors, stereo matching, orbx_pyramid_level) use.  Otherwise it is
    // copied first (k_copy_level0), as it is when the host-pointer entry points have uploaded it into the pyramid themselves.
    const bool in_place = !level0_ready && ((uintptr_t)d_imgs % 16 == 0) && (row_stride % 16 == 0) && (frame_stride % 16 == 0);
    const int n_cells = (int)cells.size();
    const int quads = (sel_frame_entries + 7) / 8;       // workgroups of k_orient_desc per frame: 8 key points each (two per wave)

    // One range of frames [f0, f0 + nB) through the whole pipeline on stream `s`; the blur either on `blur_s` beside the octree
    // (a single range: the octree is one wave per frame and level and leaves most of the chip idle) or on `s` itself.
    auto run_range = [&](int f0, int nB, hipStream_t s, hipStream_t blur_s, bool marks) -> int {
        uint8_t* pyr = d_pyr.p + (size_t)f0 * pyr_frame_bytes;
        uint8_t* blr = d_blur.p + (size_t)f0 * pyr_frame_bytes;
        SrcImage lvl0;
        lvl0.base = nullptr; lvl0.frame_stride = 0; lvl0.stride = 0; lvl0.w = levels[0].w; lvl0.h = levels[0].h;
        if (in_place) { lvl0.base = d_imgs + (size_t)f0 * frame_stride; lvl0.frame_stride = frame_stride; lvl0.stride = row_stride; }
        if (marks) mark();
        if (!level0_ready && !in_place) {
            const LevelDesc& L0 = levels[0];
            dim3 g((L0.w / 4 + 255) / 256, L0.h, nB);
            hipLaunchKernelGGL(k_copy_level0, g, dim3(256), 0, s, d_imgs + (size_t)f0 * frame_stride, row_stride, frame_stride, pyr, pyr_frame_bytes, L0);
            ORBX_LAUNCHED("k_copy_level0");
        }
        if (marks) mark();
        // (resize_tail_first: the levels from there on are one launch, a workgroup per frame -- k_resize_tail)
        // (only with many frames: a single workgroup walks a small batch's levels slower than one wide launch per level does)
        int l_tail = (nB >= 64 && resize_tail_first >= 2 && resize_tail_first < nlevels) ? resize_tail_first : nlevels;
        for (int l = l_tail; l < nlevels; l++) if (resize_generic[l]) l_tail = nlevels;      // (the fused tail only knows the table-driven kernel)
        // FAST on level 0 needs no resized level, so the whole resize chain (levels 1 .. and the tail) runs on the side stream beside
        // it; `s` waits for the levels before the tail (ev_resize) in front of FAST on levels 1 ..  (1.005 -> 0.991 ms per 256-frame
        // step.  Measured and dropped in the same session: the blur on a stream of its own right behind the resize chain, beside
        // FAST -- two issue-bound whole-chip kernels side by side: 1.10 ms.)
        const bool chain_beside = resize_beside && blur_s && oct_stream && n_cells > 0 && l_tail < nlevels && l_tail > 1 &&
                                  !strips.empty() && strips[0].level == 0;
        hipStream_t rs = chain_beside ? oct_stream : s;
        if (chain_beside) { ORBX_HIP(hipEventRecord(ev_fork, s)); ORBX_HIP(hipStreamWaitEvent(rs, ev_fork, 0)); }
        for (int l = 1; l < l_tail; l++) {
            const LevelDesc& D = levels[l];
            dim3 g(xcd_grid(((D.w + kResizeTW - 1) / kResizeTW) * ((D.h + kResizeRows - 1) / kResizeRows)), nB);
            const LevelDesc& P = levels[l - 1];
            SrcImage src;
            src.base = pyr + P.off; src.frame_stride = pyr_frame_bytes; src.stride = P.stride; src.w = P.w; src.h = P.h;
            if (l == 1 && in_place) src = lvl0;
            if (resize_generic[l])
                hipLaunchKernelGGL(k_resize_generic, dim3((D.w + 255) / 256, D.h, nB), dim3(256), 0, rs, src, pyr, pyr_frame_bytes, D,
                                   d_xofs[l].p, d_ialpha[l].p, d_yofs[l].p, d_ibeta[l].p);
            else
            hipLaunchKernelGGL(k_resize, g, dim3(256), 0, rs, src, pyr, pyr_frame_bytes, D,
                               d_qsx0[l].p, d_qsel[l].p, d_qalpha[l].p, d_yofs[l].p, d_ibeta[l].p);
            ORBX_LAUNCHED("k_resize");
        }
        if (chain_beside) ORBX_HIP(hipEventRecord(ev_resize, rs));
        // The octree is one wave per (level, frame).  With few frames the chip is empty anyway and a wave's latency is the
        // whole stage: keep the ping-pong key buffers in LDS (no L2 round trip per DivideNode).  With many frames the larger
        // LDS footprint would halve the resident waves, and the L2-resident scratch wins.
        const bool small_batch = !oct_keys_forced && oct_small_keys > 0 && (long long)B * nlevels <= 512;
        // (one wave per level and frame: below 32 frames every workgroup has a CU of its own, LDS is free)
        const bool dyn_keys = !oct_keys_forced && !small_batch && oct_dyn_keys > 0 && B <= oct_dyn_max_batch && !oct_dyn_off;
        // (with the blur beside the octree -- 32 frames and more -- a workgroup takes a smaller share of its CU's LDS)
        const int dyn_k = (blur_s && B >= 32) ? std::min(oct_dyn_keys, oct_dyn_keys_beside) : oct_dyn_keys;
        const size_t o_lds = small_batch ? oct_small_lds : (dyn_keys ? oct_dyn_lds - 8 * (size_t)(oct_dyn_keys - dyn_k) : oct_lds);
        const int o_keys = small_batch ? oct_small_keys : (dyn_keys ? dyn_k : oct_lds_keys);
        uint32_t* sel = d_sel.p + (size_t)f0 * sel_frame_entries;
        int* sel_cnt = d_sel_count.p + (size_t)f0 * nlevels;
        int* kp_dst = d_kp_dst.p + (size_t)f0 * sel_frame_entries;
        auto oct_kernel = oct_nodes_hbm ? k_octree<false, false> : (dyn_keys ? k_octree_dyn : (o_keys > 0 ? k_octree<true, true> : k_octree<false, true>));
        // (keys in LDS: one launch, the buffers are sized for the largest level anyway)
        const bool two_launches = !oct_nodes_hbm && (o_keys == 0 || dyn_keys) && oct_split > 0 && oct_stream && blur_s;
        const int lv_lo = two_launches ? oct_split : nlevels;
        auto launch_octree = [&](hipStream_t qs, int lv0, int lv1, size_t lds, int pool, int keys) {
            if (lv1 > lv0)
                hipLaunchKernelGGL(oct_kernel, dim3(lv1 - lv0, nB), dim3(64), lds, qs, d_levels.p, d_cells.p, d_cand.p + (size_t)f0 * cand_frame_entries, (size_t)cand_frame_entries,
                                   d_cell_count.p + (size_t)f0 * std::max<size_t>(cells.size(), 1), n_cells, d_scratch.p + (size_t)f0 * 2 * cand_frame_entries, (size_t)2 * cand_frame_entries, pool, keys,
                                   sel, sel_frame_entries, sel_cnt, nlevels, o_status + f0,
                                   d_oct_nodes.p ? d_oct_nodes.p + (size_t)f0 * nlevels * oct_node_stride : nullptr, oct_node_stride, lv0);
        };
        auto launch_blur = [&](hipStream_t bs) {
            hipLaunchKernelGGL(k_blur, dim3(xcd_grid((int)tiles.size()), nB), dim3(256), 0, bs, lvl0, pyr, in_place ? pyr : nullptr, blr, pyr_frame_bytes,
                               d_levels.p, d_tiles.p, (int)tiles.size(), taps[0], taps[1], taps[2], taps[3]);
        };
        bool oct0_early = false;            // level 0's octree already runs beside FAST on the levels above it (below)
        bool tail_beside = false;           // the resize tail ran on oct_stream: ev_tail orders its levels before the blur
        uint32_t* cand = d_cand.p + (size_t)f0 * cand_frame_entries;
        int* cell_cnt = d_cell_count.p + (size_t)f0 * std::max<size_t>(cells.size(), 1);
        auto launch_fast = [&](hipStream_t fs, int s0, int s1) {
            if (s1 > s0)
                hipLaunchKernelGGL(k_fast_strips, dim3(xcd_grid(s1 - s0), nB), dim3(256), fast_lds, fs, lvl0, pyr, pyr_frame_bytes, d_levels.p, d_cells.p,
                                   d_strips.p + s0, s1 - s0, n_cells, ini_th, min_th, fast_layout, cand, (size_t)cand_frame_entries, cell_cnt);
        };
        // first strip of the levels the resize tail produces (the strips are in level order)
        int tail_strip = (int)strips.size();
        for (int si = (int)strips.size() - 1; si >= 0 && strips[si].level >= l_tail; si--) tail_strip = si;
        if (l_tail < nlevels) {
            ResizeTables T;
            std::memset(&T, 0, sizeof(T));
            for (int l = l_tail; l < nlevels; l++) { T.q_sx0[l] = d_qsx0[l].p; T.q_sel[l] = d_qsel[l].p; T.q_alpha[l] = d_qalpha[l].p; T.yofs[l] = d_yofs[l].p; T.ibeta[l] = d_ibeta[l].p; }
            // With side streams the tail -- one latency-bound workgroup per frame -- and FAST on its levels run BESIDE FAST on the
            // lower levels (which only need the levels before the tail) instead of in front of it.
            const bool beside = blur_s && oct_stream && n_cells > 0 && tail_strip > 0;
            hipStream_t ts = beside ? oct_stream : s;
            if (beside && !chain_beside) { ORBX_HIP(hipEventRecord(ev_fork, s)); ORBX_HIP(hipStreamWaitEvent(ts, ev_fork, 0)); }
            if (dbg_tail_delay_us > 0) hipLaunchKernelGGL(k_debug_spin, dim3(1), dim3(64), 0, ts, (long long)dbg_tail_delay_us * 100);
            hipLaunchKernelGGL(k_resize_tail, dim3(nB), dim3(1024), 0, ts, pyr, pyr_frame_bytes, d_levels.p, T, l_tail, nlevels);
            ORBX_LAUNCHED("k_resize_tail");
            if (beside) {
                // the blur runs on `s` and reads EVERY level: it has to wait for the tail's levels (FAST on them may still run)
                ORBX_HIP(hipEventRecord(ev_tail, ts));
                tail_beside = true;
                launch_fast(ts, tail_strip, (int)strips.size());
                ORBX_HIP(hipEventRecord(ev_oct_join, ts));
                // FAST on level 0 first; its octree -- the longest chain of the stage -- then runs beside FAST on levels 1..
                // (a launch per level with every level's octree started early was measured too: the extra launch tails cost
                // FAST 40 us, more than the octrees gain)
                int l1_strip = tail_strip;
                for (int si = tail_strip - 1; si >= 0 && strips[si].level >= 1; si--) l1_strip = si;
                if (two_launches && oct_split > 1 && l1_strip > 0 && l1_strip < tail_strip) {
                    launch_fast(s, 0, l1_strip);
                    ORBX_HIP(hipEventRecord(ev_fast0, s));
                    ORBX_HIP(hipStreamWaitEvent(blur_s, ev_fast0, 0));
                    launch_octree(blur_s, 0, 1, o_lds, oct_pool, o_keys);
                    oct0_early = true;
                    if (chain_beside) ORBX_HIP(hipStreamWaitEvent(s, ev_resize, 0));
                    launch_fast(s, l1_strip, tail_strip);
                } else if (chain_beside && l1_strip > 0 && l1_strip < tail_strip) {
                    launch_fast(s, 0, l1_strip);
                    ORBX_HIP(hipStreamWaitEvent(s, ev_resize, 0));
                    launch_fast(s, l1_strip, tail_strip);
                } else {
                    if (chain_beside) ORBX_HIP(hipStreamWaitEvent(s, ev_resize, 0));
                    launch_fast(s, 0, tail_strip);
                }
                ORBX_HIP(hipStreamWaitEvent(blur_s, ev_oct_join, 0));
            } else {
                if (marks) mark();
                if (n_cells > 0) launch_fast(s, 0, (int)strips.size());
            }
        } else {
            if (marks) mark();
            if (n_cells > 0) launch_fast(s, 0, (int)strips.size());
        }
        ORBX_LAUNCHED("k_fast_strips");
        if (marks) mark();
        // With a side stream the throughput-bound blur stays on the launch stream (it starts the moment FAST ends) and the
        // latency-bound octree + index go beside it: octree of the lower levels and the index on `blur_s`, the upper levels'
        // octree on `oct_stream`.  Without one (profiling) everything is serial on `s`.
        hipStream_t os = blur_s ? blur_s : s;
        if (blur_s) {
            ORBX_HIP(hipEventRecord(ev_fork, s));
            ORBX_HIP(hipStreamWaitEvent(blur_s, ev_fork, 0));
        }
        if (two_launches) {
            ORBX_HIP(hipStreamWaitEvent(oct_stream, ev_fork, 0));
            {   // (dynamic LDS keys: the upper levels hold fewer candidates -- half the allotment)
                const int k_hi = dyn_keys ? std::max(64, o_keys / 2) : 0;
                launch_octree(oct_stream, oct_split, nlevels, oct_lds_hi + 8 * (size_t)k_hi, oct_pool_hi, k_hi);
            }
            ORBX_HIP(hipEventRecord(ev_oct_join, oct_stream));
        }
        launch_octree(os, oct0_early ? 1 : 0, lv_lo, o_lds, oct_pool, o_keys);
        if (blur_s) {
            if (tail_beside) ORBX_HIP(hipStreamWaitEvent(s, ev_tail, 0));
            launch_blur(s);
        }
        last_octree_variant = (oct_nodes_hbm ? 0 : (dyn_keys ? 3 : (o_keys > 0 ? 1 : 2))) | (two_launches ? 4 : 0) | (oct0_early ? 8 : 0) | (in_place ? 16 : 0) | (tail_beside ? 32 : 0) | (chain_beside ? 64 : 0);
        if (two_launches) ORBX_HIP(hipStreamWaitEvent(os, ev_oct_join, 0));
        ORBX_LAUNCHED("k_octree / k_blur");
        if (marks) mark();
        hipLaunchKernelGGL(k_index, dim3(nB), dim3(64), 0, os, d_levels.p, nlevels, sel, sel_frame_entries, sel_cnt,
                           lap0, lap1, cap, kp_dst, sel_frame_entries, o_n + f0, o_mono + f0, o_status + f0);
        ORBX_LAUNCHED("k_index");
        if (marks) mark();
        if (blur_s) { ORBX_HIP(hipEventRecord(ev_join, blur_s)); ORBX_HIP(hipStreamWaitEvent(s, ev_join, 0)); }
        else launch_blur(s);
        if (marks) mark();
        hipLaunchKernelGGL(k_orient_desc, dim3((unsigned)(((long long)quads * nB + 7) / 8 * 8)), dim3(256), 0, s, pyr, blr, pyr_frame_bytes,
                           d_levels.p, nlevels, sel, sel_frame_entries, sel_cnt, kp_dst, sel_frame_entries,
                           o_kps + (size_t)f0 * cap, o_desc + (size_t)f0 * cap * 32, cap, d_lvl_kps.p + (size_t)f0 * sel_frame_entries, quads, nB);
        ORBX_LAUNCHED("k_orient_desc");
        if (marks) mark();
        return ORBX_OK;
    };

    // Optionally (ORBX_SPLIT=2..4) a large batch is cut into ranges, each on a stream of its own, so that one range's latency-bound
    // stages (octree, the dependent resize launches) could hide behind the others' FAST / blur / descriptors.  Measured on MI355X
    // (B = 256, round 2): 1 range 1.344 ms per step, 2 ranges 1.348, 3 ranges 1.637, 4 ranges 1.649 -- the blur beside the octree
    // already fills the idle chip and whole-chip kernels of different streams do not co-run, so the default stays ONE range.
    int parts = 1;
    if (!profile && split_parts > 1 && B >= 32 * split_parts) parts = std::min(split_parts, 1 + (int)aux_streams.size());
    int r;
    if (parts == 1) {
        // (small batches: the blur is a few microseconds, less than the fork / join across streams costs)
        if ((r = run_range(0, B, st, (!profile && !serial_schedule && side_stream && B >= 32) ? side_stream : nullptr, true))) return r;
    } else {
        ORBX_HIP(hipEventRecord(ev_parts_fork, st));
        for (int p = 0; p < parts; p++) {
            const int f0 = (int)((long long)B * p / parts), f1 = (int)((long long)B * (p + 1) / parts);
            hipStream_t s = p == 0 ? st : aux_streams[p - 1];
            if (p > 0) ORBX_HIP(hipStreamWaitEvent(s, ev_parts_fork, 0));
            if ((r = run_range(f0, f1 - f0, s, nullptr, false))) return r;
            if (p > 0) { ORBX_HIP(hipEventRecord(ev_parts_join[p - 1], s)); ORBX_HIP(hipStreamWaitEvent(st, ev_parts_join[p - 1], 0)); }
        }
    }
    prof_marks = mark_i;
    prof_stream = st;
    ORBX_HIP(hipGetLastError());
    last_batch = B;
    return ORBX_OK;
}

extern "C" {

// Per-stage device time of the LAST enqueue, measured with HIP events recorded on the launch stream.
// stages: 0 level-0 copy, 1 pyramid resize (all levels), 2 FAST cells, 3 octree, 4 index, 5 blur, 6 orient+descriptor.
int orbx_profile_enable(orbx_extractor* e, int on)
{
    if (!e) return fail(ORBX_ERR_ARG, "NULL handle");
    ORBX_HIP(hipSetDevice(e->device));
    if (on && !e->prof_ev[0])
        for (int i = 0; i < orbx_extractor::kProfEvents; i++) ORBX_HIP(hipEventCreate(&e->prof_ev[i]));
    e->profile = on != 0;
    e->prof_marks = 0;
    return ORBX_OK;
}

int orbx_profile_read(orbx_extractor* e, float* stage_ms, int n_stages)
{
    if (!e || !stage_ms) return fail(ORBX_ERR_ARG, "NULL argument");
    if (!e->profile || e->prof_marks < 2) return fail(ORBX_ERR_ARG, "profiling was not enabled for the last call");
    ORBX_HIP(hipSetDevice(e->device));
    ORBX_HIP(hipStreamSynchronize(e->prof_stream));
    for (int i = 0; i < n_stages; i++) {
        stage_ms[i] = 0.f;
        if (i + 1 < e->prof_marks) ORBX_HIP(hipEventElapsedTime(&stage_ms[i], e->prof_ev[i], e->prof_ev[i + 1]));
    }
    return ORBX_OK;
}

const char* orbx_last_error(void) { return g_last_error.c_str(); }

int orbx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int orbx_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast, int device, orbx_extractor** out)
{
    if (!out) return fail(ORBX_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (nlevels < 1 || nlevels > kMaxLevels || nfeatures < 1 || !(scale_factor > 1.0f))
        return fail(ORBX_ERR_ARG, "bad extractor parameters (nfeatures=%d scale=%f nlevels=%d)", nfeatures, scale_factor, nlevels);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ORBX_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(ORBX_ERR_ARG, "device %d out of range (%d devices)", device, ndev);
    ORBX_HIP(hipSetDevice(device));
    orbx_extractor* e = new orbx_extractor();
    e->device = device;
    e->nfeatures = nfeatures; e->nlevels = nlevels; e->ini_th = ini_th_fast; e->min_th = min_th_fast;
    e->scale_factor_f = scale_factor; e->scale_factor_d = (double)scale_factor;
    build_tables(e);
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { delete e; return fail(ORBX_ERR_HIP, "stream create failed"); }
    if (hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_parts_fork, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&e->oct_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_oct_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_tail, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_resize, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_fast0, hipEventDisableTiming) != hipSuccess) { orbx_destroy(e); return fail(ORBX_ERR_HIP, "side stream create failed"); }
    if (const char* env = getenv("ORBX_SERIAL")) e->serial_schedule = atoi(env) != 0;
    if (const char* env = getenv("ORBX_SPLIT")) e->split_parts = std::max(1, std::min(atoi(env), 4));
    if (const char* env = getenv("ORBX_RESIZE_TAIL")) e->resize_tail_first = atoi(env);
    if (const char* env = getenv("ORBX_RESIZE_BESIDE")) e->resize_beside = atoi(env) != 0;
    if (const char* env = getenv("ORBX_OCT_DYN")) e->oct_dyn_off = atoi(env) == 0;
    if (const char* env = getenv("ORBX_OCT_DYN_MAXB")) e->oct_dyn_max_batch = atoi(env);
    if (const char* env = getenv("ORBX_OCT_DYN_KEYS_BESIDE")) e->oct_dyn_keys_beside = std::max(64, atoi(env));
    for (int i = 0; i + 1 < e->split_parts; i++) {
        hipStream_t s = nullptr; hipEvent_t ev = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            if (s) (void)hipStreamDestroy(s);
            break;      // fewer ranges then
        }
        e->aux_streams.push_back(s); e->ev_parts_join.push_back(ev);
    }
    *out = e;
    return ORBX_OK;
}

void orbx_destroy(orbx_extractor* e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) { (void)hipStreamSynchronize(e->stream); (void)hipStreamDestroy(e->stream); }
    if (e->side_stream) { (void)hipStreamSynchronize(e->side_stream); (void)hipStreamDestroy(e->side_stream); }
    if (e->oct_stream) { (void)hipStreamSynchronize(e->oct_stream); (void)hipStreamDestroy(e->oct_stream); }
    if (e->ev_oct_join) (void)hipEventDestroy(e->ev_oct_join);
    if (e->ev_fast0) (void)hipEventDestroy(e->ev_fast0);
    if (e->ev_resize) (void)hipEventDestroy(e->ev_resize);
    if (e->ev_tail) (void)hipEventDestroy(e->ev_tail);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    for (hipStream_t s : e->aux_streams) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipEvent_t ev : e->ev_parts_join) (void)hipEventDestroy(ev);
    if (e->ev_parts_fork) (void)hipEventDestroy(e->ev_parts_fork);
    e->d_levels.release(); e->d_cells.release(); e->d_tiles.release(); e->d_strips.release();
    for (auto& b : e->d_qsx0) b.release();
    for (auto& b : e->d_yofs) b.release();
    for (auto& b : e->d_qsel) b.release();
    for (auto& b : e->d_qalpha) b.release();
    for (auto& b : e->d_ibeta) b.release();
    for (auto& b : e->d_xofs) b.release();
    for (auto& b : e->d_ialpha) b.release();
    e->d_pyr.release(); e->d_blur.release(); e->d_cand.release(); e->d_scratch.release(); e->d_sel.release(); e->d_oct_nodes.release();
    e->d_cell_count.release(); e->d_sel_count.release(); e->d_kp_dst.release(); e->d_lvl_kps.release();
    e->d_kps.release(); e->d_desc.release(); e->d_n.release(); e->d_mono.release(); e->d_status.release();
    e->d_sad.release(); e->d_stereo_io.release();
    delete e;
}

int orbx_max_keypoints(const orbx_extractor* e) { return e ? e->max_kp : 0; }

// the same bound for ONE image geometry: per level max(N + 3, 32, 4 x nIni) with the level's own nIni (the bordered area of the upper
// levels of a wide image is relatively wider still: the 2 x 16-px border does not shrink with the level)
int orbx_max_keypoints_for(const orbx_extractor* e, int width, int height)
{
    if (!e || width < 1 || height < 1) return 0;
    int total = 0;
    for (int l = 0; l < e->nlevels; l++) {
        const int lw = (int)std::nearbyintf((float)width * e->inv_scale[l]), lh = (int)std::nearbyintf((float)height * e->inv_scale[l]);
        const int bw = (lw - kEdge + 3) - (kEdge - 3), bh = (lh - kEdge + 3) - (kEdge - 3);
        const int n_ini = bh > 0 ? (int)std::roundf((float)bw / (float)bh) : 0;
        total += std::max({e->nfeat[l] + 3, 32, 4 * std::max(n_ini, 0)});
    }
    return total;
}
int orbx_levels(const orbx_extractor* e) { return e ? e->nlevels : 0; }
float orbx_scale_factor(const orbx_extractor* e) { return e ? e->scale_factor_f : 0.f; }

int orbx_scale_tables(const orbx_extractor* e, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2)
{
    if (!e) return fail(ORBX_ERR_ARG, "NULL handle");
    for (int i = 0; i < e->nlevels; i++) {
        if (scale) scale[i] = e->scale[i];
        if (inv_scale) inv_scale[i] = e->inv_scale[i];
        if (sigma2) sigma2[i] = e->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = e->inv_sigma2[i];
    }
    return ORBX_OK;
}

int orbx_features_per_level(const orbx_extractor* e, int* n_per_level)
{
    if (!e || !n_per_level) return fail(ORBX_ERR_ARG, "NULL argument");
    for (int i = 0; i < e->nlevels; i++) n_per_level[i] = e->nfeat[i];
    return ORBX_OK;
}

int orbx_extract_batch_device(orbx_extractor* e, const uint8_t* d_imgs, int batch, int width, int height,
                              int row_stride, size_t frame_stride, int lap0, int lap1,
                              OrbxKeyPoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_n, int32_t* d_mono,
                              int32_t* d_status, void* stream)
{
    if (!e || !d_imgs || !d_kps || !d_desc || !d_n || !d_mono || !d_status) return fail(ORBX_ERR_ARG, "NULL argument");
    if (batch < 1 || width < 1 || height < 1 || row_stride < width || cap < 1) return fail(ORBX_ERR_ARG, "bad batch/geometry");
    ORBX_HIP(hipSetDevice(e->device));
    int r;
    if ((r = e->setup_geometry(width, height)) || (r = e->ensure_batch(batch))) return r;
    return e->enqueue(d_imgs, false, batch, row_stride, frame_stride, lap0, lap1, d_kps, d_desc, cap, d_n, d_mono, d_status,
                      (hipStream_t)stream);
}

int orbx_extract_batch(orbx_extractor* e, const uint8_t* const* imgs, int batch, int width, int height, int stride,
                       int lap0, int lap1, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n, int* mono_index)
{
    if (!e || !imgs || !kps || !desc || !n || !mono_index) return fail(ORBX_ERR_ARG, "NULL argument");
    if (batch < 1 || cap < 1) return fail(ORBX_ERR_ARG, "bad batch/capacity");
    if (width < 1 || height < 1) { for (int b = 0; b < batch; b++) { n[b] = 0; mono_index[b] = -1; } return ORBX_ERR_EMPTY; }
    if (stride < width) return fail(ORBX_ERR_ARG, "stride < width");
    ORBX_HIP(hipSetDevice(e->device));
    int r;
    if ((r = e->setup_geometry(width, height)) || (r = e->ensure_batch(batch))) return r;
    const size_t B = (size_t)batch;
    if ((r = e->d_kps.ensure(B * cap)) || (r = e->d_desc.ensure(B * cap * 32))) return r;
    const LevelDesc& L0 = e->levels[0];
    for (int b = 0; b < batch; b++) {
        if (!imgs[b]) return fail(ORBX_ERR_ARG, "imgs[%d] is NULL", b);
        ORBX_HIP(hipMemcpy2DAsync(e->d_pyr.p + b * e->pyr_frame_bytes + L0.off, L0.stride, imgs[b], stride, width, height,
                                  hipMemcpyHostToDevice, e->stream));
    }
    if ((r = e->enqueue(nullptr, true, batch, 0, 0, lap0, lap1, e->d_kps.p, e->d_desc.p, cap, e->d_n.p, e->d_mono.p, e->d_status.p, e->stream)))
        return r;
    std::vector<int> st(batch);
    ORBX_HIP(hipMemcpyAsync(n, e->d_n.p, sizeof(int) * B, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(mono_index, e->d_mono.p, sizeof(int) * B, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(st.data(), e->d_status.p, sizeof(int) * B, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(kps, e->d_kps.p, sizeof(OrbxKeyPoint) * B * cap, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipMemcpyAsync(desc, e->d_desc.p, B * cap * 32, hipMemcpyDeviceToHost, e->stream));
    ORBX_HIP(hipStreamSynchronize(e->stream));
    for (int b = 0; b < batch; b++)
        if (st[b] != ORBX_OK) return fail(st[b], "frame %d: device status %d (n=%d, cap=%d)", b, st[b], n[b], cap);
    return ORBX_OK;
}

int orbx_extract(orbx_extractor* e, const uint8_t* img, int width, int height, int stride,
                 int lap0, int lap1, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n, int* mono_index)
{
    if (!img || width < 1 || height < 1) {      // _image.empty() (:1090-1091)
        if (n) *n = 0;
        if (mono_index) *mono_index = -1;
        return ORBX_ERR_EMPTY;
    }
    const uint8_t* imgs[1] = {img};
    return orbx_extract_batch(e, imgs, 1, width, height, stride, lap0, lap1, kps, desc, cap, n, mono_index);
}

int orbx_pyramid_level_size(const orbx_extractor* e, int level, int* width, int* height)
{
    if (!e || level < 0 || level >= e->nlevels || e->geo_w == 0) return fail(ORBX_ERR_ARG, "no pyramid (level %d)", level);
    if (width) *width = e->levels[level].w;
    if (height) *height = e->levels[level].h;
    return ORBX_OK;
}

static int download_level(orbx_extractor* e, const uint8_t* base, int frame, int level, int border, uint8_t* dst, int dst_stride)
{
    if (!e || !dst || level < 0 || level >= e->nlevels || frame < 0 || frame >= e->last_batch || border < 0)
        return fail(ORBX_ERR_ARG, "bad frame/level");
    const LevelDesc& L = e->levels[level];
    if (dst_stride < L.w + 2 * border) return fail(ORBX_ERR_ARG, "dst_stride too small");
    ORBX_HIP(hipSetDevice(e->device));
    ORBX_HIP(hipStreamSynchronize(e->stream));
    ORBX_HIP(hipMemcpy2D(dst + (size_t)border * dst_stride + border, dst_stride, base + (size_t)frame * e->pyr_frame_bytes + L.off,
                         L.stride, L.w, L.h, hipMemcpyDeviceToHost));
    if (border > 0) {       // copyMakeBorder(..., BORDER_REFLECT_101) (:1185,:1190)
        auto refl = [](int p, int len) { if (len == 1) return 0; while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p; return p; };
        for (int y = 0; y < L.h + 2 * border; y++) {
            const int sy = refl(y - border, L.h) + border;
            uint8_t* row = dst + (size_t)y * dst_stride;
            const uint8_t* srow = dst + (size_t)sy * dst_stride;
            if (sy != y) std::memcpy(row + border, srow + border, L.w);
            for (int x = 0; x < border; x++) {
                row[x] = row[border + refl(x - border, L.w)];
                row[border + L.w + x] = row[border + refl(L.w + x, L.w)];
            }
        }
    }
    return ORBX_OK;
}

int orbx_pyramid_level(orbx_extractor* e, int frame, int level, int border, uint8_t* dst, int dst_stride)
{
    return download_level(e, e ? e->d_pyr.p : nullptr, frame, level, border, dst, dst_stride);
}

int orbx_debug_blurred_level(orbx_extractor* e, int frame, int level, uint8_t* dst, int dst_stride)
{
    return download_level(e, e ? e->d_blur.p : nullptr, frame, level, 0, dst, dst_stride);
}

int orbx_debug_candidates(orbx_extractor* e, int frame, int level, OrbxKeyPoint* out, int cap, int* n)
{
    if (!e || !out || !n || level < 0 || level >= e->nlevels || frame < 0 || frame >= e->last_batch) return fail(ORBX_ERR_ARG, "bad frame/level");
    ORBX_HIP(hipSetDevice(e->device));
    ORBX_HIP(hipStreamSynchronize(e->stream));
    const LevelDesc& L = e->levels[level];
    std::vector<int> cnt(std::max(L.cell_count, 1));
    std::vector<uint32_t> ent(std::max(L.cand_cap, 1));
    if (L.cell_count > 0) {
        ORBX_HIP(hipMemcpy(cnt.data(), e->d_cell_count.p + (size_t)frame * e->cells.size() + L.cell_begin, sizeof(int) * L.cell_count, hipMemcpyDeviceToHost));
        ORBX_HIP(hipMemcpy(ent.data(), e->d_cand.p + (size_t)frame * e->cand_frame_entries + L.cand_off, sizeof(uint32_t) * L.cand_cap, hipMemcpyDeviceToHost));
    }
    int k = 0;
    for (int c = 0; c < L.cell_count; c++) {
        const CellDesc& cd = e->cells[L.cell_begin + c];
        for (int i = 0; i < cnt[c]; i++, k++) {
            if (k >= cap) continue;
            const uint32_t en = ent[cd.slot_off - L.cand_off + i];
            OrbxKeyPoint kp;
            kp.x = (float)key_x(en); kp.y = (float)key_y(en); kp.size = 7.f; kp.angle = -1.f; kp.response = (float)key_resp(en);
            kp.octave = 0; kp.class_id = -1;
            out[k] = kp;
        }
    }
    *n = k;
    return ORBX_OK;
}

int orbx_debug_level_keypoints(orbx_extractor* e, int frame, int level, OrbxKeyPoint* out, int cap, int* n)
{
    if (!e || !out || !n || level < 0 || level >= e->nlevels || frame < 0 || frame >= e->last_batch) return fail(ORBX_ERR_ARG, "bad frame/level");
    ORBX_HIP(hipSetDevice(e->device));
    ORBX_HIP(hipStreamSynchronize(e->stream));
    const LevelDesc& L = e->levels[level];
    int cnt = 0;
    ORBX_HIP(hipMemcpy(&cnt, e->d_sel_count.p + (size_t)frame * e->nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    *n = cnt;
    const int m = std::min(cnt, cap);
    if (m > 0)
        ORBX_HIP(hipMemcpy(out, e->d_lvl_kps.p + (size_t)frame * e->sel_frame_entries + L.sel_off, sizeof(OrbxKeyPoint) * m, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

// Test hook: shrinks the per-wave corner list of the FAST kernel so that ordinary images take its overflow path (every pixel of
// the wave's rows goes through nms / emission).  cap <= 0 restores the default.  Takes effect at the next geometry setup.
int orbx_debug_set_tail_delay(orbx_extractor* e, int microseconds)
{
    if (!e || microseconds < 0 || microseconds > 20000) return fail(ORBX_ERR_ARG, "bad tail delay");
    e->dbg_tail_delay_us = microseconds;
    return ORBX_OK;
}

int orbx_debug_last_schedule(orbx_extractor* e)
{
    if (!e) return fail(ORBX_ERR_ARG, "NULL handle");
    return e->last_octree_variant;
}

int orbx_debug_set_fast_corner_cap(orbx_extractor* e, int cap)
{
    if (!e) return fail(ORBX_ERR_ARG, "NULL handle");
    e->fast_corner_cap = cap > 0 ? std::min(cap, 4096) : 0;
    e->geo_w = e->geo_h = 0;            // force setup_geometry to rebuild the LDS layout
    return ORBX_OK;
}

// host-side run of the device introsort restatement (pins it against std::sort in the CPU tests)
int orbx_debug_introsort(int32_t* count, int32_t* ulx, int32_t* node, int n)
{
    if (n < 0) return ORBX_ERR_ARG;
    std::vector<SortNode> v(n);
    for (int i = 0; i < n; i++) { v[i].count = count[i]; v[i].ulx = ulx[i]; v[i].node = node[i]; }
    std::vector<int> stack(3 * kIntrosortStack);
    introsort_nodes(v.data(), n, stack.data());
    for (int i = 0; i < n; i++) { count[i] = v[i].count; ulx[i] = v[i].ulx; node[i] = v[i].node; }
    return ORBX_OK;
}

// the octree's wave-parallel std::sort on the device (one wave, arrays staged in LDS exactly as k_octree holds them): pinned
// against std::sort itself by tests/test_extractor_gpu.py
int orbx_debug_wave_sort(int32_t* count, int32_t* ulx, int32_t* node, int n)
{
    if (n < 0 || n > 4000) return fail(ORBX_ERR_ARG, "n %d outside 0..4000", n);
    if (n == 0) return ORBX_OK;
    std::vector<SortNode> v(n);
    for (int i = 0; i < n; i++) { v[i].count = count[i]; v[i].ulx = ulx[i]; v[i].node = node[i]; }
    SortNode* d = nullptr;
    ORBX_HIP(hipMalloc(&d, sizeof(SortNode) * n));
    int r = ORBX_OK;
    const size_t lds = 2 * sizeof(SortNode) * (size_t)n + 2 * (size_t)n + 16;
    if (hipMemcpy(d, v.data(), sizeof(SortNode) * n, hipMemcpyHostToDevice) != hipSuccess) r = ORBX_ERR_HIP;
    if (!r && hipFuncSetAttribute((const void*)k_debug_wave_sort, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) r = ORBX_ERR_HIP;
    if (!r) {
        hipLaunchKernelGGL(k_debug_wave_sort, dim3(1), dim3(64), lds, 0, d, n);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) r = ORBX_ERR_HIP;
    }
    if (!r && hipMemcpy(v.data(), d, sizeof(SortNode) * n, hipMemcpyDeviceToHost) != hipSuccess) r = ORBX_ERR_HIP;
    (void)hipFree(d);
    if (r) return fail(r, "device sort failed");
    for (int i = 0; i < n; i++) { count[i] = v[i].count; ulx[i] = v[i].ulx; node[i] = v[i].node; }
    return ORBX_OK;
}

// Frame::ComputeStereoMatches (src/Frame.cc:931-1101) on the pyramids both extractors hold from their LAST extract call.
static int stereo_enqueue(orbx_extractor* l, orbx_extractor* r, int frame0, int batch,
                          const OrbxKeyPoint* d_kps_l, const uint8_t* d_desc_l, const int32_t* d_n_l,
                          const OrbxKeyPoint* d_kps_r, const uint8_t* d_desc_r, const int32_t* d_n_r, int cap,
                          float mb, float mbf, float* d_u_right, float* d_depth, hipStream_t st)
{
    if (l->geo_w == 0 || l->geo_w != r->geo_w || l->geo_h != r->geo_h || l->nlevels != r->nlevels || l->scale_factor_f != r->scale_factor_f)
        return fail(ORBX_ERR_ARG, "the two extractors do not hold pyramids of the same geometry");
    if (l->device != r->device) return fail(ORBX_ERR_ARG, "the two extractors live on different devices");
    if (frame0 < 0 || batch < 1 || frame0 + batch > l->last_batch || frame0 + batch > r->last_batch) return fail(ORBX_ERR_ARG, "frames outside the last extract call");
    if (l->nlevels > 16) return fail(ORBX_ERR_ARG, "more than 16 pyramid levels");
    if (cap < 1 || cap > 8192) return fail(ORBX_ERR_ARG, "cap %d outside 1..8192", cap);
    if (!(mb > 0.f) || !(mbf > 0.f)) return fail(ORBX_ERR_ARG, "mb and mbf must be positive");
    int rr = l->d_sad.ensure((size_t)batch * cap);
    if (rr) return rr;
    StereoTables T;
    for (int i = 0; i < 16; i++) { T.scale[i] = l->scale[std::min(i, l->nlevels - 1)]; T.inv_scale[i] = l->inv_scale[std::min(i, l->nlevels - 1)]; }
    const uint8_t* pl = l->d_pyr.p + (size_t)frame0 * l->pyr_frame_bytes;
    const uint8_t* pr = r->d_pyr.p + (size_t)frame0 * r->pyr_frame_bytes;
    hipLaunchKernelGGL(k_stereo_match, dim3((cap + 3) / 4, batch), dim3(256), 0, st, pl, pr, l->pyr_frame_bytes, l->d_levels.p, T,
                       d_kps_l, d_desc_l, d_n_l, d_kps_r, d_desc_r, d_n_r, cap, mb, mbf, l->nlevels, d_u_right, d_depth, l->d_sad.p);
    int n_pow2 = 2;
    while (n_pow2 < cap) n_pow2 <<= 1;
    hipLaunchKernelGGL(k_stereo_median, dim3(batch), dim3(256), (size_t)n_pow2 * sizeof(int), st, d_n_l, cap, n_pow2, d_u_right, d_depth, l->d_sad.p);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

int orbx_stereo_matches_device(orbx_extractor* left, orbx_extractor* right, int batch,
                               const OrbxKeyPoint* d_kps_l, const uint8_t* d_desc_l, const int32_t* d_n_l,
                               const OrbxKeyPoint* d_kps_r, const uint8_t* d_desc_r, const int32_t* d_n_r, int cap,
                               float mb, float mbf, float* d_u_right, float* d_depth, void* stream)
{
    if (!left || !right || !d_kps_l || !d_desc_l || !d_n_l || !d_kps_r || !d_desc_r || !d_n_r || !d_u_right || !d_depth) return fail(ORBX_ERR_ARG, "NULL argument");
    ORBX_HIP(hipSetDevice(left->device));
    return stereo_enqueue(left, right, 0, batch, d_kps_l, d_desc_l, d_n_l, d_kps_r, d_desc_r, d_n_r, cap, mb, mbf, d_u_right, d_depth, (hipStream_t)stream);
}

int orbx_stereo_matches(orbx_extractor* left, orbx_extractor* right, int frame,
                        const OrbxKeyPoint* kps_l, const uint8_t* desc_l, int n_l,
                        const OrbxKeyPoint* kps_r, const uint8_t* desc_r, int n_r, float mb, float mbf, float* u_right, float* depth)
{
    if (!left || !right || n_l < 0 || n_r < 0 || (n_l > 0 && (!kps_l || !desc_l || !u_right || !depth)) || (n_r > 0 && (!kps_r || !desc_r)))
        return fail(ORBX_ERR_ARG, "NULL argument");
    if (n_l == 0) return ORBX_OK;
    ORBX_HIP(hipSetDevice(left->device));
    const int cap = std::max(std::max(n_l, n_r), 1);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_kl = 0, o_dl = al(o_kl + sizeof(OrbxKeyPoint) * (size_t)cap), o_kr = al(o_dl + 32 * (size_t)cap), o_dr = al(o_kr + sizeof(OrbxKeyPoint) * (size_t)cap),
                 o_n = al(o_dr + 32 * (size_t)cap), o_u = al(o_n + 8), o_d = al(o_u + 4 * (size_t)cap), total = al(o_d + 4 * (size_t)cap);
    int r = left->d_stereo_io.ensure(total);
    if (r) return r;
    uint8_t* b = left->d_stereo_io.p;
    hipStream_t st = left->stream;
    const int32_t nn[2] = {n_l, n_r};
    ORBX_HIP(hipMemcpyAsync(b + o_kl, kps_l, sizeof(OrbxKeyPoint) * (size_t)n_l, hipMemcpyHostToDevice, st));
    ORBX_HIP(hipMemcpyAsync(b + o_dl, desc_l, 32 * (size_t)n_l, hipMemcpyHostToDevice, st));
    if (n_r > 0) {
        ORBX_HIP(hipMemcpyAsync(b + o_kr, kps_r, sizeof(OrbxKeyPoint) * (size_t)n_r, hipMemcpyHostToDevice, st));
        ORBX_HIP(hipMemcpyAsync(b + o_dr, desc_r, 32 * (size_t)n_r, hipMemcpyHostToDevice, st));
    }
    ORBX_HIP(hipMemcpyAsync(b + o_n, nn, 8, hipMemcpyHostToDevice, st));
    ORBX_HIP(hipStreamSynchronize(right->stream));      // the right pyramid was written on the other handle's stream
    r = stereo_enqueue(left, right, frame, 1, (const OrbxKeyPoint*)(b + o_kl), b + o_dl, (const int32_t*)(b + o_n),
                       (const OrbxKeyPoint*)(b + o_kr), b + o_dr, (const int32_t*)(b + o_n) + 1, cap, mb, mbf, (float*)(b + o_u), (float*)(b + o_d), st);
    if (r) return r;
    ORBX_HIP(hipMemcpyAsync(u_right, b + o_u, 4 * (size_t)n_l, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipMemcpyAsync(depth, b + o_d, 4 * (size_t)n_l, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipStreamSynchronize(st));
    return ORBX_OK;
}

// host-side evaluation of the shared float helpers (pins them against the oracle in the CPU tests)
float orbx_debug_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }
void orbx_debug_sincos(float a, float* c, float* s) { sincos_f32(a, c, s); }

#ifdef ORBX_FAST_TIMING
int orbx_debug_fast_prof(unsigned long long* out16)
{
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(d_fast_prof), sizeof(z)) != hipSuccess) return ORBX_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(d_fast_prof), z, sizeof(z)) != hipSuccess) return ORBX_ERR_HIP;
    return ORBX_OK;
}
#endif
#ifdef ORBX_OCT_TIMING
int orbx_debug_oct_prof(unsigned long long* out10)
{
    if (hipMemcpyFromSymbol(out10, HIP_SYMBOL(d_oct_prof), 10 * sizeof(unsigned long long)) != hipSuccess) return ORBX_ERR_HIP;
    return ORBX_OK;
}
#endif


}  // extern "C"
