// orbx_kernels.hip -- gfx950 kernels of the ORB extractor (reference src/ORBextractor.cc).
//
// Pipeline per batch of B frames (all launches cover the whole batch; blockIdx.y / .z = frame):
//   k_resize        level l-1 -> l, fixed-point bilinear; k_resize_tail: the upper levels in one launch (E1, ComputePyramid :1170-1195)
//   k_fast_strips   FAST-9/16 score + per-cell NMS / threshold fallback / ordered compaction, a workgroup per strip of cells
//                                                                    (E2, ComputeKeyPointsOctTree :787-872)
//   k_octree        quadtree keypoint selection, one wave per (frame, level), a lane per node within a pass
//                                                                    (E3, DistributeOctTree :555-779)
//   k_index         output slot of every keypoint (lapping-area split) (E8, operator() :1140-1167)
//   k_blur          7x7 sigma-2 Gaussian, Q8.8 fixed point             (E6, :1132-1133)
//   k_orient_desc   intensity-centroid angle + steered BRIEF-256, half a wave per keypoint (E5 :76-103, E7 :107-146)
//
// Integer stages are exact by construction; float stages use orbx_math.h (no FMA, no libm).
// HBM layout: one pyramid buffer per frame, levels back to back, row stride rounded up to 64 B so that
// every row starts on a 64-B boundary (coalesced dword loads); blurred pyramid has the same layout.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_device.h"
#include "orbx_introsort.h"
#include "orbx_math.h"

namespace orbx {

__device__ __align__(16) const signed char d_pattern[1024] = {
#include "orb_pattern_31.inc"
};

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(uint32_t v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ uint32_t as_u32(us2 v) { return __builtin_bit_cast(uint32_t, v); }

// ------------------------------------------------------------------------------------------------
// E1: bilinear resize, OpenCV fixed-point scheme (cv::resize INTER_LINEAR 8UC1).  One thread -> 4 horizontally
// adjacent output pixels (one aligned dword store); a wave = 256 pixels of one output row, a workgroup = 4 rows.
// Everything that depends on the output column only is a host-built table per QUAD of columns: the first source column
// sx0, and per pixel a v_perm selector that pulls (S[sx], S[sx + 1]) out of the 8 source bytes starting at sx0 as a u16 pair,
// and the coefficient pair (ialpha0, ialpha1) -- so the horizontal pass of a pixel and source row is v_perm + v_dot2.
// The source rows are read straight from L2 / HBM (three dwords per row and thread, no LDS staging, no barrier).
// The source is addressed on its own (base, frame stride, row stride): level 1 is made from the caller's image in place.
// ------------------------------------------------------------------------------------------------
struct SrcImage {
    const uint8_t* base;        // first pixel of frame 0
    size_t frame_stride;        // bytes between frames
    int32_t stride, w, h;       // row stride in bytes (multiple of 4), size
};

struct __attribute__((aligned(4))) Dwords3 { uint32_t x, y, z; };     // 12 bytes at dword alignment: one global_load_dwordx3
struct __attribute__((aligned(4))) Dwords4 { uint32_t x, y, z, w; };  // 16 bytes at dword alignment: one global_load_dwordx4
constexpr int kResizeRows = 64, kResizeTW = 64;      // output tile of a workgroup: 64 x 64 (16 rows per wave, 64-pixel segments of four rows per wave step)

// one tile of 64 x 64 outputs of level `dst` of frame `frame` by 4 waves (`wave` 0..3, 16 rows each).  A wave covers 16 quads
// (64 pixels) of FOUR rows at a time: levels are 179..533 pixels wide, and whole waves per row (256 pixels) left a third of the
// lanes idle on average (533 px = 2.08 waves); with 64-pixel segments the idle share is a few percent.
__device__ __forceinline__ void resize_tile(const SrcImage& src, uint8_t* __restrict__ pyr, size_t frame_stride, const LevelDesc& dst,
                                            const int* __restrict__ q_sx0, const uint4* __restrict__ q_sel, const uint4* __restrict__ q_alpha,
                                            const int* __restrict__ yofs, const short* __restrict__ ibeta, int tile, int frame, int lane, int wave)
{
    const int tiles_x = (dst.w + kResizeTW - 1) / kResizeTW;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int q = tx * (kResizeTW / 4) + (lane & 15), rsel = lane >> 4;
    if (4 * q >= dst.w) return;
    const int sx0 = q_sx0[q];
    const uint4 sel = q_sel[q], al = q_alpha[q];
    const uint32_t sels[4] = {sel.x, sel.y, sel.z, sel.w}, als[4] = {al.x, al.y, al.z, al.w};
    const int a = sx0 & ~3;
    const uint32_t sh = (uint32_t)(sx0 & 3);
    const int last = src.stride - 4;                            // last dword of a row: bytes past the row's pixels only meet coefficient 0
    const int o0 = a, o1 = min(a + 4, last), o2 = min(a + 8, last);
    const uint8_t* S = src.base + (size_t)frame * src.frame_stride;
    uint8_t* D = pyr + (size_t)frame * frame_stride + dst.off + 4 * q;
    constexpr int NR = kResizeRows / 16;                        // row groups of 4 per wave
    const int dy0 = ty * kResizeRows + wave * (kResizeRows / 4) + rsel;
    // all the loads of the wave's rows are issued before the first result is stored (the compiler cannot move a load above a
    // store through pointers it cannot tell apart)
    uint32_t u[NR][3], v[NR][3], bb[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int dy = min(dy0 + 4 * r, dst.h - 1);
        const int sy = yofs[dy];
        const int sy0 = min(max(sy, 0), src.h - 1), sy1 = min(max(sy + 1, 0), src.h - 1);
        bb[r] = ((const uint32_t*)ibeta)[dy];                      // (ibeta0, ibeta1) as one dword
        const uint8_t* r0p = S + (size_t)sy0 * src.stride;
        const uint8_t* r1p = S + (size_t)sy1 * src.stride;
        if (a + 8 <= last) {        // one 12-byte load per row: the address unit handles 4 lanes per clock whatever the width
            const Dwords3 tu = *(const Dwords3*)(r0p + a), tv = *(const Dwords3*)(r1p + a);
            u[r][0] = tu.x; u[r][1] = tu.y; u[r][2] = tu.z;
            v[r][0] = tv.x; v[r][1] = tv.y; v[r][2] = tv.z;
        } else {                    // the last quads of a row: never read past the row's last dword
            u[r][0] = *(const uint32_t*)(r0p + o0); u[r][1] = *(const uint32_t*)(r0p + o1); u[r][2] = *(const uint32_t*)(r0p + o2);
            v[r][0] = *(const uint32_t*)(r1p + o0); v[r][1] = *(const uint32_t*)(r1p + o1); v[r][2] = *(const uint32_t*)(r1p + o2);
        }
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int dy = dy0 + 4 * r;
        if (dy >= dst.h) continue;
        const uint32_t b0 = bb[r] & 0xFFFFu, b1 = bb[r] >> 16;
        // the 8 source bytes from column sx0 on, per row
        const uint32_t ulo = __builtin_amdgcn_alignbyte(u[r][1], u[r][0], sh), uhi = __builtin_amdgcn_alignbyte(u[r][2], u[r][1], sh);
        const uint32_t vlo = __builtin_amdgcn_alignbyte(v[r][1], v[r][0], sh), vhi = __builtin_amdgcn_alignbyte(v[r][2], v[r][1], sh);
        uint32_t out[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const us2 ak = as_us2(als[k]);
            const uint32_t h0 = __builtin_amdgcn_udot2(as_us2(__builtin_amdgcn_perm(uhi, ulo, sels[k])), ak, 0u, false);    // S0[sx] a0 + S0[sx+1] a1
            const uint32_t h1 = __builtin_amdgcn_udot2(as_us2(__builtin_amdgcn_perm(vhi, vlo, sels[k])), ak, 0u, false);
            const uint32_t val = ((__umul24(b0, h0 >> 4) >> 16) + (__umul24(b1, h1 >> 4) >> 16) + 2u) >> 2;
            out[k] = min(val, 255u);
        }
        // columns past dst.w: coefficients 0 -> 0; the row padding absorbs the tail of the last dword
        *(uint32_t*)(D + (size_t)dy * dst.stride) = out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24);
    }
}

__global__ __launch_bounds__(256) void k_resize(SrcImage src, uint8_t* __restrict__ pyr, size_t frame_stride, LevelDesc dst,
                                                const int* __restrict__ q_sx0, const uint4* __restrict__ q_sel, const uint4* __restrict__ q_alpha,
                                                const int* __restrict__ yofs, const short* __restrict__ ibeta)
{
    // tiles of 64 x 64 outputs are numbered row-major and handed to the XCDs in contiguous bands (xcd_remap), so the source
    // rows a band of destination rows needs are fetched by one L2 only
    const int tiles_x = (dst.w + kResizeTW - 1) / kResizeTW, tiles_y = (dst.h + kResizeRows - 1) / kResizeRows;
    const int tile = xcd_remap(blockIdx.x, gridDim.x, blockIdx.y);
    if (tile >= tiles_x * tiles_y) return;
    resize_tile(src, pyr, frame_stride, dst, q_sx0, q_sel, q_alpha, yofs, ibeta, tile, blockIdx.y, threadIdx.x & 63,
                __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

// Fallback for levels whose four-column source span exceeds the 8-byte window of the table-driven kernel above (scale factors of
// 2 and more: 3 x scale_x >= 6 source columns between the first and the last pixel of a quad, plus its right neighbour): one thread
// per output pixel straight from the per-column / per-row tables of cv::resize -- the same fixed-point arithmetic, no window.
__global__ __launch_bounds__(256) void k_resize_generic(SrcImage src, uint8_t* __restrict__ pyr, size_t frame_stride, LevelDesc dst,
                                                        const int* __restrict__ xofs, const short* __restrict__ ialpha,
                                                        const int* __restrict__ yofs, const short* __restrict__ ibeta)
{
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y, frame = blockIdx.z;
    if (dx >= dst.w) return;
    const uint8_t* S = src.base + (size_t)frame * src.frame_stride;
    const int sx = xofs[dx], sy = yofs[dy];
    const uint32_t a0 = (uint16_t)ialpha[2 * dx], a1 = (uint16_t)ialpha[2 * dx + 1];
    const uint32_t b0 = (uint16_t)ibeta[2 * dy], b1 = (uint16_t)ibeta[2 * dy + 1];
    const int sy0 = min(max(sy, 0), src.h - 1), sy1 = min(max(sy + 1, 0), src.h - 1);
    const int sx1 = min(sx + 1, src.w - 1);                     // (its coefficient is 0 where sx is the last column)
    const uint8_t* r0 = S + (size_t)sy0 * src.stride;
    const uint8_t* r1 = S + (size_t)sy1 * src.stride;
    const uint32_t h0 = r0[sx] * a0 + r0[sx1] * a1, h1 = r1[sx] * a0 + r1[sx1] * a1;
    const uint32_t val = ((__umul24(b0, h0 >> 4) >> 16) + (__umul24(b1, h1 >> 4) >> 16) + 2u) >> 2;
    pyr[(size_t)frame * frame_stride + dst.off + (size_t)dy * dst.stride + dx] = (uint8_t)min(val, 255u);
}

// The upper pyramid levels are small and each depends on the one below: as launches of their own they are a chain of short,
// launch-latency-bound kernels.  Here ONE 1024-thread workgroup per frame walks the levels l_first .. n_levels-1 in turn (four
// tiles at a time, a workgroup barrier between levels: the level just written is read back by the same CU).
struct ResizeTables {
    const int* q_sx0[kMaxLevels]; const uint4* q_sel[kMaxLevels]; const uint4* q_alpha[kMaxLevels];
    const int* yofs[kMaxLevels]; const short* ibeta[kMaxLevels];
};
__global__ __launch_bounds__(1024) void k_resize_tail(uint8_t* __restrict__ pyr, size_t frame_stride, const LevelDesc* __restrict__ levels,
                                                      ResizeTables T, int l_first, int n_levels)
{
    const int frame = blockIdx.x;
    const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
    const int lane = threadIdx.x & 63;
    for (int l = l_first; l < n_levels; l++) {
        const LevelDesc D = levels[l], P = levels[l - 1];
        SrcImage src;
        src.base = pyr + P.off; src.frame_stride = frame_stride; src.stride = P.stride; src.w = P.w; src.h = P.h;
        const int tiles = ((D.w + kResizeTW - 1) / kResizeTW) * ((D.h + kResizeRows - 1) / kResizeRows);
        for (int t = sub; t < tiles; t += 4)
            resize_tile(src, pyr, frame_stride, D, T.q_sx0[l], T.q_sel[l], T.q_alpha[l], T.yofs[l], T.ibeta[l], t, frame, lane, wave);
        __syncthreads();            // the level is complete (and visible to this CU) before it becomes the next one's source
    }
}

// ------------------------------------------------------------------------------------------------
// E2: FAST-9/16 on strips of cells -- orbx_fast_strips.inc
// ------------------------------------------------------------------------------------------------
#include "orbx_fast_strips.inc"

// ------------------------------------------------------------------------------------------------
// E3: DistributeOctTree.  One wave per (frame, level).  The std::list of nodes is an index-linked list
// in LDS; a node's keys are a contiguous, order-preserving segment of a ping-pong key array (LDS when
// the level's candidates fit, HBM scratch otherwise).  A pass of the reference's loops is a batch of up to 64
// DivideNode calls, a lane per node (see k_octree); std::sort is wave-parallel (orbx_introsort.h).
// ------------------------------------------------------------------------------------------------
#ifdef ORBX_OCT_TIMING      // cycle split of the (level 0, frame 0) wave (tools/oct_timing.py); never defined in the product build
__device__ unsigned long long d_oct_prof[10];
#endif
struct OctLds {
    short* ulx; short* uly; short* brx; short* bry;
    int* beg; int* cnt;
    int* pidx; short* freelist; short* order;       // pidx indexes the push log (6 x pool entries: beyond 16 bits for large pools)
    short* plog;            // push log: the std::list order is the REVERSE of this array without its tombstones (-1)
    uint8_t* flg;           // bit0: bNoMore, bit1: keys live in buffer 1
};

// LDS_KEYS: both ping-pong key buffers live in LDS (the host sizes them for the largest level, so the choice is static and
// every key access is a ds_ instruction); otherwise they live in the L2-resident HBM scratch (global_ instructions).
// LDS_NODES: the node pool (boxes, key ranges, push log, the two sort arrays) lives in LDS, which holds about 2500 nodes; levels
// that ask for more features than that (e.g. 12000 features per frame) take the instantiation whose pool lives in an HBM /
// L2-resident scratch of its own -- the same code, global_ instead of ds_ instructions.
//
// One pass of the reference's loops is a BATCH of up to 64 independent DivideNode calls (the nodes of the list that can still
// be divided, resp. the next 64 entries of the sorted to-expand vector), one lane per node: the lanes count their node's keys
// per child (a node with many keys is counted and scattered by the whole wave instead), prefix sums over the lanes give every
// child its node slot, its place in the push log and in the next to-expand vector in exactly the order the sequential loop
// would have produced, and -- in the sorted phase -- the lane at which the running size reaches N (the reference breaks
// there: later lanes do not divide).
template <bool LDS_KEYS, bool LDS_NODES>
__device__ __forceinline__ void octree_body(const LevelDesc* __restrict__ levels, const CellDesc* __restrict__ cells,
                                            const uint32_t* __restrict__ cand, size_t cand_frame_stride,
                                            const int* __restrict__ cell_count, int n_cells,
                                            uint32_t* __restrict__ scratch, size_t scratch_frame_stride,
                                            int pool, int lds_keys_cap,
                                            uint32_t* __restrict__ sel, int sel_frame_stride,
                                            int* __restrict__ sel_count, int n_levels, int* __restrict__ status,
                                            uint8_t* __restrict__ node_scratch, size_t node_stride, int level_first,
                                            uint8_t* __restrict__ smem, int* __restrict__ s_sort_stack)
{
    static_assert(LDS_NODES || !LDS_KEYS, "a pool too large for LDS leaves no room for LDS keys");
    const int level = level_first + blockIdx.x, frame = blockIdx.y;     // (a launch covers the levels level_first ..: see enqueue())
    const int lane = threadIdx.x;
    const LevelDesc L = levels[level];
    const int N = L.nfeat;
    const unsigned long long lt = (1ull << lane) - 1ull;
#ifdef ORBX_OCT_TIMING
    long long tq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool timed = level == 0 && frame == 0 && lane == 0;
    long long t_prev = clock64();
    const long long t_kernel0 = t_prev;
    int n_div = 0;
#define ORBX_OTICK(k) if (timed) { const long long t_now = clock64(); tq[k] += t_now - t_prev; t_prev = t_now; }
#else
#define ORBX_OTICK(k)
#endif

    // carve the node pool
    OctLds S;
    uint8_t* p;
    if constexpr (LDS_NODES) p = smem;
    else p = node_scratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * node_stride;
    SortNode* const ex_new = (SortNode*)p; p += sizeof(SortNode) * pool;    // the to-expand vector being appended to
    SortNode* const ex_sorted = (SortNode*)p; p += sizeof(SortNode) * pool; // the sorted previous one (and the sort's scratch)
    S.beg = (int*)p; p += 4 * pool;
    S.cnt = (int*)p; p += 4 * pool;
    S.ulx = (short*)p; p += 2 * pool;  S.uly = (short*)p; p += 2 * pool;
    S.brx = (short*)p; p += 2 * pool;  S.bry = (short*)p; p += 2 * pool;
    S.pidx = (int*)p; p += 4 * pool;
    S.freelist = (short*)p; p += 2 * pool; S.order = (short*)p; p += 2 * pool;
    S.plog = (short*)p; p += 2 * kOctLogFactor * pool;
    S.flg = p; p += (pool + 15) & ~15;
    // the two key buffers: kb[0 .. cap) and kb[cap .. 2 cap)
    uint32_t* kb;
    int cap;
    if constexpr (LDS_KEYS) { kb = (uint32_t*)p; cap = lds_keys_cap; }
    else { kb = scratch + (size_t)frame * scratch_frame_stride + 2 * (size_t)L.cand_off; cap = L.cand_cap; }

    int* out_count = sel_count + (size_t)frame * n_levels + level;
    uint32_t* out = sel + (size_t)frame * sel_frame_stride + L.sel_off;

    // ---- gather the level's candidates in vToDistributeKeys order (cell row-major, row-major inside a cell): one lane per
    // cell, the in-wave prefix of the cell counts gives each cell its place in the ordered list
    const uint32_t* fc = cand + (size_t)frame * cand_frame_stride;
    const int* cc = cell_count + (size_t)frame * n_cells;
    int total = 0;
    for (int c0 = 0; c0 < L.cell_count; c0 += 64) {
        const int ci = c0 + lane;
        int n = 0, slot_off = 0;
        if (ci < L.cell_count) { n = cc[L.cell_begin + ci]; slot_off = cells[L.cell_begin + ci].slot_off; }
        const int incl = wave_incl_scan(n);
        const int base = total + incl - n;
        if (base + n > cap) n = 0;          // cannot happen (the buffers hold a whole level); reported below
        for (int k = 0; __ballot(k < n) != 0ull; k += 8) {
            uint32_t e[8];
#pragma unroll
            for (int u = 0; u < 8; u++) e[u] = (k + u < n) ? fc[slot_off + k + u] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++) if (k + u < n) kb[base + k + u] = e[u];
        }
        total += __builtin_amdgcn_readlane(incl, 63);
    }
    if (total > cap) {
        if (lane == 0) { atomicExch(status + frame, ORBX_ERR_INTERNAL); *out_count = 0; }
        return;
    }
    __syncthreads();
    ORBX_OTICK(0)

    const int width = (L.w - kEdge + 3) - (kEdge - 3), height = (L.h - kEdge + 3) - (kEdge - 3);
    const int nIni = (height > 0) ? (int)roundf((float)width / (float)height) : 0;      // :559
    if (total == 0 || nIni <= 0 || nIni > pool / 2) {
        if (lane == 0) {
            *out_count = 0;
            if (total != 0 && nIni > pool / 2) atomicExch(status + frame, ORBX_ERR_INTERNAL);
        }
        return;
    }

    // ---- the list.  std::list<ExtractorNode> with push_front / erase only needs an append-only push log: iterating the
    // list from begin() is walking the log backwards, erase() leaves a tombstone, and nodes pushed while a pass is running
    // sit behind its start index exactly like nodes pushed in front of a running list iterator.
    const int plog_cap = kOctLogFactor * pool;
    int np = 0, size = 0, nfree = pool, n_ex = 0;
    for (int i = lane; i < pool; i += 64) S.freelist[i] = (short)(pool - 1 - i);
    __syncthreads();
    bool overflow = false;
    auto compact = [&]() {          // drop tombstones, order preserved (wave-parallel, in place)
        int w = 0;
        for (int i0 = 0; i0 < np; i0 += 64) {
            const int i = i0 + lane;
            const int id = (i < np) ? (int)S.plog[i] : -1;
            const unsigned long long m = __ballot(id >= 0);
            if (id >= 0) { const int pos = w + __popcll(m & lt); S.plog[pos] = (short)id; S.pidx[id] = pos; }
            w += __popcll(m);
        }
        np = w;
        __syncthreads();
    };

    // ---- root nodes (:564-586): nIni vertical strips of width hX
    const float hX = (float)width / (float)nIni;
    if (nIni == 1) {
        // one root that owns every key: they already stand in order in buffer 0 ((int)(x / hX) is 0 for every x < width)
        if (lane == 0) {
            const int id = S.freelist[nfree - 1];
            S.ulx[id] = 0; S.uly[id] = 0; S.brx[id] = (short)(int)hX; S.bry[id] = (short)height;
            S.beg[id] = 0; S.cnt[id] = total; S.flg[id] = (uint8_t)(total == 1 ? 1 : 0);
            S.plog[0] = (short)id; S.pidx[id] = 0;
        }
        nfree--; np = 1; size = 1;
        __syncthreads();
    } else {
        // stable partition of the keys by strip into buffer 1
        int strip_beg = 0;
        for (int s = 0; s < nIni; s++) {
            int cnt = 0;
            for (int k0 = 0; k0 < total; k0 += 64) {
                const int k = k0 + lane;
                bool in = false;
                uint32_t e = 0;
                if (k < total) { e = kb[k]; in = ((int)((float)key_x(e) / hX) == s); }
                const unsigned long long m = __ballot(in);
                if (in) kb[cap + strip_beg + cnt + __popcll(m & lt)] = e;
                cnt += __popcll(m);
            }
            int id = -1;
            if (cnt > 0) {      // empty roots are erased (:595-596)
                id = S.freelist[--nfree];
                if (lane == 0) {
                    S.ulx[id] = (short)(int)(hX * (float)s);       S.uly[id] = 0;
                    S.brx[id] = (short)(int)(hX * (float)(s + 1)); S.bry[id] = (short)height;
                    S.beg[id] = strip_beg; S.cnt[id] = cnt;
                    S.flg[id] = (uint8_t)(2 | (cnt == 1 ? 1 : 0));
                }
            }
            if (lane == 0) S.order[s] = (short)id;
            strip_beg += cnt;
        }
        __syncthreads();
        // the reference push_back()s the roots in strip order; in push-log terms the first strip is pushed last
        for (int s = nIni - 1; s >= 0; s--) {
            const int id = S.order[s];
            if (id >= 0) { if (lane == 0) { S.plog[np] = (short)id; S.pidx[id] = np; } np++; size++; }
        }
        __syncthreads();
    }
    ORBX_OTICK(1)

    // One batch of DivideNode calls (:480-536 and the loop bodies :610-676 / :706-750): lane l divides node `id` (or none: -1);
    // lanes are in processing order.  sorted_phase: stop after the lane whose divide brings the list to N nodes; returns
    // whether that happened.
    auto divide_batch = [&](const int id, const bool sorted_phase, int& n_to_expand) -> bool {
        const bool act = id >= 0;
        int ulx = 0, uly = 0, brx = 0, bry = 0, beg = 0, cnt = 0, src = 0;
        if (act) {
            ulx = S.ulx[id]; uly = S.uly[id]; brx = S.brx[id]; bry = S.bry[id];
            beg = S.beg[id]; cnt = S.cnt[id]; src = (S.flg[id] >> 1) & 1;
        }
        const int halfX = (brx - ulx + 1) >> 1;     // ceil(float(w)/2), w >= 0
        const int halfY = (bry - uly + 1) >> 1;
        const int midx = ulx + halfX, midy = uly + halfY;
        const int in_off = (src ? cap : 0) + beg, out_off = (src ? 0 : cap) + beg;
        // Scattering a node's keys into the other buffer is harmless when the node ends up not being divided (its range of
        // the other buffer is dead space of its own), so counting and scattering do not wait for the cut.
        const unsigned long long m_act = __ballot(act);
        const int small_max = (__popcll(m_act) <= 6) ? 0 : 32;       // a handful of nodes: the whole wave takes them one by one
        const bool small = act && cnt <= small_max;
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (__ballot(small) != 0ull) {
            // a lane per node: count, then stable scatter
            for (int k = 0; __ballot(small && k < cnt) != 0ull; k += 4) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (small && k + u < cnt) {
                        const uint32_t e = kb[in_off + k + u];
                        const int cls = ((int)key_x(e) >= midx ? 1 : 0) + ((int)key_y(e) >= midy ? 2 : 0);
                        c0 += cls == 0; c1 += cls == 1; c2 += cls == 2; c3 += cls == 3;
                    }
                }
            }
            int w0 = out_off, w1 = out_off + c0, w2 = w1 + c1, w3 = w2 + c2;
            for (int k = 0; __ballot(small && k < cnt) != 0ull; k += 4) {
                uint32_t e[4];
#pragma unroll
                for (int u = 0; u < 4; u++) e[u] = (small && k + u < cnt) ? kb[in_off + k + u] : 0u;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (small && k + u < cnt) {
                        const bool r = (int)key_x(e[u]) >= midx, d = (int)key_y(e[u]) >= midy;
                        const int pos = d ? (r ? w3 : w2) : (r ? w1 : w0);
                        kb[pos] = e[u];
                        w0 += (!d && !r); w1 += (!d && r); w2 += (d && !r); w3 += (d && r);
                    }
                }
            }
        }
        for (unsigned long long mb = __ballot(act && !small); mb != 0ull; mb &= mb - 1ull) {
            // the wave per node: ballot + prefix popcount
            const int b = __builtin_ctzll(mb);
            const int bin = __builtin_amdgcn_readlane(in_off, b), bout = __builtin_amdgcn_readlane(out_off, b);
            const int bcnt = __builtin_amdgcn_readlane(cnt, b);
            const int bmx = __builtin_amdgcn_readlane(midx, b), bmy = __builtin_amdgcn_readlane(midy, b);
            int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            if (bcnt <= 64) {
                int cls = -1;
                uint32_t e = 0;
                if (lane < bcnt) { e = kb[bin + lane]; cls = ((int)key_x(e) >= bmx ? 1 : 0) + ((int)key_y(e) >= bmy ? 2 : 0); }
                const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2), m3 = __ballot(cls == 3);
                s0 = __popcll(m0); s1 = __popcll(m1); s2 = __popcll(m2); s3 = __popcll(m3);
                const unsigned long long mm = (cls == 0) ? m0 : (cls == 1) ? m1 : (cls == 2) ? m2 : m3;
                const int basec = (cls == 0) ? 0 : (cls == 1) ? s0 : (cls == 2) ? s0 + s1 : s0 + s1 + s2;
                if (cls >= 0) kb[bout + basec + __popcll(mm & lt)] = e;
            } else {
                // pass 1: child sizes (four chunks of 64 keys in flight)
                for (int k0 = 0; k0 < bcnt; k0 += 256) {
                    uint32_t e[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int k = k0 + 64 * u + lane; e[u] = (k < bcnt) ? kb[bin + k] : 0u; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int k = k0 + 64 * u + lane;
                        const int cls = (k < bcnt) ? ((int)key_x(e[u]) >= bmx ? 1 : 0) + ((int)key_y(e[u]) >= bmy ? 2 : 0) : -1;
                        s0 += __popcll(__ballot(cls == 0)); s1 += __popcll(__ballot(cls == 1));
                        s2 += __popcll(__ballot(cls == 2)); s3 += __popcll(__ballot(cls == 3));
                    }
                }
                // pass 2: stable scatter into the other buffer
                int w0 = bout, w1 = bout + s0, w2 = w1 + s1, w3 = w2 + s2;
                for (int k0 = 0; k0 < bcnt; k0 += 256) {
                    uint32_t e[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int k = k0 + 64 * u + lane; e[u] = (k < bcnt) ? kb[bin + k] : 0u; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int k = k0 + 64 * u + lane;
                        const int cls = (k < bcnt) ? ((int)key_x(e[u]) >= bmx ? 1 : 0) + ((int)key_y(e[u]) >= bmy ? 2 : 0) : -1;
                        const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2), m3 = __ballot(cls == 3);
                        const unsigned long long mm = (cls == 0) ? m0 : (cls == 1) ? m1 : (cls == 2) ? m2 : m3;
                        const int wc = (cls == 0) ? w0 : (cls == 1) ? w1 : (cls == 2) ? w2 : w3;
                        if (cls >= 0) kb[wc + __popcll(mm & lt)] = e[u];
                        w0 += __popcll(m0); w1 += __popcll(m1); w2 += __popcll(m2); w3 += __popcll(m3);
                    }
                }
            }
            if (lane == b) { c0 = s0; c1 = s1; c2 = s2; c3 = s3; }
        }
        ORBX_OTICK(2)

        // ---- bookkeeping, a lane per node
        bool keep = act, reached = false;
        if (sorted_phase) {
            const int grow = act ? (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0) - 1 : 0;
            const int incl_g = wave_incl_scan(grow);
            const unsigned long long hit = __ballot(act && size + incl_g >= N);     // `if ((int)lNodes.size() >= N) break;` after this divide
            if (hit != 0ull) { reached = true; keep = act && lane <= __builtin_ctzll(hit); }
        }
        // The children take their node slots before the parents return theirs.  Normally the free list holds enough for the whole
        // batch; when it does not (the pool is only N + 16 nodes, as for the sequential loop), the batch goes in rounds: the longest
        // prefix of the remaining lanes whose children fit.
        for (unsigned long long rem = __ballot(keep); rem != 0ull;) {
            const bool in = (rem >> lane) & 1ull;
            const int ne = in ? (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0) : 0;
            const int nx = in ? (c0 > 1) + (c1 > 1) + (c2 > 1) + (c3 > 1) : 0;
            const int incl_e = wave_incl_scan(ne), incl_x = wave_incl_scan(nx), incl_a = wave_incl_scan(in ? 1 : 0);
            const bool take = in && incl_e <= nfree;
            const unsigned long long m_take = __ballot(take);
            if (m_take == 0ull) { overflow = true; return true; }
            const int last = 63 - __builtin_clzll(m_take);
            const int tot_e = __builtin_amdgcn_readlane(incl_e, last), tot_x = __builtin_amdgcn_readlane(incl_x, last), tot_a = __builtin_amdgcn_readlane(incl_a, last);
            if (np + tot_e > plog_cap || n_ex + tot_x > pool) { overflow = true; return true; }
            if (take) {
                int slot = np + incl_e - ne, fl = nfree - 1 - (incl_e - ne), xs = n_ex + incl_x - nx, cb = beg;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int cn = (c == 0) ? c0 : (c == 1) ? c1 : (c == 2) ? c2 : c3;
                    if (cn > 0) {
                        const int ch = S.freelist[fl--];
                        const int ux = (c & 1) ? midx : ulx, uy = (c & 2) ? midy : uly;
                        S.ulx[ch] = (short)ux; S.uly[ch] = (short)uy;
                        S.brx[ch] = (short)((c & 1) ? brx : midx); S.bry[ch] = (short)((c & 2) ? bry : midy);
                        S.beg[ch] = cb; S.cnt[ch] = cn;
                        S.flg[ch] = (uint8_t)(((src ^ 1) << 1) | (cn == 1 ? 1 : 0));
                        S.plog[slot] = (short)ch; S.pidx[ch] = slot; slot++;            // push_front, children n1..n4 in order
                        if (cn > 1) { SortNode sn; sn.count = cn; sn.ulx = ux; sn.node = ch; ex_new[xs++] = sn; }
                        cb += cn;
                    }
                }
                S.plog[S.pidx[id]] = -1;        // erase the parent
            }
            __syncthreads();                    // the free-list reads above come before the writes below
            np += tot_e; size += tot_e - tot_a; nfree -= tot_e;
            if (take) S.freelist[nfree + incl_a - 1] = (short)id;
            nfree += tot_a;
            n_to_expand += tot_x; n_ex += tot_x;
#ifdef ORBX_OCT_TIMING
            n_div += tot_a;
#endif
            rem &= ~m_take;
            __syncthreads();        // key scatter and node records visible before the next round / batch reads them
        }
        ORBX_OTICK(3)
        return reached;
    };

    bool finish = false;
    int guard = 0;
    while (!finish && !overflow && guard++ < 64) {
        const int prev_size = size;
        int n_to_expand = 0;
        n_ex = 0;
        compact();
        const int np0 = np;                 // children pushed by this pass land behind the pass
        for (int b0 = 0; b0 < np0 && !overflow; b0 += 64) {
            const int idx = np0 - 1 - (b0 + lane);
            int id = -1;
            if (idx >= 0) { const int cur = S.plog[idx]; if (cur >= 0 && !(S.flg[cur] & 1)) id = cur; }
            if (__ballot(id >= 0) != 0ull) divide_batch(id, false, n_to_expand);
        }
        if (size >= N || size == prev_size) {
            finish = true;
        } else if (size + n_to_expand * 3 > N) {
            int guard2 = 0;
            while (!finish && !overflow && guard2++ < 4096) {
                const int prev_size2 = size;
                compact();
                const int n_prev = min(n_ex, pool);
                ORBX_OTICK(4)
                wave_sort_nodes(ex_new, ex_sorted, n_prev, s_sort_stack, S.order);     // std::sort(..., compareNodes) (:700)
                ORBX_OTICK(5)
                n_ex = 0;
                int dummy = 0;
                for (int b0 = 0; b0 < n_prev && !overflow; b0 += 64) {          // for (j = size - 1; j >= 0; j--)
                    const int j = n_prev - 1 - (b0 + lane);
                    const int id = (j >= 0) ? ex_sorted[j].node : -1;
                    if (divide_batch(id, true, dummy)) break;
                }
                if (size >= N || size == prev_size2) finish = true;
            }
        }
    }
    if (overflow) {
        if (lane == 0) { atomicExch(status + frame, ORBX_ERR_INTERNAL); *out_count = 0; }
        return;
    }
    ORBX_OTICK(4)

    // ---- final list order, then the best-response key of every node, first wins ties (:758-776)
    compact();
    for (int k = lane; k < np && k < pool; k += 64) S.order[k] = S.plog[np - 1 - k];
    __syncthreads();
    const int n_out = min(size, L.sel_cap);
    for (int k0 = 0; k0 < n_out; k0 += 64) {
        const int k = k0 + lane;
        int off = 0, cnt = 0;
        if (k < n_out) { const int id = S.order[k]; off = (((S.flg[id] >> 1) & 1) ? cap : 0) + S.beg[id]; cnt = S.cnt[id]; }
        uint32_t best = 0;
        for (int i = 0; __ballot(i < cnt) != 0ull; i += 4) {
            uint32_t e[4];
#pragma unroll
            for (int u = 0; u < 4; u++) e[u] = (i + u < cnt) ? kb[off + i + u] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) if (i + u < cnt && (i + u == 0 || key_resp(e[u]) > key_resp(best))) best = e[u];
        }
        if (k < n_out) out[k] = best;
    }
    if (lane == 0) {
        *out_count = n_out;
        if (size > L.sel_cap) atomicExch(status + frame, ORBX_ERR_INTERNAL);
    }
    ORBX_OTICK(6)
#ifdef ORBX_OCT_TIMING
    if (timed) {
        for (int i = 0; i < 7; i++) d_oct_prof[i] = (unsigned long long)tq[i];
        d_oct_prof[7] = (unsigned long long)(clock64() - t_kernel0);
        d_oct_prof[8] = (unsigned long long)total; d_oct_prof[9] = (unsigned long long)n_div;
    }
#endif
}

template <bool LDS_KEYS, bool LDS_NODES = true>
__global__ __launch_bounds__(64) void k_octree(const LevelDesc* __restrict__ levels, const CellDesc* __restrict__ cells,
                                               const uint32_t* __restrict__ cand, size_t cand_frame_stride,
                                               const int* __restrict__ cell_count, int n_cells,
                                               uint32_t* __restrict__ scratch, size_t scratch_frame_stride,
                                               int pool, int lds_keys_cap,
                                               uint32_t* __restrict__ sel, int sel_frame_stride,
                                               int* __restrict__ sel_count, int n_levels, int* __restrict__ status,
                                               uint8_t* __restrict__ node_scratch, size_t node_stride, int level_first)
{
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int s_sort_stack[3 * kIntrosortStack];       // the sort's explicit stack (in LDS, not in private scratch)
    octree_body<LDS_KEYS, LDS_NODES>(levels, cells, cand, cand_frame_stride, cell_count, n_cells, scratch, scratch_frame_stride, pool, lds_keys_cap,
                                     sel, sel_frame_stride, sel_count, n_levels, status, node_scratch, node_stride, level_first, smem, s_sort_stack);
}

// Small batches (a single frame: the reference's own use) are bound by the latency of the level-0 wave, and a level's key buffers are
// SIZED for the 3x3-NMS worst case (a quarter of the pixels: 614 KB for two buffers at 640 x 480) while a real level holds a few
// thousand candidates.  This kernel counts the level's candidates first and takes the keys-in-LDS body when they fit the LDS it was
// given (no L2 round trip per key-partition step: 0.079 -> 0.062 ms for the octree of one 320 x 200 frame), the scratch body otherwise;
// both bodies are compiled in, each with its own static addressing (no flat accesses).
__global__ __launch_bounds__(64) void k_octree_dyn(const LevelDesc* __restrict__ levels, const CellDesc* __restrict__ cells,
                                                   const uint32_t* __restrict__ cand, size_t cand_frame_stride,
                                                   const int* __restrict__ cell_count, int n_cells,
                                                   uint32_t* __restrict__ scratch, size_t scratch_frame_stride,
                                                   int pool, int lds_keys_cap,
                                                   uint32_t* __restrict__ sel, int sel_frame_stride,
                                                   int* __restrict__ sel_count, int n_levels, int* __restrict__ status,
                                                   uint8_t* __restrict__ node_scratch, size_t node_stride, int level_first)
{
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int s_sort_stack[3 * kIntrosortStack];
    const LevelDesc L = levels[level_first + blockIdx.x];
    const int* cc = cell_count + (size_t)blockIdx.y * n_cells + L.cell_begin;
    int total = 0;
    for (int ci = threadIdx.x; ci < L.cell_count; ci += 64) total += cc[ci];
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    total = __builtin_amdgcn_readfirstlane(total);
    if (total <= lds_keys_cap)
        octree_body<true, true>(levels, cells, cand, cand_frame_stride, cell_count, n_cells, scratch, scratch_frame_stride, pool, lds_keys_cap,
                                sel, sel_frame_stride, sel_count, n_levels, status, node_scratch, node_stride, level_first, smem, s_sort_stack);
    else
        octree_body<false, true>(levels, cells, cand, cand_frame_stride, cell_count, n_cells, scratch, scratch_frame_stride, pool, lds_keys_cap,
                                 sel, sel_frame_stride, sel_count, n_levels, status, node_scratch, node_stride, level_first, smem, s_sort_stack);
}

// test hook: wave_sort_nodes on one array (orbx_debug_wave_sort)
__global__ __launch_bounds__(64) void k_debug_wave_sort(SortNode* __restrict__ data, int n)
{
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ int s_stack[3 * kIntrosortStack];
    SortNode* const v = (SortNode*)smem;
    SortNode* const out = v + n;
    short* const rank_tmp = (short*)(out + n);
    for (int i = threadIdx.x; i < n; i += 64) v[i] = data[i];
    __syncthreads();
    wave_sort_nodes(v, out, n, s_stack, rank_tmp);
    for (int i = threadIdx.x; i < n; i += 64) data[i] = out[i];
}

// ------------------------------------------------------------------------------------------------
// E8 (index part): slot of every selected keypoint in the output arrays.  Level-major order; keypoints
// whose scaled x lies in [lap0, lap1] are written from the back (stereoIndex--), the rest from the front.
// One wave per frame.  meta[k] = (level << 24 | rank within level), dst[k] = output row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_index(const LevelDesc* __restrict__ levels, int n_levels,
                                              const uint32_t* __restrict__ sel, int sel_frame_stride,
                                              const int* __restrict__ sel_count, int lap0, int lap1, int cap,
                                              int* __restrict__ kp_dst, int kp_frame_stride,
                                              int* __restrict__ n_out, int* __restrict__ mono_out, int* __restrict__ status)
{
    const int frame = blockIdx.x, lane = threadIdx.x;
    const int* sc = sel_count + (size_t)frame * n_levels;
    int total = 0;
    for (int l = 0; l < n_levels; l++) total += sc[l];
    int* dst = kp_dst + (size_t)frame * kp_frame_stride;
    int mono = 0, stereo = total - 1;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int l = 0; l < n_levels; l++) {
        const LevelDesc L = levels[l];
        const uint32_t* s = sel + (size_t)frame * sel_frame_stride + L.sel_off;
        const int n = sc[l];
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            bool valid = k < n, lap = false;
            if (valid) {
                float x = (float)(key_x(s[k]) + 16);
                if (l != 0) x = x * L.scale;
                lap = (x >= (float)lap0) && (x <= (float)lap1);
            }
            const unsigned long long ml = __ballot(valid && lap), mm = __ballot(valid && !lap);
            if (valid) dst[L.sel_off + k] = lap ? stereo - __popcll(ml & lt) : mono + __popcll(mm & lt);
            stereo -= __popcll(ml);
            mono += __popcll(mm);
        }
    }
    if (lane == 0) {
        n_out[frame] = total;
        mono_out[frame] = mono;
        if (total > cap) atomicExch(status + frame, ORBX_ERR_CAPACITY);
    }
}

// ------------------------------------------------------------------------------------------------
// E6: GaussianBlur 7x7, sigma 2, BORDER_REFLECT_101 -- OpenCV 4.x fixed-point path:
// horizontal u8 x Q8.8 taps -> Q8.8 (u16), vertical Q8.8 x Q8.8 -> Q16.16, (v + 0x8000) >> 16.
// 256 threads per 64x32 output tile; the (64+6) x (32+6) source window is staged in LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = (p < 0) ? -p : 2 * len - 2 - p;
    return p;
}

constexpr int kBlurTW = 64, kBlurTH = 58;        // outputs per tile; 58 + 6 = 64 staged rows = 32 row pairs

// 256 threads per 64 x 58 output tile.  The (64 + 6) x (58 + 6) source window is staged in LDS as 16-byte chunks (rows start
// 16 bytes left of the tile so that every chunk is an aligned dwordx4 load; chunks that touch the image border are
// assembled byte by byte with BORDER_REFLECT_101).
//   horizontal  one work item = 2 rows x 4 columns: per output two byte dot products (v_dot4_u32_u8) on v_alignbyte windows;
//               the Q8.8 results of the two rows are stored as ONE dword per column (row 2p | row 2p+1 << 16)
//   vertical    one work item = 2 rows x 4 columns: the 7 taps of an output are 4 x v_dot2_u32_u16 on those row pairs
//               (even output row: (t0 t1)(t2 t3)(t2 t1)(t0 0), odd: (0 t0)(t1 t2)(t3 t2)(t1 t0)), rounding folded into the
//               first accumulator, the four result bytes gathered with v_perm.
// raw_out != nullptr (level 0 taken from the caller's image): the tile's own pixels are also copied into the pyramid, which
// is what later readers of level 0 (descriptors, stereo matching, the mvImagePyramid getter) use.
__global__ __launch_bounds__(256) void k_blur(SrcImage lvl0, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ raw_out, uint8_t* __restrict__ blur,
                                              size_t frame_stride, const LevelDesc* __restrict__ levels, const TileDesc* __restrict__ tiles, int n_tiles,
                                              int t0, int t1, int t2, int t3)
{
    constexpr int SP = kBlurTW + 32, SR = kBlurTH + 6;      // LDS row = source columns x0-16 .. x0+79 (6 chunks); staged rows
    __shared__ __align__(16) uint8_t s_src[SR * SP];
    __shared__ __align__(16) uint32_t s_hp[(SR / 2) * kBlurTW];
    const int tile_idx = xcd_remap(blockIdx.x, gridDim.x, blockIdx.y);     // contiguous tile range per XCD (see k_fast_strips)
    if (tile_idx >= n_tiles) return;
    const TileDesc T = tiles[tile_idx];
    const LevelDesc L = levels[T.level];
    const bool ext = T.level == 0 && lvl0.base != nullptr;
    const uint8_t* img = ext ? lvl0.base + (size_t)blockIdx.y * lvl0.frame_stride : pyr + (size_t)blockIdx.y * frame_stride + L.off;
    const int istride = ext ? lvl0.stride : L.stride;
    uint8_t* dst = blur + (size_t)blockIdx.y * frame_stride + L.off;
    const int tid = threadIdx.x;
    const int x0 = T.x0, y0 = T.y0;
    for (int i = tid; i < SR * (SP / 16); i += 256) {
        const int r = i / (SP / 16), q = i - r * (SP / 16);
        const int sy = reflect101(y0 + r - 3, L.h), cx = x0 - 16 + 16 * q;
        // A chunk that starts inside the row is one aligned load (its bytes past the last column are row padding: they only reach
        // outputs that are not stored).  Only the three columns left of column 0 and the three right of column w - 1 can be read by
        // stored outputs without being pixels: they are patched in with BORDER_REFLECT_101 (six byte tests per border chunk --
        // a byte-by-byte assembly of such chunks cost more than the whole filter of an inner tile).
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        const uint8_t* rowp = img + (size_t)sy * istride;
        if (cx >= 0 && cx < L.w) v = *(const uint4*)(rowp + cx);
        if (cx < 0 || cx + 16 > L.w) {
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 1; k <= 3; k++) {
                const int bl = -k - cx, br = L.w - 1 + k - cx;              // byte positions of columns -k and w - 1 + k in this chunk
                if (bl >= 0 && bl < 16) { const uint32_t px = rowp[reflect101(-k, L.w)]; w[bl >> 2] = (w[bl >> 2] & ~(0xFFu << (8 * (bl & 3)))) | (px << (8 * (bl & 3))); }
                if (br >= 0 && br < 16) { const uint32_t px = rowp[reflect101(L.w - 1 + k, L.w)]; w[br >> 2] = (w[br >> 2] & ~(0xFFu << (8 * (br & 3)))) | (px << (8 * (br & 3))); }
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *(uint4*)(s_src + r * SP + 16 * q) = v;
    }
    __syncthreads();
    if (ext && raw_out != nullptr) {        // level 0 of the pyramid = the caller's image (what k_copy_level0 would have written)
        uint8_t* raw = raw_out + (size_t)blockIdx.y * frame_stride + L.off;
        for (int i = tid; i < kBlurTH * (kBlurTW / 4); i += 256) {
            const int r = i >> 4, q = i & 15;
            if (y0 + r < L.h && x0 + 4 * q < L.w) *(uint32_t*)(raw + (size_t)(y0 + r) * L.stride + x0 + 4 * q) = *(const uint32_t*)(s_src + (r + 3) * SP + 16 + 4 * q);
        }
    }
    const uint32_t tapA = (uint32_t)t0 | ((uint32_t)t1 << 8) | ((uint32_t)t2 << 16) | ((uint32_t)t3 << 24);
    const uint32_t tapB = (uint32_t)t2 | ((uint32_t)t1 << 8) | ((uint32_t)t0 << 16);
    // ---- horizontal: output column 4q+k of a row uses source bytes k+1 .. k+7 of the 12 bytes w0 w1 w2 (taps (t0 t1 t2 t3) on
    // k+1..k+4, (t2 t1 t0 0) on k+5..k+8).  The taps sum to 256, so a sum never exceeds 255 * 256 = 65280: it fits 16 bits. ----
    for (int i = tid; i < (SR / 2) * (kBlurTW / 4); i += 256) {
        const int p = i >> 4, q = i & 15;
        uint32_t o[2][4];
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const uint32_t* w = (const uint32_t*)(s_src + (2 * p + rr) * SP + 12) + q;
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            o[rr][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), tapA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), tapB, 0u, false), false);
            o[rr][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), tapA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), tapB, 0u, false), false);
            o[rr][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), tapA, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), tapB, 0u, false), false);
            o[rr][3] = __builtin_amdgcn_udot4(w1, tapA, __builtin_amdgcn_udot4(w2, tapB, 0u, false), false);
        }
        *(uint4*)(s_hp + p * kBlurTW + 4 * q) = make_uint4(o[0][0] | (o[1][0] << 16), o[0][1] | (o[1][1] << 16), o[0][2] | (o[1][2] << 16), o[0][3] | (o[1][3] << 16));
    }
    __syncthreads();
    // ---- vertical ----
    const us2 cE0 = as_us2((uint32_t)t0 | ((uint32_t)t1 << 16)), cE1 = as_us2((uint32_t)t2 | ((uint32_t)t3 << 16)),
              cE2 = as_us2((uint32_t)t2 | ((uint32_t)t1 << 16)), cE3 = as_us2((uint32_t)t0);
    const us2 cO0 = as_us2((uint32_t)t0 << 16), cO1 = as_us2((uint32_t)t1 | ((uint32_t)t2 << 16)),
              cO2 = as_us2((uint32_t)t3 | ((uint32_t)t2 << 16)), cO3 = as_us2((uint32_t)t1 | ((uint32_t)t0 << 16));
    for (int i = tid; i < (kBlurTH / 2) * (kBlurTW / 4); i += 256) {
        const int m = i >> 4, q = i & 15;
        const int oy = y0 + 2 * m, ox = x0 + 4 * q;
        if (oy >= L.h || ox >= L.w) continue;
        const uint4 P0 = *(const uint4*)(s_hp + m * kBlurTW + 4 * q), P1 = *(const uint4*)(s_hp + (m + 1) * kBlurTW + 4 * q),
                    P2 = *(const uint4*)(s_hp + (m + 2) * kBlurTW + 4 * q), P3 = *(const uint4*)(s_hp + (m + 3) * kBlurTW + 4 * q);
        const uint32_t a0[4] = {P0.x, P0.y, P0.z, P0.w}, a1[4] = {P1.x, P1.y, P1.z, P1.w}, a2[4] = {P2.x, P2.y, P2.z, P2.w}, a3[4] = {P3.x, P3.y, P3.z, P3.w};
        uint32_t e[4], o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {       // Q16.16 sums + 0x8000: the result byte is bits 16..23 (at most 255 * 65536 + 32768)
            e[k] = __builtin_amdgcn_udot2(as_us2(a3[k]), cE3, __builtin_amdgcn_udot2(as_us2(a2[k]), cE2, __builtin_amdgcn_udot2(as_us2(a1[k]), cE1,
                   __builtin_amdgcn_udot2(as_us2(a0[k]), cE0, 0x8000u, false), false), false), false);
            o[k] = __builtin_amdgcn_udot2(as_us2(a3[k]), cO3, __builtin_amdgcn_udot2(as_us2(a2[k]), cO2, __builtin_amdgcn_udot2(as_us2(a1[k]), cO1,
                   __builtin_amdgcn_udot2(as_us2(a0[k]), cO0, 0x8000u, false), false), false), false);
        }
        // bytes 2 of four accumulators -> one dword
        const uint32_t pe = __builtin_amdgcn_perm(__builtin_amdgcn_perm(e[3], e[2], 0x0C0C0602u), __builtin_amdgcn_perm(e[1], e[0], 0x0C0C0602u), 0x05040100u);
        const uint32_t po = __builtin_amdgcn_perm(__builtin_amdgcn_perm(o[3], o[2], 0x0C0C0602u), __builtin_amdgcn_perm(o[1], o[0], 0x0C0C0602u), 0x05040100u);
        *(uint32_t*)(dst + (size_t)oy * L.stride + ox) = pe;                       // the row padding absorbs the tail of the last dword
        if (oy + 1 < L.h) *(uint32_t*)(dst + (size_t)(oy + 1) * L.stride + ox) = po;
    }
}

// ------------------------------------------------------------------------------------------------
// E5 + E7 + E8: half a wave per keypoint.  IC_Angle on the unblurred level, steered BRIEF on the blurred
// level, then the cv::KeyPoint record and the 32-byte descriptor are written to their output row.
// ------------------------------------------------------------------------------------------------
// mask of the radius-15 disc (umax, :453-468) per patch row, as byte masks of the 8 dwords that hold columns u = -15 .. 16
struct IcMask { uint32_t m[32][8]; };
constexpr IcMask make_ic_mask()
{
    IcMask t{};
    const int um[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    for (int r = 0; r < 31; r++) {
        const int v = r - 15, av = v < 0 ? -v : v;
        for (int j = 0; j < 8; j++)
            for (int k = 0; k < 4; k++) {
                const int u = -15 + 4 * j + k, au = u < 0 ? -u : u;
                if (u <= 15 && au <= um[av]) t.m[r][j] |= 0xFFu << (8 * k);
            }
    }
    return t;
}
__device__ const IcMask d_ic_mask = make_ic_mask();

__global__ __launch_bounds__(256) void k_orient_desc(const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur, size_t frame_stride,
                                                     const LevelDesc* __restrict__ levels, int n_levels,
                                                     const uint32_t* __restrict__ sel, int sel_frame_stride,
                                                     const int* __restrict__ sel_count,
                                                     const int* __restrict__ kp_dst, int kp_frame_stride,
                                                     OrbxKeyPoint* __restrict__ kps, uint8_t* __restrict__ desc, int cap,
                                                     OrbxKeyPoint* __restrict__ lvl_kps /* optional: per-level keypoints, level coords */,
                                                     int quads_per_frame, int batch)
{
    // TWO keypoints per wave, one per half: the intensity-centroid rows (31 lanes each), the angle, its sine and cosine and
    // the bookkeeping are the same instructions for both halves, and a lane steers 8 of its keypoint's 256 pairs instead of 4 --
    // a third fewer wave instructions per keypoint than one keypoint per wave.
    // Everything a keypoint needs -- the 31x31 disc of the unblurred level (IC_Angle) and the 37x37 window of the blurred level
    // that the rotated pattern can reach (|coordinate| <= 18 < EDGE_THRESHOLD) -- is fetched up front with wide loads (the address
    // unit takes 16 clocks per wave-wide load whatever its width), so a wave sees two dependent memory round trips (selection
    // record, patches).
    constexpr int kBR = 18, kBP = 64, kPR = 15;                 // blur: 37 rows x 4 chunks
    constexpr int kBBytes = (2 * kBR + 1) * kBP;
    __shared__ __align__(16) uint8_t s_patch[8][kBBytes];
    // 1-D grid of octets_per_frame * batch workgroups (8 keypoints each).  Workgroup w runs on XCD w % 8: giving XCD k the
    // contiguous range [k * total / 8, (k + 1) * total / 8) of (frame, octet) pairs keeps all the patches of a frame in ONE L2
    // (2.4 MB of pyramid + blur per frame against 4 MB of L2) instead of fetching every frame into all eight.
    const int total_wg = gridDim.x, per_xcd = total_wg >> 3;            // the host pads the grid to a multiple of 8
    const int v_id = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int frame = v_id / quads_per_frame, quad = v_id - frame * quads_per_frame;
    if (frame >= batch) return;                                         // grid padding (uniform for the workgroup)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, hl = lane & 31;
    // the selection slots of all levels are one flat range [0, sel_frame_stride): a half-wave takes slot s and finds its level
    // from the level offsets, so no workgroup is launched for slots a level does not have
    // (the levels' slot ranges start at even offsets, so both halves of a wave are in the same level: scalar table loads)
    const int s_pair = quad * 8 + wave * 2, s_flat = s_pair + half;
    int level = 0;
    for (int l = 1; l < n_levels; l++) level = (s_pair >= levels[l].sel_off) ? l : level;
    const LevelDesc L = levels[level];
    const int k = s_flat - L.sel_off;
    const bool live = s_flat < sel_frame_stride && k < L.sel_cap && k < sel_count[(size_t)frame * n_levels + level];
    const int slot = L.sel_off + (live ? k : 0);
    const uint32_t e = live ? sel[(size_t)frame * sel_frame_stride + slot] : 0u;
    const int row = live ? kp_dst[(size_t)frame * kp_frame_stride + slot] : -1;
    uint32_t pw[8];
#pragma unroll
    for (int q = 0; q < 8; q++) pw[q] = ((const uint32_t*)d_pattern)[hl + 32 * q];       // one dword = (x0, y0, x1, y1) as int8
    const int px = (int)key_x(e) + 16, py = (int)key_y(e) + 16;     // level coordinates (:885-886)
    uint8_t* sb = s_patch[wave * 2 + half];
    const int xb = (px - kBR) & ~15;                                // 16-byte aligned window start (rows are 64-B aligned)
    // The time of this kernel is the number of keypoints a CU holds in flight over their two memory round trips (measured: it
    // scales with the LDS a workgroup is given), so LDS only holds what is addressed at random -- the blurred window.  The
    // intensity-centroid rows go straight into the registers of the lanes that own them: row r of the 31 x 31 patch = the 9 dwords
    // from the dword that holds column u = -15 on (dword-aligned wide loads).
    uint32_t icw[9];
#pragma unroll
    for (int j = 0; j < 9; j++) icw[j] = 0u;
    if (live) {
        const uint8_t* bbase = blur + (size_t)frame * frame_stride + L.off + (size_t)(py - kBR) * L.stride + xb;
        for (int i = hl; i < (2 * kBR + 1) * 4; i += 32) {
            const int r = i >> 2, c = i & 3;
            *(uint4*)(sb + r * kBP + 16 * c) = *(const uint4*)(bbase + (size_t)r * L.stride + 16 * c);
        }
        if (hl < 2 * kPR + 1) {
            const uint8_t* rp = pyr + (size_t)frame * frame_stride + L.off + (size_t)(py - kPR + hl) * L.stride + ((px - kPR) & ~3);
            const Dwords4 q0 = *(const Dwords4*)rp, q1 = *(const Dwords4*)(rp + 16);
            icw[0] = q0.x; icw[1] = q0.y; icw[2] = q0.z; icw[3] = q0.w; icw[4] = q1.x; icw[5] = q1.y; icw[6] = q1.z; icw[7] = q1.w;
            icw[8] = *(const uint32_t*)(rp + 32);
        }
    }
    // the LDS window is private to the (half-)wave: its own writes are ordered before its own reads by the LDS queue; only the
    // compiler has to be kept from reordering them -- no workgroup barrier, so a wave never waits for its three neighbours
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (__ballot(live) == 0ull) return;

    // ---- IC_Angle (:76-103): lane r < 31 of a half owns patch row v = r - 15: its 31 bytes as 8 byte-aligned dwords, masked to
    // the disc; the row sums  a = sum I  and  b = sum (u + 15) I  are byte dot products, m10 += b - 15 a, m01 += v a ----
    int m10 = 0, m01 = 0;
    if (live && hl < 2 * kPR + 1) {
        const uint32_t sh = (uint32_t)((px - kPR) & 3);
        const uint4 ma = *(const uint4*)&d_ic_mask.m[hl][0], mb = *(const uint4*)&d_ic_mask.m[hl][4];
        const uint32_t mk[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
        uint32_t a = 0u, b = 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t d = __builtin_amdgcn_alignbyte(icw[j + 1], icw[j], sh) & mk[j];
            a = __builtin_amdgcn_udot4(d, 0x01010101u, a, false);
            b = __builtin_amdgcn_udot4(d, 0x03020100u + 0x04040404u * (uint32_t)j, b, false);
        }
        m10 = (int)b - 15 * (int)a;
        m01 = (hl - kPR) * (int)a;
    }
    for (int o = 16; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }     // (inside the half)
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // ---- steered BRIEF: a lane handles the pairs hl, hl + 32, ..., hl + 224 of its keypoint; ballot q = the descriptor's 32-bit
    // word q of the lower-half keypoint in its low half, of the upper-half keypoint in its high half
    const float factorPI = (float)(3.141592653589793238462643383279502884 / 180.f);
    float a, b;
    sincos_f32(angle * factorPI, &a, &b);
    const uint8_t* bimg = sb + kBR * kBP + (px - xb);
    unsigned long long w[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const float x0 = (float)(signed char)(pw[q] & 0xFF), y0 = (float)(signed char)((pw[q] >> 8) & 0xFF);
        const float x1 = (float)(signed char)((pw[q] >> 16) & 0xFF), y1 = (float)(signed char)(pw[q] >> 24);
        int t0 = 0, t1 = 0;
        if (live) {
            t0 = bimg[cv_round_f(x0 * b + y0 * a) * kBP + cv_round_f(x0 * a - y0 * b)];
            t1 = bimg[cv_round_f(x1 * b + y1 * a) * kBP + cv_round_f(x1 * a - y1 * b)];
        }
        w[q] = __ballot(t0 < t1);
    }
    if (!live) return;
    if (hl == 0) {
        OrbxKeyPoint kp;
        kp.x = (float)px; kp.y = (float)py;
        kp.size = (float)L.patch_size; kp.angle = angle; kp.response = (float)key_resp(e);
        kp.octave = level; kp.class_id = -1;
        if (lvl_kps) lvl_kps[(size_t)frame * sel_frame_stride + slot] = kp;
        if (level != 0) { kp.x = kp.x * L.scale; kp.y = kp.y * L.scale; }      // :1149-1151
        if (row >= 0 && row < cap) kps[(size_t)frame * cap + row] = kp;
    }
    if (hl < 8 && row >= 0 && row < cap) {
        unsigned long long mine = w[0];
#pragma unroll
        for (int q = 1; q < 8; q++) mine = (hl == q) ? w[q] : mine;
        ((uint32_t*)(desc + ((size_t)frame * cap + row) * 32))[hl] = (uint32_t)(mine >> (32 * half));
    }
}

// ------------------------------------------------------------------------------------------------
// Frame::ComputeStereoMatches (reference src/Frame.cc:931-1101), SURVEY 8(f) rank 4.  One wave per left key point:
// the row table of the reference (vRowIndices) becomes a lane-strided scan of the right key points with the same band
// test; first minimum through a (distance, index) key; then the 11 x 11 SAD sliding window over 11 shifts on the left
// key point's pyramid level, parabola fit and the disparity gates, in the reference's float expressions.
// ------------------------------------------------------------------------------------------------
struct StereoTables { float scale[16], inv_scale[16]; };

__global__ __launch_bounds__(256) void k_stereo_match(const uint8_t* __restrict__ pyrL, const uint8_t* __restrict__ pyrR, size_t frame_stride,
                                                     const LevelDesc* __restrict__ levels, StereoTables T,
                                                     const OrbxKeyPoint* __restrict__ kpsL, const uint8_t* __restrict__ descL, const int32_t* __restrict__ nL,
                                                     const OrbxKeyPoint* __restrict__ kpsR, const uint8_t* __restrict__ descR, const int32_t* __restrict__ nR,
                                                     int cap, float mb, float mbf, int n_levels,
                                                     float* __restrict__ u_right, float* __restrict__ depth, int32_t* __restrict__ sad_out)
{
    __shared__ uint8_t s_l[4][11 * 12], s_r[4][11 * 24];
    __shared__ int s_part[4][11 * 11];
    const int frame = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int iL = blockIdx.x * 4 + wave;
    const int n_l = min(nL[frame], cap), n_r = min(nR[frame], cap);
    const bool live = iL < n_l;
    const size_t fo = (size_t)frame * cap;
    // ---- best right key point by descriptor distance (:973-1008) ----
    float uL = 0, vL = 0;
    int levelL = 0;
    unsigned key = 0xFFFFFFFFu;
    if (live) {
        const OrbxKeyPoint kl = kpsL[fo + iL];
        uL = kl.x; vL = kl.y; levelL = kl.octave;
        const int v_row = (int)vL;                                           // vRowIndices[vL]
        const float minZ = mb, minD = 0, maxD = mbf / minZ;
        const float minU = uL - maxD, maxU = uL - minD;
        const unsigned long long* dl = (const unsigned long long*)(descL + (fo + iL) * 32);
        const unsigned long long d0 = dl[0], d1 = dl[1], d2 = dl[2], d3 = dl[3];
        for (int iR = lane; iR < n_r; iR += 64) {
            const OrbxKeyPoint kr = kpsR[fo + iR];
            const float r = 2.0f * T.scale[kr.octave & 15];
            const int maxr = (int)ceilf(kr.y + r), minr = (int)floorf(kr.y - r);
            if (v_row < minr || v_row > maxr) continue;                      // the row table of :941-960
            if (kr.octave < levelL - 1 || kr.octave > levelL + 1) continue;
            if (!(kr.x >= minU && kr.x <= maxU)) continue;
            const unsigned long long* dr = (const unsigned long long*)(descR + (fo + iR) * 32);
            const int dist = __popcll(d0 ^ dr[0]) + __popcll(d1 ^ dr[1]) + __popcll(d2 ^ dr[2]) + __popcll(d3 ^ dr[3]);
            if (dist < 100) key = min(key, ((unsigned)dist << 20) | (unsigned)iR);     // bestDist starts at TH_HIGH; first minimum
        }
    }
    for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, o));
    const int thOrbDist = (100 + 50) / 2;
    bool ok = live && key != 0xFFFFFFFFu && (int)(key >> 20) < thOrbDist;
    if (levelL < 0 || levelL >= n_levels) ok = false;          // key points that do not come from this extractor: no table lookups out of range
    // ---- sub-pixel match by correlation (:1011-1083) ----
    float scaleduR0 = 0;
    int y0 = 0, xl0 = 0, xr_start = 0;
    size_t loff = 0;
    int lstride = 0;
    if (ok) {
        const float uR0 = kpsR[fo + (key & 0xFFFFFu)].x;
        const float scaleFactor = T.inv_scale[levelL & 15];
        const float scaleduL = roundf(uL * scaleFactor), scaledvL = roundf(vL * scaleFactor);
        scaleduR0 = roundf(uR0 * scaleFactor);
        const LevelDesc L = levels[levelL];
        loff = (size_t)L.off; lstride = L.stride;
        const int w = 5, Ls = 5;
        const float iniu = scaleduR0 + Ls - w, endu = scaleduR0 + Ls + w + 1;
        if (iniu < 0 || endu >= (float)L.w) ok = false;
        y0 = (int)(scaledvL - w); xl0 = (int)(scaleduL - w); xr_start = (int)(scaleduR0 - Ls - w);
        // cv::Mat::rowRange / colRange would assert outside the level image; key points of the extractor never get here
        if (y0 < 0 || y0 + 11 > L.h || xl0 < 0 || xl0 + 11 > L.w || xr_start < 0 || xr_start + 21 > L.w) ok = false;
    }
    if (ok) {
        const uint8_t* il = pyrL + (size_t)frame * frame_stride + loff;
        const uint8_t* ir = pyrR + (size_t)frame * frame_stride + loff;
        for (int i = lane; i < 121; i += 64) { const int r = i / 11, c = i - r * 11; s_l[wave][r * 12 + c] = il[(size_t)(y0 + r) * lstride + xl0 + c]; }
        for (int i = lane; i < 231; i += 64) { const int r = i / 21, c = i - r * 21; s_r[wave][r * 24 + c] = ir[(size_t)(y0 + r) * lstride + xr_start + c]; }
    }
    __syncthreads();
    if (ok) {
        for (int i = lane; i < 121; i += 64) {
            const int sft = i / 11, r = i - sft * 11;
            int acc = 0;
#pragma unroll
            for (int c = 0; c < 11; c++) acc += abs((int)s_l[wave][r * 12 + c] - (int)s_r[wave][r * 24 + sft + c]);
            s_part[wave][sft * 11 + r] = acc;
        }
    }
    __syncthreads();
    if (ok && lane == 0) {
        const int Ls = 5;
        float vDists[11];
        int bestDist = 0x7FFFFFFF, bestincR = 0;
        for (int sft = 0; sft < 11; sft++) {
            int sad = 0;
            for (int r = 0; r < 11; r++) sad += s_part[wave][sft * 11 + r];
            const float dist = (float)sad;                                   // cv::norm(IL, IR, NORM_L1)
            if (dist < (float)bestDist) { bestDist = (int)dist; bestincR = sft - Ls; }
            vDists[sft] = dist;
        }
        float ur_out = -1.0f, depth_out = -1.0f;
        int sad_best = -1;
        if (!(bestincR == -Ls || bestincR == Ls)) {
            const float dist1 = vDists[Ls + bestincR - 1], dist2 = vDists[Ls + bestincR], dist3 = vDists[Ls + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (!(deltaR < -1 || deltaR > 1)) {
                float bestuR = T.scale[levelL & 15] * ((float)scaleduR0 + (float)bestincR + deltaR);
                float disparity = (uL - bestuR);
                const float minD = 0, maxD = mbf / mb;
                if (disparity >= minD && disparity < maxD) {
                    if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }      // double literals, as in :1073-1074
                    depth_out = mbf / disparity;
                    ur_out = bestuR;
                    sad_best = bestDist;
                }
            }
        }
        u_right[fo + iL] = ur_out; depth[fo + iL] = depth_out; sad_out[fo + iL] = sad_best;
    } else if (live && lane == 0) {
        u_right[fo + iL] = -1.0f; depth[fo + iL] = -1.0f; sad_out[fo + iL] = -1;
    }
}

// median cut of :1086-1100: matches whose SAD is >= 1.5 * 1.4 * median are dropped.  One workgroup per frame.
__global__ __launch_bounds__(256) void k_stereo_median(const int32_t* __restrict__ nL, int cap, int n_pow2_max,
                                                      float* __restrict__ u_right, float* __restrict__ depth, const int32_t* __restrict__ sad)
{
    extern __shared__ int s_sad[];       // n_pow2_max
    __shared__ int s_cnt;
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int n = min(nL[frame], cap);
    int n_pow2 = 2;                      // this frame's own sort size (the launch's LDS is sized for the arrays' capacity)
    while (n_pow2 < n) n_pow2 <<= 1;
    n_pow2 = min(n_pow2, n_pow2_max);
    const size_t fo = (size_t)frame * cap;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    int local = 0;
    for (int i = tid; i < n_pow2; i += 256) {
        const int v = (i < n) ? sad[fo + i] : -1;
        s_sad[i] = v >= 0 ? v : 0x7FFFFFFF;
        local += v >= 0;
    }
    if (local) atomicAdd(&s_cnt, local);
    __syncthreads();
    // bitonic sort by the 4 waves: wave w owns chunk w (n_pow2 / 4 values); exchanges with a stride below the chunk size stay inside it
    // (the wave's own program order), only the three passes across chunks take a workgroup barrier
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int chunk = n_pow2 >> 2, pairs = n_pow2 >> 3;
        bool need_block = false;             // (the barrier above made the values visible)
        for (int k = 2; k <= n_pow2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                const bool cross = j >= chunk || n_pow2 < 8;
                if (cross || need_block) __syncthreads();
                else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
                need_block = cross;
                const int t0 = n_pow2 >= 8 ? wave * pairs + lane : tid, t1 = n_pow2 >= 8 ? (wave + 1) * pairs : (n_pow2 >> 1), ts = n_pow2 >= 8 ? 64 : 256;
                for (int t = t0; t < t1; t += ts) {
                    const int lo = 2 * t - (t & (j - 1)), hi = lo + j;
                    const bool up = (lo & k) == 0;
                    const int a = s_sad[lo], b = s_sad[hi];
                    if ((a > b) == up) { s_sad[lo] = b; s_sad[hi] = a; }
                }
            }
        __syncthreads();
    }
    const int m = s_cnt;
    if (m == 0) return;
    const float median = (float)s_sad[m / 2];
    const float thDist = 1.5f * 1.4f * median;
    for (int i = tid; i < n; i += 256) {
        const int v = sad[fo + i];
        if (v >= 0 && !((float)v < thDist)) { u_right[fo + i] = -1; depth[fo + i] = -1; }
    }
}

}  // namespace orbx
