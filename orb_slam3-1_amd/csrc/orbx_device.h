// orbx_device.h -- POD descriptors shared between the host driver and the gfx950 kernels.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "../../include/orbslam3_hip.h"

namespace orbx {

constexpr int kEdge = 19;          // EDGE_THRESHOLD   (reference src/ORBextractor.cc:73)
constexpr int kHalfPatch = 15;     // HALF_PATCH_SIZE  (:72)
constexpr int kPatch = 31;         // PATCH_SIZE       (:71)
constexpr int kMaxLevels = 16;
constexpr int kOctLogFactor = 6;     // octree push log capacity = 6 x node pool (<= 4 pushes per alive node between compactions)

// One pyramid level of one frame inside the per-frame pyramid buffer.
struct LevelDesc {
    int32_t w, h, stride;      // stride: bytes, multiple of 64
    int32_t nfeat;             // mnFeaturesPerLevel[level]
    int64_t off;               // byte offset inside the per-frame buffer, multiple of 64
    int32_t cell_begin, cell_count;   // range in the cell table
    int32_t cand_off, cand_cap;       // candidate slots of this level inside the per-frame candidate buffer
    int32_t sel_off, sel_cap;         // selected-keypoint slots (nfeat + 3)
    float scale;               // mvScaleFactor[level]
    int32_t patch_size;        // int(PATCH_SIZE * scale)   (:880)
};

// One FAST cell (reference :797-822): sub-image [x0,x1) x [y0,y1) in level coordinates.
struct CellDesc {
    int16_t level;
    int16_t x0, y0, x1, y1;
    int32_t slot_off, slot_cap;    // where this cell's keypoints go in the per-frame candidate buffer
};

struct TileDesc {
    int16_t level, x0, y0, pad;
};

// Workgroups are dispatched round-robin over the 8 XCDs, so workgroups b and b+8 share an L2 (MI355X_MICROARCH.md,
// "Workgroup dispatch").  With a grid of 8*chunk workgroups per frame this gives every XCD a contiguous run of `chunk` items
// (neighbouring items share image rows / 128-B lines: they stay in one L2); the run an XCD gets rotates with the frame, so
// that cheap and expensive runs (pyramid levels differ) spread evenly over the XCDs.  A speed-only mapping: any placement
// yields the same results.  The grid is padded to a multiple of 8 only -- empty workgroups still cost a dispatch slot, and the
// dispatcher starts only about two workgroups per nanosecond.
__host__ __device__ inline int xcd_grid(int n_items) { return 8 * ((n_items + 7) / 8); }
__host__ __device__ inline int xcd_remap(int block, int grid, int frame)
{
    const int chunk = grid >> 3;
    return (((block & 7) + frame) & 7) * chunk + (block >> 3);
}

// candidate / key record: x (12 bits), y (12 bits) relative to minBorder, FAST response (8 bits)
__host__ __device__ inline uint32_t pack_key(uint32_t x, uint32_t y, uint32_t resp) { return (x << 20) | (y << 8) | resp; }
__host__ __device__ inline uint32_t key_x(uint32_t e) { return e >> 20; }
__host__ __device__ inline uint32_t key_y(uint32_t e) { return (e >> 8) & 0xFFFu; }
__host__ __device__ inline uint32_t key_resp(uint32_t e) { return e & 0xFFu; }

}  // namespace orbx
