// grid_layout.h -- per-frame neighbour grid as laid out in the caller's buffer (grid.hip builds and searches it; neighbor.hip's
// scan reads the header's density estimate to take over dense frames).
#pragma once
#include "common.h"

#define GRID_DIM_MAX 128                         // cells per axis: 64 (three_nn: ~1 known point per cell) or 128 (ball query:
#define GRID_CELLS_MAX (GRID_DIM_MAX * GRID_DIM_MAX)   //  crowded near-sensor cells stay small); buffers are sized for 128

struct GridHeader {
    float x0, z0, inv_cs, cs;
    int dim;                                      // cells per axis of THIS grid
    // Estimated candidates a ball query of radius min_cell visits: (points per occupied cell) x (cells its square of
    // cell columns overlaps).  A frame whose estimate exceeds N / 32 is DENSE: the per-thread sorted hit lists of the grid
    // kernel cost ~50x a scan's packed-fp32 pair test per candidate, so there the index-order scan (which also exits as soon
    // as both lists are full) is the faster exact method.  Both kernels are launched; each exits at once on the other's frames.
    float cand;
    int pad1, pad2;
};
// per-frame block inside the caller's buffer: header | cell_start[128*128 + 1] | sorted[N] float4
static inline size_t grid_frame_bytes(int N) {
    size_t b = sizeof(GridHeader) + (size_t)(GRID_CELLS_MAX + 1) * 4;
    b = (b + 15) & ~(size_t)15;
    return b + (size_t)N * 16;
}
__device__ __forceinline__ const GridHeader* grid_header(const void* g, size_t fb, int b) { return (const GridHeader*)((const char*)g + fb * b); }
__device__ __forceinline__ bool grid_frame_dense(const GridHeader& H, int N) { return H.cand > (float)N * (1.0f / 32.0f); }

// neighbor.hip: the scan ball query restricted to the DENSE frames of a grid (grid == nullptr: every frame)
int prcnn_launch_ball_query_scan(const float* xyz, const float* new_xyz, int B, int N, int M, float radius_a, int nsample_a,
                                 int32_t* idx_a, float radius_b, int nsample_b, int32_t* idx_b, const void* grid, hipStream_t s);
