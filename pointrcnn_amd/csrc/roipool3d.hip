// roipool3d.hip -- canonical-RoI point pooling for gfx950, one fused pass.
//
// Replaces roipool3d_cuda.forward (lib/utils/roipool3d/src/roipool3d.cpp:48-79 ->
// roipool3d_kernel.cu:209-237: assign_pts_to_box3d + get_pooled_idx + roipool3d_forward with a
// B*N*M int32 temporary and two cudaMalloc/cudaFree per call).  Semantics follow the reference's CPU
// code, which is the authoritative statement (roipool3d.cpp:82-195): first <= S in-box points in
// ascending index order, wrap-duplicated (slot k >= cnt copies slot k % cnt), empty boxes flagged.
//
// One workgroup per (frame, box).  Each of the 4 waves scans a contiguous quarter of the frame's points;
// a wave-wide __ballot + prefix popcount compacts its hits in index order into its own LDS list (no
// barrier inside the scan).  The four lists are concatenated in wave order == index order, truncated at
// S.  The gather then writes the box's contiguous S x (3+C) output block as one flat, fully coalesced stream
// (reads are contiguous within a point's feature row).  Nothing but the inputs and the output touches HBM.
#include "common.h"
#include "ref_trig.h"

#define RP_THREADS 256
#define RP_WAVES 4

struct BoxConst { float cx, cy, cz, hh, hw, hl, cosa, sina; };

__device__ __forceinline__ BoxConst make_box(const float* bx) {
    // roipool3d.cpp:82-95.  cy = bottom_y - h/2 in double then rounded (exactly what the reference's
    // `h / 2.0` expression does); cos/sin as the reference's host libm evaluates cos(float) / sin(float) (ref_trig.h:
    // glibc's routines restated bit for bit -- points on a box face land on the reference's side of it).
    BoxConst b;
    b.cx = bx[0]; b.cz = bx[2];
    b.cy = (float)((double)bx[1] - (double)bx[3] / 2.0);
    b.hh = bx[3] * 0.5f; b.hw = bx[4] * 0.5f; b.hl = bx[5] * 0.5f;      // exact halvings
    b.cosa = prcnn_ref_cosf(bx[6]);
    b.sina = prcnn_ref_sinf(bx[6]);
    return b;
}

// GATE: the reference's pt_in_box3d rejects any point further than max_dis = 10 m from the centre in x or z before it rotates
// (roipool3d.cpp:87-89, roipool3d_kernel.cu:19-21) -- part of roipool3d / pts_in_boxes3d semantics.  The label generator is a
// hull test on the box corners with no such limit (kitti_rcnn_dataset.py:365-394): GATE = false.
template <bool GATE = true>
__device__ __forceinline__ bool pt_in_box(const BoxConst& b, float x, float y, float z) {
    if (fabsf(y - b.cy) > b.hh) return false;
    if (GATE && (fabsf(x - b.cx) > 10.0f || fabsf(z - b.cz) > 10.0f)) return false;
    float dx = x - b.cx, dz = z - b.cz;
    float x_rot = __fadd_rn(__fmul_rn(dx, b.cosa), __fmul_rn(dz, -b.sina));
    float z_rot = __fadd_rn(__fmul_rn(dx, b.sina), __fmul_rn(dz, b.cosa));
    return (x_rot >= -b.hl) & (x_rot <= b.hl) & (z_rot >= -b.hw) & (z_rot <= b.hw);
}

// =====================================================================================================
// Point selection.  Two forms with identical results (the membership test is pt_in_box on the same coordinates either way):
//
//  * linear: each of the 4 waves scans a contiguous quarter of the frame in index order, ballot + prefix popcount compacts
//    the hits into the wave's LDS list, the lists are concatenated in wave order (== index order), truncated at S.
//  * bins (prcnn_roipool3d_bins_build first): the frame's points are bucketed once into a 64 x 64 grid over its x-z extent
//    (counting sort, entries = (x, y, z, index) rows in bin order).  A box tests only the entries of the bins its footprint's
//    bounding square touches (a row of bins is one contiguous entry span), marks hits in an N-bit LDS bitmap, and an ordered
//    sweep of the bitmap yields the first S hits in ascending index order.  At 65536 points x 512 boxes per frame the linear
//    scan is 268 M point-box tests and 3.2 GB of L2 reads per launch (stride-12 loads: a quarter of every line fetched is
//    used); the binned scan tests ~1 % of that.  Conservative range: a point in the box has |x - cx| <= hl|cos| + hw|sin|
//    up to a few ulps of the operands (the rotation is evaluated in fp32); the range is widened by 1e-3 m + relative slack, and
//    bin indices are a monotone function of the coordinate, so no in-box point lies outside the visited bins.
// =====================================================================================================
#define RPB_G 64
#define RPB_CELLS (RPB_G * RPB_G)
#define RPB_PTS 4096                          // points per workgroup of the builder passes (256 threads x 2 x 8)
#define RPB_MAX_N 262144                      // bitmap of N bits in LDS (32 KB)
#define RPB_MAX_CHUNKS (RPB_MAX_N / RPB_PTS)
// Scratch image: B frame headers, B x N entries, then B x chunks x CELLS per-chunk bin counts / write offsets.  The builder
// is a counting sort over chunks of RPB_PTS points with every atomic in LDS (device-scope atomics on one table per frame were
// tried: 2-3 x slower than a single workgroup per frame -- they execute at the memory side of the fabric).
struct RpBinsHdr { float x0, z0, inv_x, inv_z; };
struct RpBinsMeta {
    RpBinsHdr h;
    unsigned ext[RPB_MAX_CHUNKS][4];          // per chunk: max of ~enc(x), enc(x), ~enc(z), enc(z) over its finite coordinates
    int32_t start[RPB_CELLS + 4];             // exclusive offsets (+ total)
};
__host__ __device__ inline int rpb_chunks(int N) { return (N + RPB_PTS - 1) / RPB_PTS; }
__host__ __device__ inline size_t rpb_bytes(int B, int N) {
    return (size_t)B * (sizeof(RpBinsMeta) + (size_t)N * 16 + (size_t)rpb_chunks(N) * RPB_CELLS * 4);
}

__device__ __forceinline__ unsigned rpb_enc(float v) {          // order-preserving float -> uint
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float rpb_dec(unsigned e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }

__device__ __forceinline__ RpBinsHdr rpb_hdr(const unsigned (*ext)[4], int chunks) {
    unsigned e0 = 0u, e1 = 0u, e2 = 0u, e3 = 0u;
    for (int c = 0; c < chunks; c++) { e0 = max(e0, ext[c][0]); e1 = max(e1, ext[c][1]); e2 = max(e2, ext[c][2]); e3 = max(e3, ext[c][3]); }
    RpBinsHdr h;
    const bool hx = e1 != 0u, hz = e3 != 0u;                    // any finite coordinate seen
    const float xmin = hx ? rpb_dec(~e0) : 0.f, xmax = hx ? rpb_dec(e1) : 0.f;
    const float zmin = hz ? rpb_dec(~e2) : 0.f, zmax = hz ? rpb_dec(e3) : 0.f;
    h.x0 = xmin; h.z0 = zmin;
    h.inv_x = (xmax > xmin) ? (float)RPB_G / (xmax - xmin) : 0.f;
    h.inv_z = (zmax > zmin) ? (float)RPB_G / (zmax - zmin) : 0.f;
    if (!(h.inv_x < INFINITY)) h.inv_x = 0.f;
    if (!(h.inv_z < INFINITY)) h.inv_z = 0.f;
    return h;
}

__device__ __forceinline__ int rpb_cell(float v, float v0, float inv) {
    float t = (v - v0) * inv;                 // monotone in v; NaN -> 0 through fmaxf
    t = fminf(fmaxf(t, 0.f), (float)(RPB_G - 1));
    return (int)t;
}
__device__ __forceinline__ int rpb_cell2(const RpBinsHdr& h, float x, float z) {
    return rpb_cell(z, h.z0, h.inv_z) * RPB_G + rpb_cell(x, h.x0, h.inv_x);
}

// pass 1: extent of the finite coordinates of every chunk (grid: chunks x frames)
__global__ __launch_bounds__(256) void rp_bins_extent_kernel(const float* __restrict__ xyz, int N, RpBinsMeta* __restrict__ meta) {
    __shared__ unsigned red[4][4];
    const int b = blockIdx.y, tid = threadIdx.x;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    unsigned e0 = 0u, e1 = 0u, e2 = 0u, e3 = 0u;
    for (int half = 0; half < 2; half++) {
        float xs[8], zs[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = min(blockIdx.x * RPB_PTS + (half * 8 + u) * 256 + tid, N - 1);
            xs[u] = p[k * 3]; zs[u] = p[k * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (fabsf(xs[u]) < INFINITY) { const unsigned e = rpb_enc(xs[u]); e0 = max(e0, ~e); e1 = max(e1, e); }
            if (fabsf(zs[u]) < INFINITY) { const unsigned e = rpb_enc(zs[u]); e2 = max(e2, ~e); e3 = max(e3, e); }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        e0 = max(e0, (unsigned)__shfl_xor((int)e0, o)); e1 = max(e1, (unsigned)__shfl_xor((int)e1, o));
        e2 = max(e2, (unsigned)__shfl_xor((int)e2, o)); e3 = max(e3, (unsigned)__shfl_xor((int)e3, o));
    }
    if ((tid & 63) == 0) { red[tid >> 6][0] = e0; red[tid >> 6][1] = e1; red[tid >> 6][2] = e2; red[tid >> 6][3] = e3; }
    __syncthreads();
    if (tid < 4) meta[b].ext[blockIdx.x][tid] = max(max(red[0][tid], red[1][tid]), max(red[2][tid], red[3][tid]));
}

// pass 2: points per bin of every chunk
__global__ __launch_bounds__(256) void rp_bins_count_kernel(const float* __restrict__ xyz, int N, RpBinsMeta* __restrict__ meta,
                                                            int32_t* __restrict__ counts_all) {
    __shared__ int hist[RPB_CELLS];
    const int b = blockIdx.y, tid = threadIdx.x, chunks = gridDim.x;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    const RpBinsHdr h = rpb_hdr(meta[b].ext, chunks);
    if (blockIdx.x == 0 && tid == 0) meta[b].h = h;
    for (int i = tid; i < RPB_CELLS; i += 256) hist[i] = 0;
    __syncthreads();
    for (int half = 0; half < 2; half++) {
        float xs[8], zs[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = min(blockIdx.x * RPB_PTS + (half * 8 + u) * 256 + tid, N - 1);
            xs[u] = p[k * 3]; zs[u] = p[k * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (blockIdx.x * RPB_PTS + (half * 8 + u) * 256 + tid < N) atomicAdd(&hist[rpb_cell2(h, xs[u], zs[u])], 1);
    }
    __syncthreads();
    int32_t* __restrict__ out = counts_all + ((size_t)b * chunks + blockIdx.x) * RPB_CELLS;
    for (int i = tid; i < RPB_CELLS; i += 256) out[i] = hist[i];
}

// pass 3 (one workgroup per frame, 4 consecutive bins per thread): bin offsets; per-chunk counts -> per-chunk write offsets
__global__ __launch_bounds__(1024) void rp_bins_scan_kernel(RpBinsMeta* __restrict__ meta, int32_t* __restrict__ counts_all, int chunks) {
    __shared__ int wsum[16];
    RpBinsMeta& M = meta[blockIdx.x];
    int32_t* __restrict__ cnt = counts_all + (size_t)blockIdx.x * chunks * RPB_CELLS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c4[4] = {0, 0, 0, 0};
    for (int c = 0; c < chunks; c++) {
        const int4 v = *reinterpret_cast<const int4*>(cnt + (size_t)c * RPB_CELLS + tid * 4);
        c4[0] += v.x; c4[1] += v.y; c4[2] += v.z; c4[3] += v.w;
    }
    const int tsum = c4[0] + c4[1] + c4[2] + c4[3];
    int incl = tsum;
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - tsum;
    for (int w = 0; w < wave; w++) run += wsum[w];
    int4 base;
    base.x = run; base.y = run + c4[0]; base.z = base.y + c4[1]; base.w = base.z + c4[2];
    *reinterpret_cast<int4*>(M.start + tid * 4) = base;
    if (tid == 1023) M.start[RPB_CELLS] = base.w + c4[3];
    for (int c = 0; c < chunks; c++) {
        int4* q = reinterpret_cast<int4*>(cnt + (size_t)c * RPB_CELLS + tid * 4);
        const int4 v = *q;
        *q = base;
        base.x += v.x; base.y += v.y; base.z += v.z; base.w += v.w;
    }
}

// pass 4: entries (x, y, z, index) in bin order (order inside a bin is arbitrary: the pooling kernel restores index order)
__global__ __launch_bounds__(256) void rp_bins_fill_kernel(const float* __restrict__ xyz, int N, const RpBinsMeta* __restrict__ meta,
                                                           const int32_t* __restrict__ counts_all, float4* __restrict__ ent_all) {
    __shared__ int cur[RPB_CELLS];
    const int b = blockIdx.y, tid = threadIdx.x, chunks = gridDim.x;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    float4* __restrict__ ent = ent_all + (size_t)b * N;
    const RpBinsHdr h = meta[b].h;
    const int32_t* __restrict__ base = counts_all + ((size_t)b * chunks + blockIdx.x) * RPB_CELLS;
    for (int i = tid; i < RPB_CELLS; i += 256) cur[i] = base[i];
    __syncthreads();
    for (int half = 0; half < 2; half++) {
        float xs[8], ys[8], zs[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = min(blockIdx.x * RPB_PTS + (half * 8 + u) * 256 + tid, N - 1);
            xs[u] = p[k * 3]; ys[u] = p[k * 3 + 1]; zs[u] = p[k * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = blockIdx.x * RPB_PTS + (half * 8 + u) * 256 + tid;
            if (k < N) {
                const int pos = atomicAdd(&cur[rpb_cell2(h, xs[u], zs[u])], 1);
                ent[pos] = make_float4(xs[u], ys[u], zs[u], __int_as_float(k));
            }
        }
    }
}

// -> number of selected points (<= S), indices in sel[0 .. total) ascending; wave-uniform result.  lds: (RP_WAVES + 1) * S ints
__device__ __forceinline__ int rp_select_linear(const BoxConst& box, const float* __restrict__ p, int N, int S, int32_t* lds, int* wcnt,
                                                int32_t*& sel) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t* mylist = lds + wave * S;
    const int per_wave = (((N + RP_WAVES - 1) / RP_WAVES) + 63) & ~63;
    const int k_begin = wave * per_wave, k_end = min(N, k_begin + per_wave);
    int cnt = 0;                               // wave-uniform
    for (int k0 = k_begin; k0 < k_end && cnt < S; k0 += 256) {
        // 4 x 64 points per trip: the 12 loads are independent, the four ballots are consumed in index order
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int k = k0 + u * 64 + lane;
            float x = 0.f, y = 0.f, z = 0.f;
            bool ok = k < k_end;
            if (ok) { x = p[k * 3]; y = p[k * 3 + 1]; z = p[k * 3 + 2]; }
            in[u] = ok && pt_in_box(box, x, y, z);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            unsigned long long mask = __ballot(in[u]);
            int pos = cnt + __popcll(mask & ((1ULL << lane) - 1ULL));
            if (in[u] && pos < S) mylist[pos] = k0 + u * 64 + lane;
            cnt += __popcll(mask);
        }
    }
    if (lane == 0) wcnt[wave] = min(cnt, S);
    __syncthreads();
    // merge in wave order (== ascending point index), truncate at S
    sel = lds + RP_WAVES * S;
    int off = 0;
    for (int w = 0; w < wave; w++) off += wcnt[w];
    int total = 0;
    for (int w = 0; w < RP_WAVES; w++) total += wcnt[w];
    total = min(total, S);
    for (int i = lane; i < wcnt[wave]; i += 64)
        if (off + i < S) sel[off + i] = mylist[i];
    __syncthreads();
    return total;
}

// the same selection from the frame's bins.  lds: S ints (sel) + ceil(N/32) bitmap words
__device__ __forceinline__ int rp_select_bins(const BoxConst& box, const RpBinsMeta* __restrict__ fm, const float4* __restrict__ ent, int N,
                                              int S, int32_t* lds, int* wcnt, int32_t*& sel) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    sel = lds;
    uint32_t* bm = reinterpret_cast<uint32_t*>(lds + S);
    const int words = (N + 31) >> 5;
    for (int i = tid; i < words; i += RP_THREADS) bm[i] = 0u;
    const RpBinsHdr h = fm->h;
    const int32_t* __restrict__ start = fm->start;
    const float ac = fabsf(box.cosa), as = fabsf(box.sina);
    float ex = box.hl * ac + box.hw * as, ez = box.hl * as + box.hw * ac;
    ex += 1e-3f + 1e-5f * ex + 4e-7f * fabsf(box.cx);
    ez += 1e-3f + 1e-5f * ez + 4e-7f * fabsf(box.cz);
    const int cx0 = rpb_cell(box.cx - ex, h.x0, h.inv_x), cx1 = rpb_cell(box.cx + ex, h.x0, h.inv_x);
    const int cz0 = rpb_cell(box.cz - ez, h.z0, h.inv_z), cz1 = rpb_cell(box.cz + ez, h.z0, h.inv_z);
    __syncthreads();
    if (ex == ex && ez == ez) {                // a NaN box holds nothing (every comparison of pt_in_box fails)
        for (int r = cz0 + wave; r <= cz1; r += RP_WAVES) {
            const int s0 = start[r * RPB_G + cx0], s1 = start[r * RPB_G + cx1 + 1];
            for (int i = s0 + lane; i < s1; i += 64) {
                const float4 e = ent[i];
                if (pt_in_box(box, e.x, e.y, e.z)) {
                    const int k = __float_as_int(e.w);
                    atomicOr(&bm[k >> 5], 1u << (k & 31));
                }
            }
        }
    }
    __syncthreads();
    // ordered sweep: thread t owns words [t * wpt, (t + 1) * wpt)
    const int wpt = (words + RP_THREADS - 1) / RP_THREADS;
    const int w0 = tid * wpt, w1 = min(words, w0 + wpt);
    int cnt = 0;
    for (int w = w0; w < w1; w++) cnt += __popc(bm[w]);
    int incl = cnt;
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wcnt[wave] = incl;
    __syncthreads();
    int off = incl - cnt, total = 0;
    for (int w = 0; w < RP_WAVES; w++) { if (w < wave) off += wcnt[w]; total += wcnt[w]; }
    for (int w = w0; w < w1 && off < S; w++) {
        uint32_t bits = bm[w];
        while (bits && off < S) {
            const int bpos = __ffs(bits) - 1;
            sel[off++] = w * 32 + bpos;
            bits &= bits - 1;
        }
    }
    __syncthreads();
    return min(total, S);
}

template <bool BINS>
__global__ __launch_bounds__(RP_THREADS) void roipool3d_kernel(const float* __restrict__ xyz,
                                                               const float* __restrict__ boxes3d,
                                                               const float* __restrict__ feat, int N, int M, int C,
                                                               int S, float* __restrict__ pooled,
                                                               int32_t* __restrict__ empty, const char* __restrict__ bins) {
    extern __shared__ int32_t lds[];          // linear: RP_WAVES lists of S indices + the merged list; bins: S indices + the bitmap
    __shared__ BoxConst sbox;
    __shared__ int wcnt[RP_WAVES];
    const int m = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid == 0) sbox = make_box(boxes3d + ((size_t)b * M + m) * 7);
    __syncthreads();
    const BoxConst box = sbox;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    int32_t* sel;
    const RpBinsMeta* bm = reinterpret_cast<const RpBinsMeta*>(bins);
    const float4* ents = reinterpret_cast<const float4*>(bins + (size_t)gridDim.y * sizeof(RpBinsMeta));
    const int total = BINS ? rp_select_bins(box, bm + b, ents + (size_t)b * N, N, S, lds, wcnt, sel) : rp_select_linear(box, p, N, S, lds, wcnt, sel);

    const int W = 3 + C;
    float* __restrict__ o = pooled + ((size_t)b * M + m) * S * W;
    if (tid == 0) empty[(size_t)b * M + m] = total == 0 ? 1 : 0;
    if (total == 0) {
        for (size_t e = tid; e < (size_t)S * W; e += RP_THREADS) o[e] = 0.f;
        return;
    }
    // Flattened copy of the box's contiguous S x W output block: thread t handles elements t, t+256, ... of the `total`
    // DISTINCT rows; every store instruction writes 64 consecutive floats (a row of 3+C = 133 floats does not divide into
    // wave-sized pieces, so a row-per-wave mapping would leave the third pass almost empty).  (row, col) of the element is
    // tracked incrementally -- no integer division in the loop.  Eight gathers are in flight per thread before the first
    // store: the copy is bound by bytes in flight x latency, not by issue.
    // Wrap-duplication (slot k >= total copies slot k % total, roipool3d.cpp:176-191): the element is read ONCE and stored
    // to every slot it fills (the block of the distinct rows repeats with period total * W, so the replicated stores are as
    // coalesced as the first).  Re-gathering the rows per copy, as a modulo in the index list does, re-reads them from the
    // fabric: 256 resident workgroups x ~34 KB of live rows overflow the XCD's 4 MB L2 (PMC: 587 MB fetched per config-3
    // launch against 130 MB of distinct rows).
    const float* __restrict__ f = feat + (size_t)b * N * C;
    const int total_e = S * W, live_e = total * W;
    const int qstep = RP_THREADS / W, rstep = RP_THREADS - qstep * W;
    int srow = tid / W, scol = tid - srow * W;
    constexpr int U = 16;
    // (starting every workgroup at its own chunk / replica, in case blocks 1064 x 256 bytes apart camp on memory channels: no effect)
    for (int e = tid; e < live_e; e += U * RP_THREADS) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (e + u * RP_THREADS < live_e) {
                const int k = sel[srow];
                v[u] = scol < 3 ? p[k * 3 + scol] : f[(size_t)k * C + (scol - 3)];
            }
            srow += qstep; scol += rstep;
            if (scol >= W) { scol -= W; srow++; }
        }
        // replica-major: the workgroup writes its U x 256 consecutive elements of one copy of the block, then the same elements of the
        // next copy (u-major order spread every 1 KB piece over all the copies before the next piece)
        for (int rep = 0; rep < total_e; rep += live_e) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q = rep + e + u * RP_THREADS;
                if (e + u * RP_THREADS < live_e && q < total_e) o[q] = v[u];
            }
        }
    }
}

// ---- fused RCNN input builder (lib/net/rcnn_net.py:127-154) -------------------------------
// The reference concatenates [seg_mask, depth, rpn_features] into a (B,N,130) tensor, pools it into (B,M,S,133),
// subtracts the RoI centre in place and rotates each frame's RoIs in a Python loop (kitti_utils.py:45-63), after which
// rcnn_net.py slices the pooled tensor apart again ([:, :, 0:5] -> xyz_up_layer, [:, :, 5:] -> cat -> merge_down_layer).
// This kernel is the same point selection as roipool3d_kernel, but it reads the per-point channels where they already
// are, applies the canonical transform while gathering, and writes the two consumers' operands directly:
//   out_pts  rows [x', y', z', extra0, extra1] (stride ld_pts)             -> xyz_up_layer / the SA modules' xyz
//   out_feat rows of C features at any (stride, column) of a wider buffer  -> the second half of merge_down_layer's input
struct CanonParams {
    const float* xyz;        // (B, N, 3)
    const float* pool_boxes; // (B, M, 7) boxes used for the point-in-box test (already enlarged by the caller)
    const float* rois;       // (B, M, 7) boxes defining the canonical frame, or NULL: keep scene coordinates
    const float* extra[2];   // per-point scalar channels (B, N), NULL-terminated
    const float* feat;       // (B, N, C) rows, stride ld_feat
    float* out_pts;          // (B*M*S) rows, stride ld_pts, 3 + n_extra columns written
    float* out_feat;         // (B*M*S) rows, stride ld_out, C columns written from column 0 of this pointer
    int32_t* empty;          // (B, M)
    int32_t* distinct;       // (B, M) or NULL: number of distinct rows; feature rows of the wrap-copies are then not written
    int N, M, C, S, n_extra, ld_feat, ld_pts, ld_out;
    const char* bins;        // bin image of the frames (BINS kernels)
};

template <bool BINS>
__global__ __launch_bounds__(RP_THREADS) void roipool3d_canonical_kernel(CanonParams P) {
    extern __shared__ int32_t lds[];          // see roipool3d_kernel
    __shared__ BoxConst sbox;
    __shared__ float sroi[5];                 // centre x,y,z, cos(ry), sin(ry)
    __shared__ int wcnt[RP_WAVES];
    const int m = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    const int N = P.N, S = P.S, C = P.C;
    if (tid == 0) sbox = make_box(P.pool_boxes + ((size_t)b * P.M + m) * 7);
    if (tid == 64 && P.rois) {
        const float* r = P.rois + ((size_t)b * P.M + m) * 7;
        sroi[0] = r[0]; sroi[1] = r[1]; sroi[2] = r[2];
        sroi[3] = (float)cos((double)r[6]); sroi[4] = (float)sin((double)r[6]);
    }
    __syncthreads();
    const BoxConst box = sbox;
    const float* __restrict__ p = P.xyz + (size_t)b * N * 3;
    int32_t* sel;
    const RpBinsMeta* bm = reinterpret_cast<const RpBinsMeta*>(P.bins);
    const float4* ents = reinterpret_cast<const float4*>(P.bins + (size_t)gridDim.y * sizeof(RpBinsMeta));
    const int total = BINS ? rp_select_bins(box, bm + b, ents + (size_t)b * N, N, S, lds, wcnt, sel) : rp_select_linear(box, p, N, S, lds, wcnt, sel);
    if (tid == 0) {
        P.empty[(size_t)b * P.M + m] = total == 0 ? 1 : 0;
        if (P.distinct) P.distinct[(size_t)b * P.M + m] = total > 0 ? total : 1;      // an empty RoI: S equal rows
    }
    for (int s2 = total + tid; s2 < S; s2 += RP_THREADS) sel[s2] = total ? sel[s2 % total] : -1;
    __syncthreads();

    // ---- xyz (+ canonical transform) and the scalar channels: one row per thread ----
    const size_t row0 = ((size_t)b * P.M + m) * S;
    const bool canon = P.rois != nullptr;
    const float cx = sroi[0], cy = sroi[1], cz = sroi[2], ca = sroi[3], sa = sroi[4];
    for (int s = tid; s < S; s += RP_THREADS) {
        const int k = sel[s];
        float x = 0.f, y = 0.f, z = 0.f;                      // an empty RoI pools zeros (roipool3d.cpp:165-170)
        if (k >= 0) { x = p[k * 3]; y = p[k * 3 + 1]; z = p[k * 3 + 2]; }
        if (canon) {
            x = __fsub_rn(x, cx); y = __fsub_rn(y, cy); z = __fsub_rn(z, cz);          // rcnn_net.py:146-147
            const float nx = __fadd_rn(__fmul_rn(x, ca), __fmul_rn(z, -sa));             // kitti_utils.py:52-61
            const float nz = __fadd_rn(__fmul_rn(x, sa), __fmul_rn(z, ca));
            x = nx; z = nz;
        }
        float* o = P.out_pts + (row0 + s) * P.ld_pts;
        o[0] = x; o[1] = y; o[2] = z;
        for (int e = 0; e < P.n_extra; e++) o[3 + e] = k >= 0 ? P.extra[e][(size_t)b * N + k] : 0.f;
    }
    // ---- features: flat coalesced copy of S rows of C floats (row stride ld_out) ----
    if (C > 0) {
        const float* __restrict__ f = P.feat + (size_t)b * N * P.ld_feat;
        float* __restrict__ of = P.out_feat + row0 * P.ld_out;
        const int total_e = (P.distinct ? (total > 0 ? total : 1) : S) * C;
        const int qstep = RP_THREADS / C, rstep = RP_THREADS - qstep * C;
        int srow = tid / C, scol = tid - srow * C;
        constexpr int U = 8;                  // eight gathers in flight per thread before the first store
        for (int e = tid; e < total_e; e += U * RP_THREADS) {
            float v[U];
            size_t oo[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                oo[u] = (size_t)srow * P.ld_out + scol;
                if (e + u * RP_THREADS < total_e) {
                    const int k = sel[srow];
                    v[u] = k >= 0 ? f[(size_t)k * P.ld_feat + scol] : 0.f;
                }
                srow += qstep; scol += rstep;
                if (scol >= C) { scol -= C; srow++; }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (e + u * RP_THREADS < total_e) of[oo[u]] = v[u];
        }
    }
}

__global__ __launch_bounds__(256) void pts_in_boxes3d_kernel(const float* __restrict__ pts,
                                                            const float* __restrict__ boxes3d, int N, int M,
                                                            int32_t* __restrict__ flags) {
    __shared__ BoxConst sbox;
    const int m = blockIdx.y;
    if (threadIdx.x == 0) sbox = make_box(boxes3d + (size_t)m * 7);
    __syncthreads();
    const BoxConst box = sbox;
    int k = blockIdx.x * 256 + threadIdx.x;
    if (k < N) flags[(size_t)m * N + k] = pt_in_box(box, pts[k * 3], pts[k * 3 + 1], pts[k * 3 + 2]) ? 1 : 0;
}

// LDS of one pooling workgroup: linear = RP_WAVES + 1 index lists; bins = one index list + an N-bit bitmap
static size_t rp_lds_bytes(bool bins, int N, int S) {
    return bins ? ((size_t)S + (size_t)((N + 31) >> 5)) * sizeof(int32_t) : (size_t)(RP_WAVES + 1) * S * sizeof(int32_t);
}
static bool rp_use_bins(const void* work, size_t work_bytes, int B, int N) {
    return work && N <= RPB_MAX_N && work_bytes >= rpb_bytes(B, N);
}

PRCNN_API size_t prcnn_roipool3d_work_bytes(int B, int N) {
    return (B > 0 && N > 0 && N <= RPB_MAX_N) ? rpb_bytes(B, N) : 0;
}

static int rp_bins_build(const float* xyz, int B, int N, void* work, hipStream_t s) {
    RpBinsMeta* meta = (RpBinsMeta*)work;
    float4* ent = (float4*)((char*)work + (size_t)B * sizeof(RpBinsMeta));
    int32_t* counts = (int32_t*)((char*)ent + (size_t)B * N * 16);
    const int chunks = rpb_chunks(N);
    const dim3 grid(chunks, B);
    hipLaunchKernelGGL(rp_bins_extent_kernel, grid, dim3(256), 0, s, xyz, N, meta);
    hipLaunchKernelGGL(rp_bins_count_kernel, grid, dim3(256), 0, s, xyz, N, meta, counts);
    hipLaunchKernelGGL(rp_bins_scan_kernel, dim3(B), dim3(1024), 0, s, meta, counts, chunks);
    hipLaunchKernelGGL(rp_bins_fill_kernel, grid, dim3(256), 0, s, xyz, N, (const RpBinsMeta*)meta, (const int32_t*)counts, ent);
    PRCNN_LAUNCH_CHECK("prcnn_roipool3d: bins");
    return PRCNN_OK;
}

PRCNN_API int prcnn_roipool3d_ws(const float* xyz, const float* boxes3d, const float* feat, int B, int N, int M, int C, int S, float* pooled,
                                 int32_t* empty, void* work, size_t work_bytes, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && N > 0 && M >= 0 && C >= 0 && S > 0, "prcnn_roipool3d: bad shape B=%d N=%d M=%d C=%d S=%d", B, N, M, C, S);
    if (B == 0 || M == 0) return PRCNN_OK;
    PRCNN_REQUIRE(xyz && boxes3d && pooled && empty && (C == 0 || feat), "prcnn_roipool3d: null pointer");
    const bool bins = rp_use_bins(work, work_bytes, B, N);
    const size_t lds_bytes = rp_lds_bytes(bins, N, S);
    PRCNN_REQUIRE(lds_bytes <= 60 * 1024, "prcnn_roipool3d: sampled_pt_num %d too large for the LDS index lists", S);
    hipStream_t s = (hipStream_t)stream;
    if (bins) {
        if (int rc = rp_bins_build(xyz, B, N, work, s)) return rc;
        hipLaunchKernelGGL(roipool3d_kernel<true>, dim3(M, B), dim3(RP_THREADS), lds_bytes, s, xyz, boxes3d, feat, N, M, C, S, pooled, empty,
                           (const char*)work);
    } else {
        hipLaunchKernelGGL(roipool3d_kernel<false>, dim3(M, B), dim3(RP_THREADS), lds_bytes, s, xyz, boxes3d, feat, N, M, C, S, pooled, empty,
                           (const char*)nullptr);
    }
    PRCNN_LAUNCH_CHECK("prcnn_roipool3d");
    return PRCNN_OK;
}

PRCNN_API int prcnn_roipool3d(const float* xyz, const float* boxes3d, const float* feat, int B, int N, int M, int C,
                              int S, float* pooled, int32_t* empty, prcnn_stream_t stream) {
    return prcnn_roipool3d_ws(xyz, boxes3d, feat, B, N, M, C, S, pooled, empty, nullptr, 0, stream);
}

PRCNN_API int prcnn_pts_in_boxes3d(const float* pts, const float* boxes3d, int N, int M, int32_t* flags,
                                   prcnn_stream_t stream) {
    PRCNN_REQUIRE(N >= 0 && M >= 0, "prcnn_pts_in_boxes3d: bad shape");
    if (N == 0 || M == 0) return PRCNN_OK;
    PRCNN_REQUIRE(pts && boxes3d && flags, "prcnn_pts_in_boxes3d: null pointer");
    hipLaunchKernelGGL(pts_in_boxes3d_kernel, dim3(prcnn_divup(N, 256), M), dim3(256), 0, (hipStream_t)stream, pts,
                       boxes3d, N, M, flags);
    PRCNN_LAUNCH_CHECK("prcnn_pts_in_boxes3d");
    return PRCNN_OK;
}

PRCNN_API int prcnn_roipool3d_canonical_ws(const float* xyz, const float* pool_boxes3d, const float* rois, const float* extra0,
                                           const float* extra1, const float* feat, int ld_feat, int B, int N, int M, int C, int S,
                                           float* out_pts, int ld_pts, float* out_feat, int ld_out, int32_t* empty, int32_t* distinct,
                                           void* work, size_t work_bytes, prcnn_stream_t stream) {
    const char* op = "prcnn_roipool3d_canonical";
    PRCNN_REQUIRE(B >= 0 && N > 0 && M >= 0 && C >= 0 && S > 0, "%s: bad shape B=%d N=%d M=%d C=%d S=%d", op, B, N, M, C, S);
    if (B == 0 || M == 0) return PRCNN_OK;
    const int n_extra = extra0 ? (extra1 ? 2 : 1) : 0;
    PRCNN_REQUIRE(extra0 || !extra1, "%s: extra1 without extra0", op);
    PRCNN_REQUIRE(xyz && pool_boxes3d && out_pts && empty && (C == 0 || (feat && out_feat)), "%s: null pointer", op);
    PRCNN_REQUIRE(ld_pts >= 3 + n_extra && (C == 0 || (ld_feat >= C && ld_out >= C)), "%s: row strides smaller than the rows", op);
    const bool bins = rp_use_bins(work, work_bytes, B, N);
    const size_t lds_bytes = rp_lds_bytes(bins, N, S);
    PRCNN_REQUIRE(lds_bytes <= 60 * 1024, "%s: sampled_pt_num %d too large for the LDS index lists", op, S);
    CanonParams P;
    P.xyz = xyz; P.pool_boxes = pool_boxes3d; P.rois = rois; P.extra[0] = extra0; P.extra[1] = extra1; P.feat = feat;
    P.out_pts = out_pts; P.out_feat = out_feat; P.empty = empty; P.distinct = distinct;
    P.N = N; P.M = M; P.C = C; P.S = S; P.n_extra = n_extra; P.ld_feat = ld_feat; P.ld_pts = ld_pts; P.ld_out = ld_out;
    P.bins = (const char*)work;
    hipStream_t s = (hipStream_t)stream;
    if (bins) {
        if (int rc = rp_bins_build(xyz, B, N, work, s)) return rc;
        hipLaunchKernelGGL(roipool3d_canonical_kernel<true>, dim3(M, B), dim3(RP_THREADS), lds_bytes, s, P);
    } else {
        P.bins = nullptr;
        hipLaunchKernelGGL(roipool3d_canonical_kernel<false>, dim3(M, B), dim3(RP_THREADS), lds_bytes, s, P);
    }
    PRCNN_LAUNCH_CHECK(op);
    return PRCNN_OK;
}

PRCNN_API int prcnn_roipool3d_canonical(const float* xyz, const float* pool_boxes3d, const float* rois, const float* extra0,
                                        const float* extra1, const float* feat, int ld_feat, int B, int N, int M, int C, int S,
                                        float* out_pts, int ld_pts, float* out_feat, int ld_out, int32_t* empty, int32_t* distinct,
                                        prcnn_stream_t stream) {
    return prcnn_roipool3d_canonical_ws(xyz, pool_boxes3d, rois, extra0, extra1, feat, ld_feat, B, N, M, C, S, out_pts, ld_pts, out_feat,
                                        ld_out, empty, distinct, nullptr, 0, stream);
}

// =====================================================================================================
// RPN training labels on the device (second half).  Replaces the per-sample numpy / scipy code of
// KittiRCNNDataset.generate_rpn_training_labels (lib/datasets/kitti_rcnn_dataset.py:365-394): per GT box a Delaunay
// hull test of all 16384 points against the box and against the box enlarged by 0.2 m (kitti_utils.enlarge_box3d),
// foreground = 1, "in the enlarged box only" = -1 (ignored), regression target = (centre - point, h, w, l, ry).
// Here: one thread per point walks the frame's <= LABEL_MAX_GT boxes (constants in LDS) IN BOX ORDER, so that a point in
// several boxes ends up with exactly what the reference's sequential overwrites leave.  The in-box test is the
// analytic one of roipool3d (pt_in_box above, WITHOUT its 10 m centre-distance gate: the hull test has none, so a GT box
// with a half extent over 10 m keeps all its points) instead of the hull test: identical except for points within rounding
// distance of a face.
// =====================================================================================================
#define LABEL_MAX_GT 128
__global__ __launch_bounds__(256) void rpn_labels_kernel(const float* __restrict__ pts, const float* __restrict__ gt_boxes3d,
                                                         const int32_t* __restrict__ num_gt, int N, int G, float extra,
                                                         int32_t* __restrict__ cls_label, float* __restrict__ reg_label) {
    __shared__ BoxConst sbox[LABEL_MAX_GT], sbig[LABEL_MAX_GT];
    __shared__ float sraw[LABEL_MAX_GT][7];
    const int b = blockIdx.y;
    const int g = num_gt ? min(max(num_gt[b], 0), G) : G;
    for (int k = threadIdx.x; k < g; k += 256) {
        const float* bx = gt_boxes3d + ((size_t)b * G + k) * 7;
        float big[7];
#pragma unroll
        for (int c = 0; c < 7; c++) { sraw[k][c] = bx[c]; big[c] = bx[c]; }
        big[3] = bx[3] + extra * 2; big[4] = bx[4] + extra * 2; big[5] = bx[5] + extra * 2; big[1] = bx[1] + extra;   // kitti_utils.py:150-160
        sbox[k] = make_box(bx);
        sbig[k] = make_box(big);
    }
    __syncthreads();
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float* p = pts + ((size_t)b * N + n) * 3;
    const float x = p[0], y = p[1], z = p[2];
    int cls = 0;
    float r[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < g; k++) {
        const bool fg = pt_in_box<false>(sbox[k], x, y, z), en = pt_in_box<false>(sbig[k], x, y, z);
        if (fg) {
            cls = 1;
            const float cy = sraw[k][1] - sraw[k][3] / 2;          // centre y (float32 arithmetic, as numpy does it)
            r[0] = sraw[k][0] - x; r[1] = cy - y; r[2] = sraw[k][2] - z;
            r[3] = sraw[k][3]; r[4] = sraw[k][4]; r[5] = sraw[k][5]; r[6] = sraw[k][6];
        }
        if (fg != en) cls = -1;
    }
    cls_label[(size_t)b * N + n] = cls;
    float* o = reg_label + ((size_t)b * N + n) * 7;
#pragma unroll
    for (int c = 0; c < 7; c++) o[c] = r[c];
}

PRCNN_API int prcnn_rpn_labels(const float* pts, const float* gt_boxes3d, const int32_t* num_gt, int B, int N, int G, float extra_width,
                               int32_t* cls_label, float* reg_label, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && N >= 0 && G >= 0, "prcnn_rpn_labels: bad shape B=%d N=%d G=%d", B, N, G);
    PRCNN_REQUIRE(G <= LABEL_MAX_GT, "prcnn_rpn_labels: at most %d GT boxes per frame (got %d)", LABEL_MAX_GT, G);
    if (B == 0 || N == 0) return PRCNN_OK;
    PRCNN_REQUIRE(pts && cls_label && reg_label && (G == 0 || gt_boxes3d), "prcnn_rpn_labels: null pointer");
    hipLaunchKernelGGL(rpn_labels_kernel, dim3(prcnn_divup(N, 256), B), dim3(256), 0, (hipStream_t)stream, pts, gt_boxes3d, num_gt,
                       N, G, extra_width, cls_label, reg_label);
    PRCNN_LAUNCH_CHECK("prcnn_rpn_labels");
    return PRCNN_OK;
}


// =====================================================================================================
// GT-augmentation scene edit (second half).  KittiRCNNDataset.apply_gt_aug_to_one_scene
// (lib/datasets/kitti_rcnn_dataset.py:408-511) pastes database objects into a scene: for every ACCEPTED object it calls
// pts_in_boxes3d_cpu on the whole scene against the object's box with h += 2 (:484-489), clears those points' keep flag, and
// after the loop keeps `pts_rect[src_pts_flag == 1]` and concatenates the pasted objects' points (:501-507) -- one full-scene
// scan per object plus two boolean-mask copies on the host.  Here the whole edit is one launch for a batch of scenes: a
// workgroup per scene holds the accepted boxes' constants in LDS, tests each point against all of them once, compacts the
// survivors IN INDEX ORDER (ballot + prefix popcount per wave, the 16 wave counts through a double-buffered LDS row: one
// barrier per 1024 points) and appends the new points; the tail of the (N + P)-row output is zero-filled.
// The sampling loop around it (database lookup, road plane, the shapely collision test) stays host code and out of scope.
// In-box test: pt_in_box above (== prcnn_pts_in_boxes3d).
// =====================================================================================================
#define AUG_MAX_BOXES 64
#define AUG_THREADS 1024
__global__ __launch_bounds__(AUG_THREADS) void gt_aug_edit_kernel(
    const float* __restrict__ pts, const float* __restrict__ inten, const int32_t* __restrict__ num_pts,
    const float* __restrict__ boxes, const int32_t* __restrict__ num_boxes, float extra_h, const float* __restrict__ new_pts,
    const float* __restrict__ new_inten, const int32_t* __restrict__ num_new, int N, int K, int P, float* __restrict__ out_pts,
    float* __restrict__ out_inten, int32_t* __restrict__ out_count, int32_t* __restrict__ removed) {
    __shared__ BoxConst sbox[AUG_MAX_BOXES];
    __shared__ int wsum[2][AUG_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = num_pts ? min(max(num_pts[b], 0), N) : N;
    const int k = num_boxes ? min(max(num_boxes[b], 0), K) : K;
    const int np = num_new ? min(max(num_new[b], 0), P) : P;
    for (int i = tid; i < k; i += AUG_THREADS) {
        const float* bx = boxes + ((size_t)b * K + i) * 7;
        float big[7];
#pragma unroll
        for (int c = 0; c < 7; c++) big[c] = bx[c];
        big[3] = bx[3] + extra_h;                              // kitti_rcnn_dataset.py:484-485 (float32 add, as numpy does it)
        sbox[i] = make_box(big);
    }
    __syncthreads();
    const float* __restrict__ p = pts + (size_t)b * N * 3;
    const float* __restrict__ it = inten ? inten + (size_t)b * N : nullptr;
    float* __restrict__ op = out_pts + (size_t)b * (N + P) * 3;
    float* __restrict__ oi = out_inten ? out_inten + (size_t)b * (N + P) : nullptr;
    int kept = 0;                                              // workgroup-uniform running count
    int trip = 0;
    // the next trip's point is requested before this trip's test / barrier (the loop is one load round trip per trip otherwise)
    float nx = 0.f, ny = 0.f, nz = 0.f, nw = 0.f;
    if (tid < n) { nx = p[tid * 3]; ny = p[tid * 3 + 1]; nz = p[tid * 3 + 2]; if (it) nw = it[tid]; }
    for (int base = 0; base < n; base += AUG_THREADS, trip ^= 1) {
        const int i = base + tid;
        const float x = nx, y = ny, z = nz, w = nw;
        const int i2 = i + AUG_THREADS;
        if (i2 < n) { nx = p[i2 * 3]; ny = p[i2 * 3 + 1]; nz = p[i2 * 3 + 2]; if (it) nw = it[i2]; }
        bool keep = i < n;
        if (keep) {
            bool inside = false;
            for (int q = 0; q < k; q++) inside |= pt_in_box(sbox[q], x, y, z);
            keep = !inside;
            if (removed) removed[(size_t)b * N + i] = inside ? 1 : 0;
        }
        const unsigned long long mask = __ballot(keep);
        if (lane == 0) wsum[trip][wave] = __popcll(mask);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int v = 0; v < AUG_THREADS / 64; v++) {
            const int c = wsum[trip][v];
            off += v < wave ? c : 0;
            total += c;
        }
        if (keep) {
            const int pos = kept + off + __popcll(mask & ((1ULL << lane) - 1ULL));
            op[pos * 3] = x; op[pos * 3 + 1] = y; op[pos * 3 + 2] = z;
            if (oi) oi[pos] = w;
        }
        kept += total;
    }
    const float* __restrict__ q3 = new_pts + (size_t)b * P * 3;
    for (int i = tid; i < np * 3; i += AUG_THREADS) op[kept * 3 + i] = q3[i];
    if (oi) {
        const float* __restrict__ qi = new_inten + (size_t)b * P;
        for (int i = tid; i < np; i += AUG_THREADS) oi[kept + i] = qi[i];
    }
    const int cnt = kept + np;
    for (int i = cnt * 3 + tid; i < (N + P) * 3; i += AUG_THREADS) op[i] = 0.f;
    if (oi)
        for (int i = cnt + tid; i < N + P; i += AUG_THREADS) oi[i] = 0.f;
    if (removed)
        for (int i = n + tid; i < N; i += AUG_THREADS) removed[(size_t)b * N + i] = 0;
    if (tid == 0) out_count[b] = cnt;
}

PRCNN_API int prcnn_gt_aug_edit(const float* pts, const float* intensity, const int32_t* num_pts, const float* boxes3d,
                                const int32_t* num_boxes, float extra_h, const float* new_pts, const float* new_intensity,
                                const int32_t* num_new, int B, int N, int K, int P, float* out_pts, float* out_intensity,
                                int32_t* out_count, int32_t* removed, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && N >= 0 && K >= 0 && P >= 0, "prcnn_gt_aug_edit: bad shape");
    PRCNN_REQUIRE(K <= AUG_MAX_BOXES, "prcnn_gt_aug_edit: more than 64 boxes per scene");
    PRCNN_REQUIRE((long)N + P < (1L << 29), "prcnn_gt_aug_edit: scene too large");
    if (B == 0) return PRCNN_OK;
    PRCNN_REQUIRE(out_count && (N + P == 0 || out_pts), "prcnn_gt_aug_edit: null output");
    PRCNN_REQUIRE((N == 0 || pts) && (K == 0 || boxes3d) && (P == 0 || new_pts), "prcnn_gt_aug_edit: null input");
    PRCNN_REQUIRE(!out_intensity || ((N == 0 || intensity) && (P == 0 || new_intensity)),
                  "prcnn_gt_aug_edit: out_intensity needs intensity and new_intensity");
    hipLaunchKernelGGL(gt_aug_edit_kernel, dim3(B), dim3(AUG_THREADS), 0, (hipStream_t)stream, pts, out_intensity ? intensity : nullptr,
                       num_pts, boxes3d, num_boxes, extra_h, new_pts, new_intensity, num_new, N, K, P, out_pts, out_intensity, out_count,
                       removed);
    PRCNN_LAUNCH_CHECK("prcnn_gt_aug_edit");
    return PRCNN_OK;
}
