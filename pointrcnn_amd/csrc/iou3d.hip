// iou3d.hip -- rotated BEV overlap / IoU matrices and greedy NMS (rotated + axis-aligned) for gfx950.
//
// Replaces iou3d_cuda.{boxes_overlap_bev_gpu, boxes_iou_bev_gpu, nms_gpu, nms_normal_gpu}
// (lib/utils/iou3d/src/iou3d.cpp:31,52,73,123 -> iou3d_kernel.cu:223-387).  The geometry follows the
// reference's clipping algorithm step for step (iou3d_kernel.cu:34-221: edge-edge intersections, contained
// corners, centroid, angular sort, shoelace) in the reference's own host arithmetic -- glibc's float sinf / cosf / atan2f restated
// bit for bit (ref_trig.h; oracle/prcnn_oracle.c trig_mode 2) -- so results are bit-identical to the oracle AND to the reference's
// sources compiled for the host (the test suite's reference build).
//
// What is different from the reference's execution plan:
//   * per-box work (cos/sin, centre, rotated corners) is done ONCE per box into LDS, not once per pair;
//     the angular sort key is computed once per vertex, not 2x per bubble-sort comparison
//     (the reference evaluates atan2 cnt*(cnt-1) times per pair: iou3d_kernel.cu:104-106,188-196);
//   * the NMS mask kernel only evaluates upper-triangle 64x64 tiles (the reference computes all:
//     iou3d_kernel.cu:258), and the greedy sweep runs ON THE DEVICE (one wave per problem, suppression
//     bitmap in LDS) instead of a synchronous D2H copy of the N x N/64 mask plus a host loop
//     (iou3d.cpp:86-116): no host synchronisation anywhere.
#include "iou3d_geom.h"

// ---- pairwise matrices: 16x16 pair tile per workgroup, boxes prepared once in LDS ---------------
template <bool IOU>
__global__ __launch_bounds__(256) void pair_matrix_kernel(const float* __restrict__ boxes_a, int na,
                                                          const float* __restrict__ boxes_b, int nb,
                                                          float* __restrict__ out) {
    __shared__ RBox sa[16], sb[16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int a0 = blockIdx.y * 16, b0 = blockIdx.x * 16;
    if (threadIdx.x < 16) { if (a0 + threadIdx.x < na) make_rbox(boxes_a + (size_t)(a0 + threadIdx.x) * 5, sa[threadIdx.x]); }
    else if (threadIdx.x < 32) { int t = threadIdx.x - 16; if (b0 + t < nb) make_rbox(boxes_b + (size_t)(b0 + t) * 5, sb[t]); }
    __syncthreads();
    const int ai = a0 + ty, bi = b0 + tx;
    if (ai >= na || bi >= nb) return;
    out[(size_t)ai * nb + bi] = IOU ? iou_bev(sa[ty], sb[tx]) : box_overlap(sa[ty], sb[tx]);
}

// ---- NMS: upper-triangle suppression mask (64x64 tiles) -----------------------------------------
template <int KIND>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int N, float thresh, int W,
                                                      unsigned long long* __restrict__ mask, unsigned char* __restrict__ tflag) {
    const int row_blk = blockIdx.y, col_blk = blockIdx.x;
    if (col_blk < row_blk) return;                       // never read by the sweep
    const int t = threadIdx.x;
    const int row = row_blk * 64 + t, col = col_blk * 64 + t;
    const int col_size = min(64, N - col_blk * 64);
    unsigned long long bits = 0;
    if (KIND == PRCNN_NMS_ROTATED) {
        // Round 5: the tile's pairs that need the polygon clip are LISTED first (circumscribed circles not far apart: a dozen of the
        // 4096 on scattered boxes) and then clipped one per lane.  Rounds 1-4 walked the 64 columns with every row's lane deciding for
        // itself: whenever one lane of the wave needed a clip the other 63 waited for it -- ~32 one-lane clips per tile.  The listed
        // pairs are the same pairs, the tests the same tests: same mask bits.
        __shared__ RBox srow[64], scol[64];
        __shared__ unsigned short plist[64 * 64];
        __shared__ unsigned rbits[64][2];
        if (row < N) make_rbox(boxes + (size_t)row * 5, srow[t]);
        if (col < N) make_rbox(boxes + (size_t)col * 5, scol[t]);
        rbits[t][0] = 0u; rbits[t][1] = 0u;
        __syncthreads();
        const bool skip_far = thresh >= 0.0f;            // a zero overlap suppresses only under a negative threshold
        const int start = (row_blk == col_blk) ? t + 1 : 0;    // iou3d_kernel.cu:281-283
        int np = 0;                                      // uniform: one wave per workgroup
        for (int i = 0; i < col_size; i++) {
            const bool pass = row < N && i >= start && !(skip_far && decided_without_clip(srow[t], scol[i], thresh));
            const unsigned long long bm = __ballot(pass);
            if (pass) plist[np + (int)__popcll(bm & ((1ULL << t) - 1ULL))] = (unsigned short)((t << 6) | i);
            np += (int)__popcll(bm);
        }
        __syncthreads();
        for (int e = t; e < np; e += 64) {
            const int code = plist[e], r = code >> 6, i = code & 63;
            if (iou_bev(srow[r], scol[i]) > thresh) atomicOr(&rbits[r][i >> 5], 1u << (i & 31));
        }
        __syncthreads();
        bits = ((unsigned long long)rbits[t][1] << 32) | rbits[t][0];
    } else {
        __shared__ float scolb[64 * 5];
        if (col < N)
            for (int c = 0; c < 5; c++) scolb[t * 5 + c] = boxes[(size_t)col * 5 + c];
        __syncthreads();
        if (row < N) {
            float rb[5];
            for (int c = 0; c < 5; c++) rb[c] = boxes[(size_t)row * 5 + c];
            int start = (row_blk == col_blk) ? t + 1 : 0;
            for (int i = start; i < col_size; i++)
                if (iou_normal(rb, scolb + i * 5) > thresh) bits |= 1ULL << i;
        }
    }
    if (row < N) mask[(size_t)row * W + col_blk] = bits;
    // one byte per tile: does any row of it suppress anything?  On scattered boxes nine tiles in ten are empty and the sweep's folders
    // skip their rows (the fold of a block is bound by what one CU can pull from L2: 64 rows x (W - i) words per block)
    const bool some = __ballot(row < N && bits != 0ULL) != 0ULL;
    if (t == 0) tflag[(size_t)row_blk * W + col_blk] = some ? 1 : 0;
}

// ---- NMS: greedy sweep on the device (iou3d.cpp:100-119), one workgroup per problem ------------------
// The sweep is a serial chain over the 64-box blocks (N = 6300: 99 of them) and a lone caller -- the reference's proposal layer
// issues one NMS at a time and waits for each (lib/rpn/proposal_layer.py:100-105) -- sees its LATENCY: rounds 1-5 took ~4 us per
// block (every thread issued its 64 row loads, wave 0 walked all 64 rows of the block, every thread folded 64 selects, two barriers;
// 0.4-0.5 ms per call, half of the GPU time of the reference's unchanged evaluation loop).  Round 6 splits the roles:
//   * wave 0, the RESOLVER, runs the chain alone.  Only rows whose diagonal word is non-zero can change the block's outcome, and a
//     row's suppression state is final once every lower row is done (a row only suppresses higher ones), so the block resolves by
//     walking the set bits of (non-zero rows & ~suppressed) -- a handful on scattered boxes, at most 64 -- and kept = ~cur.  The
//     contribution of block i to the NEXT block's word is folded by the resolver itself (one word per lane, OR-reduced on the DPP
//     network), so it never waits for the folders' result of the block it has just resolved;
//   * waves 1-8, the FOLDERS, run one block behind: during iteration i they OR the kept rows of block i-1 into the suppression words
//     >= i+1 (16 rows x one word per thread, rows requested an iteration ahead so no load sits on anybody's path, LDS ds_or).
// One barrier per block, ~0.3 us per block on scattered boxes.  Same greedy order, same keep list; max_keep > 0 ends the sweep as soon
// as enough boxes are kept.
#define SWEEP_THREADS 576                                // resolver wave + 8 folder waves
#define SWEEP_FOLD_ROWS 16

// wave-wide OR of a 64-bit word per lane (DPP, the two halves interleaved so that one fills the other's wait states) -> uniform
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long x) {
    int a = (int)(unsigned)x, b = (int)(unsigned)(x >> 32);
#define SWEEP_OR_STEP(ctrl, rmask) { const int ta = PRCNN_DPP(a, ctrl, rmask), tb = PRCNN_DPP(b, ctrl, rmask); a |= ta; b |= tb; }
    SWEEP_OR_STEP(0x111, 0xf) SWEEP_OR_STEP(0x112, 0xf) SWEEP_OR_STEP(0x114, 0xf) SWEEP_OR_STEP(0x118, 0xf)
    SWEEP_OR_STEP(0x142, 0xa) SWEEP_OR_STEP(0x143, 0xc)
#undef SWEEP_OR_STEP
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(b, 63) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readlane(a, 63);
}
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    // readfirstlane returns a SIGNED int: go through unsigned or the low word sign-extends into the high one
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}

// -DSWEEP_TIMING (tools/build_variant.py + tools/nms_timing.py): cycle counters of the resolver's and of folder wave 1's phases, written
// over the head of the mask after the last block; never defined in the product build
#ifdef SWEEP_TIMING
#define SWEEP_T(k) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; }
#else
#define SWEEP_T(k)
#endif

#define SWEEP_DEPTH 4                                    // blocks between the request of a block's words and their use
#define SWEEP_FLAGS_LDS_MAX_W 128                        // tile flags of up to 128 x 128 tiles (N <= 8 192) are staged in LDS

struct SweepShared {
    unsigned long long* remv;                            // W words: bit set = suppressed
    unsigned long long* kslot;                           // [i & 1] = kept bits of block i
    unsigned long long* stop;                            // [i & 1] != 0: enough boxes kept after block i
    const unsigned char* flags;                          // W x W tile flags: in LDS when they fit, else the mask kernel's array
};

// The mask was written by workgroups on all eight XCDs and the sweep reads it on one: every word comes from the memory side
// (~1 us), not from this XCD's L2.  Whatever the chain needs is therefore requested SWEEP_DEPTH blocks ahead into a ring of register
// sets that the unrolled loop walks without ever copying a value in flight (a copy would wait for the load).

// Resolver, block i.  d / nr: word i / word i+1 of the block's rows (one row per lane), requested SWEEP_DEPTH blocks ago; they leave
// holding the requests for block i+SWEEP_DEPTH (issued after their last use, so they stay in the same registers).
// -> true when enough boxes are kept.
__device__ __forceinline__ bool sweep_resolve(const unsigned long long* __restrict__ mask, int N, int W, int max_keep, int i, int lane,
                                              const SweepShared& sh, unsigned long long& d, unsigned long long& nr,
                                              unsigned long long& carry, int& num, int64_t* __restrict__ keep
#ifdef SWEEP_TIMING
                                              , unsigned long long (&tacc)[8], unsigned long long& tlast
#endif
                                              ) {
    const int nrows = min(64, N - i * 64);
    unsigned long long cur = uniform_u64(sh.remv[i]) | carry;
    SWEEP_T(0)
    const unsigned dlo = (unsigned)d, dhi = (unsigned)(d >> 32);
    unsigned long long todo = __ballot(d != 0ULL) & ~cur;
    while (todo) {                                       // ascending over the rows that can still suppress something
        const int t = __builtin_ctzll(todo);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)dlo, t);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)dhi, t);
        cur |= ((unsigned long long)hi << 32) | lo;
        todo &= todo - 1ULL;
        todo &= ~cur;
    }
    SWEEP_T(1)
    const int rn = min((i + SWEEP_DEPTH) * 64 + lane, N - 1);      // clamped: rows / words past the end are read again and never used
    d = mask[(size_t)rn * W + min(i + SWEEP_DEPTH, W - 1)];
    const unsigned long long valid = nrows == 64 ? ~0ULL : ((1ULL << nrows) - 1ULL);
    const unsigned long long kept = ~cur & valid;
    const bool mine = (kept >> lane) & 1ULL;
    if (mine) keep[num + __popcll(kept & ((1ULL << lane) - 1ULL))] = i * 64 + lane;
    num += __popcll(kept);
    const bool enough = max_keep > 0 && num >= max_keep;
    if (lane == 0) { sh.kslot[i & 1] = kept; sh.stop[i & 1] = enough ? 1ULL : 0ULL; }
    SWEEP_T(2)
    __syncthreads();
    SWEEP_T(3)
    carry = i + 1 < W ? wave_or_u64(mine ? nr : 0ULL) : 0ULL;
    nr = mask[(size_t)rn * W + min(i + SWEEP_DEPTH + 1, W - 1)];
    SWEEP_T(4)
    return enough;
}

// Folder, iteration i: ORs the kept rows [c*16, c*16+16) of block i-1 (`part`, requested SWEEP_DEPTH iterations ago when the tile's flag
// `has` said it holds anything) into the suppression words i+1 + g*64 + lane (+128, ...), then requests the same rows of block
// i-1+SWEEP_DEPTH into `part`.  -> true when the resolver said stop.
__device__ __forceinline__ bool sweep_tile_flag(const SweepShared& sh, int W, int blk, int w) {
    return blk < W && w < W && sh.flags[(size_t)min(blk, W - 1) * W + min(w, W - 1)] != 0;          // clamped: no guarded load
}
__device__ __forceinline__ void sweep_request(const unsigned long long* __restrict__ mask, int N, int W, int blk, int lane, int c, int g,
                                              const SweepShared& sh, unsigned long long (&part)[SWEEP_FOLD_ROWS], bool& has) {
    // rows [c*16, c*16+16) of block `blk`, word blk+2 + g*64 + lane (the first word the folders own for that block); rows past the last
    // one are clamped (read again, never selected: kept has no bit there)
    const int w = blk + 2 + g * 64 + lane;
    has = sweep_tile_flag(sh, W, blk, w);
    if (has) {
        // (the mask is allocated in whole blocks of 64 rows: rows past the last box are never written and never selected -- kept has no
        //  bit there -- so a row address is the first row's plus r row strides, no clamp: 16 x 7 scalar instructions less per request)
        const char* rowb = reinterpret_cast<const char*>(mask + (size_t)(blk * 64 + c * SWEEP_FOLD_ROWS) * W);   // uniform
        const size_t stride = (size_t)W * 8;
        const unsigned voff = (unsigned)w * 8u;          // uniform base + 32-bit lane offset: the saddr form of the load, one scalar add per row
#pragma unroll
        for (int r = 0; r < SWEEP_FOLD_ROWS; r++) {
            part[r] = *reinterpret_cast<const unsigned long long*>(rowb + voff);
            rowb += stride;
        }
    }
}
__device__ __forceinline__ bool sweep_fold(const unsigned long long* __restrict__ mask, int N, int W,
                                           int i, int lane, int c, int g, const SweepShared& sh,
                                           unsigned long long (&part)[SWEEP_FOLD_ROWS], bool& has
#ifdef SWEEP_TIMING
                                           , unsigned long long (&tacc)[8], unsigned long long& tlast
#endif
                                           ) {
    if (i >= 1) {
        const unsigned kb = (unsigned)(uniform_u64(sh.kslot[(i - 1) & 1]) >> (c * SWEEP_FOLD_ROWS)) & 0xffffu;
        if (kb) {
            const int w0 = i + 1 + g * 64 + lane;
            if (has) {                                   // (has implies w0 < W)
                unsigned lo = 0u, hi = 0u;               // kept rows folded with uniform masks: (row & mask) | acc is one v_and_or_b32 per half
#pragma unroll
                for (int r = 0; r < SWEEP_FOLD_ROWS; r++) {
                    const unsigned mk = 0u - ((kb >> r) & 1u);
                    lo |= (unsigned)part[r] & mk;
                    hi |= (unsigned)(part[r] >> 32) & mk;
                }
                const unsigned long long acc = ((unsigned long long)hi << 32) | lo;
                if (acc) atomicOr(&sh.remv[w0], acc);
            }
            for (int w = w0 + 128; w < W; w += 128) {          // more than 129 blocks (N > 8 256): the far words, not prefetched
                if (!sh.flags[(size_t)(i - 1) * W + w]) continue;
                unsigned long long acc = 0ULL;
                const unsigned long long* col = mask + (size_t)((i - 1) * 64 + c * SWEEP_FOLD_ROWS) * W + w;
#pragma unroll
                for (int r = 0; r < SWEEP_FOLD_ROWS; r++)
                    if ((kb >> r) & 1u) acc |= col[(size_t)r * W];
                if (acc) atomicOr(&sh.remv[w], acc);
            }
        }
    }
    SWEEP_T(0)
    sweep_request(mask, N, W, i - 1 + SWEEP_DEPTH, lane, c, g, sh, part, has);
    SWEEP_T(1)
    __syncthreads();
    SWEEP_T(2)
    const bool stop_now = sh.stop[i & 1] != 0ULL;
    SWEEP_T(3)
    return stop_now;
}

__global__ __launch_bounds__(SWEEP_THREADS) void nms_sweep_kernel(const unsigned long long* __restrict__ mask,
                                                                  const unsigned char* __restrict__ tflag, int N, int W,
                                                                  int max_keep, int64_t* __restrict__ keep,
                                                                  int32_t* __restrict__ num_keep) {
    extern __shared__ unsigned long long sweep_lds[];    // W suppression words, the exchange words, then the tile flags when they fit
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // in an SGPR: the folders' row addresses are scalar arithmetic
    const bool flags_in_lds = W <= SWEEP_FLAGS_LDS_MAX_W;
    const SweepShared sh = {sweep_lds, sweep_lds + W, sweep_lds + W + 2,
                            flags_in_lds ? (const unsigned char*)(sweep_lds + W + 4) : tflag};
    for (int w = tid; w < W + 4; w += SWEEP_THREADS) sweep_lds[w] = 0ULL;
    if (flags_in_lds) {
        // (the flags of the tiles below the diagonal are never written and never used: whatever is there is copied and ignored)
        unsigned char* fl = (unsigned char*)(sweep_lds + W + 4);
        for (int e = tid; e < W * W; e += SWEEP_THREADS) fl[e] = tflag[e];
    }
    __syncthreads();
#ifdef SWEEP_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define SWEEP_TARGS , tacc, tlast
#else
#define SWEEP_TARGS
#endif
    if (wave == 0) {
        unsigned long long d0, d1, d2, d3, n0, n1, n2, n3;
#define SWEEP_RQ(k, dk, nk) { const int r_ = min((k) * 64 + lane, N - 1); dk = mask[(size_t)r_ * W + min((k), W - 1)]; \
                              nk = mask[(size_t)r_ * W + min((k) + 1, W - 1)]; }
        SWEEP_RQ(0, d0, n0) SWEEP_RQ(1, d1, n1) SWEEP_RQ(2, d2, n2) SWEEP_RQ(3, d3, n3)
#undef SWEEP_RQ
        unsigned long long carry = 0ULL;
        int num = 0;
        for (int i = 0; i < W; i += 4) {
            if (sweep_resolve(mask, N, W, max_keep, i, lane, sh, d0, n0, carry, num, keep SWEEP_TARGS)) break;
            if (i + 1 >= W) break;
            if (sweep_resolve(mask, N, W, max_keep, i + 1, lane, sh, d1, n1, carry, num, keep SWEEP_TARGS)) break;
            if (i + 2 >= W) break;
            if (sweep_resolve(mask, N, W, max_keep, i + 2, lane, sh, d2, n2, carry, num, keep SWEEP_TARGS)) break;
            if (i + 3 >= W) break;
            if (sweep_resolve(mask, N, W, max_keep, i + 3, lane, sh, d3, n3, carry, num, keep SWEEP_TARGS)) break;
        }
        if (lane == 0) *num_keep = (max_keep > 0 && num > max_keep) ? max_keep : num;
#ifdef SWEEP_TIMING
        if (lane == 0) for (int k = 0; k < 8; k++) ((unsigned long long*)mask)[k] = tacc[k];
#endif
    } else {
        const int fw = wave - 1, c = fw & 3, g = fw >> 2;
        // set k holds the rows of the block consumed at iterations k, k+4, ...: iteration i consumes block i-1
        unsigned long long p0[SWEEP_FOLD_ROWS], p1[SWEEP_FOLD_ROWS], p2[SWEEP_FOLD_ROWS], p3[SWEEP_FOLD_ROWS];
        bool h0 = false, h1, h2, h3;
        sweep_request(mask, N, W, 0, lane, c, g, sh, p1, h1);
        sweep_request(mask, N, W, 1, lane, c, g, sh, p2, h2);
        sweep_request(mask, N, W, 2, lane, c, g, sh, p3, h3);
        for (int i = 0; i < W; i += 4) {
            if (sweep_fold(mask, N, W, i, lane, c, g, sh, p0, h0 SWEEP_TARGS)) break;
            if (i + 1 >= W) break;
            if (sweep_fold(mask, N, W, i + 1, lane, c, g, sh, p1, h1 SWEEP_TARGS)) break;
            if (i + 2 >= W) break;
            if (sweep_fold(mask, N, W, i + 2, lane, c, g, sh, p2, h2 SWEEP_TARGS)) break;
            if (i + 3 >= W) break;
            if (sweep_fold(mask, N, W, i + 3, lane, c, g, sh, p3, h3 SWEEP_TARGS)) break;
        }
#ifdef SWEEP_TIMING
        if (tid == 64) for (int k = 0; k < 8; k++) ((unsigned long long*)mask)[8 + k] = tacc[k];
#endif
    }
}

static int pair_matrix(const char* op, bool iou, const float* a, int na, const float* b, int nb, float* out, hipStream_t s) {
    if (na < 0 || nb < 0) return prcnn_fail(PRCNN_EINVAL, "%s: bad shape", op);
    if (na == 0 || nb == 0) return PRCNN_OK;
    if (!a || !b || !out) return prcnn_fail(PRCNN_EINVAL, "%s: null pointer", op);
    dim3 grid(prcnn_divup(nb, 16), prcnn_divup(na, 16));
    if (iou) hipLaunchKernelGGL(pair_matrix_kernel<true>, grid, dim3(256), 0, s, a, na, b, nb, out);
    else hipLaunchKernelGGL(pair_matrix_kernel<false>, grid, dim3(256), 0, s, a, na, b, nb, out);
    PRCNN_LAUNCH_CHECK(op);
    return PRCNN_OK;
}

PRCNN_API int prcnn_boxes_overlap_bev(const float* boxes_a, int Na, const float* boxes_b, int Nb, float* out,
                                      prcnn_stream_t stream) {
    return pair_matrix("prcnn_boxes_overlap_bev", false, boxes_a, Na, boxes_b, Nb, out, (hipStream_t)stream);
}
PRCNN_API int prcnn_boxes_iou_bev(const float* boxes_a, int Na, const float* boxes_b, int Nb, float* out,
                                  prcnn_stream_t stream) {
    return pair_matrix("prcnn_boxes_iou_bev", true, boxes_a, Na, boxes_b, Nb, out, (hipStream_t)stream);
}

PRCNN_API size_t prcnn_nms_workspace_bytes(int N) {
    if (N <= 0) return 0;
    size_t W = (size_t)(N + 63) / 64;
    return W * 64 * W * sizeof(unsigned long long) + W * W;         // suppression mask (whole blocks of 64 rows) + one flag byte per 64 x 64 tile
}

PRCNN_API int prcnn_nms(const float* boxes, int N, float thresh, int kind, int max_keep, int64_t* keep, int32_t* num_keep,
                        void* workspace, size_t workspace_bytes, prcnn_stream_t stream) {
    PRCNN_REQUIRE(num_keep, "prcnn_nms: null num_keep");
    PRCNN_REQUIRE(N >= 0 && max_keep >= 0, "prcnn_nms: bad N=%d max_keep=%d", N, max_keep);
    PRCNN_REQUIRE(kind == PRCNN_NMS_ROTATED || kind == PRCNN_NMS_NORMAL, "prcnn_nms: bad kind %d", kind);
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) {
        if (prcnn_fill_words(num_keep, 0u, 1, s) != hipSuccess) return prcnn_fail(PRCNN_EHIP, "prcnn_nms: memset failed");
        return PRCNN_OK;
    }
    PRCNN_REQUIRE(boxes && keep && workspace, "prcnn_nms: null pointer");
    PRCNN_REQUIRE(workspace_bytes >= prcnn_nms_workspace_bytes(N), "prcnn_nms: workspace %zu < %zu bytes", workspace_bytes,
                  prcnn_nms_workspace_bytes(N));
    const int W = (N + 63) / 64;
    PRCNN_REQUIRE((size_t)W * 8 <= 60 * 1024, "prcnn_nms: N=%d too large for the LDS suppression bitmap", N);
    unsigned long long* mask = (unsigned long long*)workspace;
    unsigned char* tflag = (unsigned char*)(mask + (size_t)W * 64 * W);
    dim3 grid(W, W);
    if (kind == PRCNN_NMS_ROTATED)
        hipLaunchKernelGGL(nms_mask_kernel<PRCNN_NMS_ROTATED>, grid, dim3(64), 0, s, boxes, N, thresh, W, mask, tflag);
    else
        hipLaunchKernelGGL(nms_mask_kernel<PRCNN_NMS_NORMAL>, grid, dim3(64), 0, s, boxes, N, thresh, W, mask, tflag);
    PRCNN_LAUNCH_CHECK("prcnn_nms(mask)");
    hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(SWEEP_THREADS), (size_t)(W + 4) * 8 + (W <= SWEEP_FLAGS_LDS_MAX_W ? (size_t)W * W : 0), s, mask, tflag, N, W, max_keep, keep,
                       num_keep);
    PRCNN_LAUNCH_CHECK("prcnn_nms(sweep)");
    return PRCNN_OK;
}


// ref_trig.h on the device, element-wise (tests: device bits == the oracle's == the host libm's): fn 0 sinf(a), 1 cosf(a), 2 atan2f(a, b)
__global__ void ref_trig_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, int fn, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = fn == 0 ? prcnn_ref_sinf(a[i]) : (fn == 1 ? prcnn_ref_cosf(a[i]) : prcnn_ref_atan2f(a[i], b[i]));
}
PRCNN_API int prcnn_ref_trig(const float* a, const float* b, int n, int fn, float* out, prcnn_stream_t stream) {
    PRCNN_REQUIRE(n >= 0 && fn >= 0 && fn <= 2, "prcnn_ref_trig: bad arguments");
    if (n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(a && out && (fn != 2 || b), "prcnn_ref_trig: null pointer");
    hipLaunchKernelGGL(ref_trig_kernel, dim3(prcnn_divup(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b ? b : a, n, fn, out);
    PRCNN_LAUNCH_CHECK("prcnn_ref_trig");
    return PRCNN_OK;
}
