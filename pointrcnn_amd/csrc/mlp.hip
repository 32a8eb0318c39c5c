// mlp.hip -- fused per-point MLP layer for gfx950: [gather | interpolate | plain] A-tile -> fp32 MFMA ->
// bias + ReLU (+ max-pool over nsample) epilogue.
//
// Replaces the SharedMLP chain (Conv2d 1x1 + BatchNorm2d + ReLU, then F.max_pool2d(kernel=[1,nsample]))
// that upstream PointNet++ modules run through cuDNN on a materialised (B, C+3, npoint, nsample) grouped
// tensor [UPSTREAM, not in tree]; layer specs: lib/net/pointnet2_msg.py:20-45, lib/net/rpn.py:20-46,
// lib/net/rcnn_net.py:23-41.  The grouped tensor never exists here: rows of the GEMM A operand are
// gathered (or 3-NN-interpolated) from channels-last features straight into LDS.
//
// GEMM view: out[rows x Nout] = A[rows x K] * W^T[K x Nout].  Workgroup tile 128 rows x 64 (or 128) cols, 4 waves
// as 2(M) x 2(N); each wave owns 64 rows x 32 (or 64) cols = two (four) v_mfma_f32_32x32x2_f32 accumulators (exact fp32
// products, fp32 accumulation: the result is an fp32 FMA chain, no reduced precision anywhere).
// K is walked in chunks of 32 (four 8-wide k-blocks).  Inside a k-block the MFMA step s pairs
// k = 8*kb + s (lanes 0-31) with k = 8*kb + 4 + s (lanes 32-63): a lane's four A values for the block
// are then 16 contiguous bytes of its LDS row (one ds_read_b128), and the matching B values are
// 16 contiguous bytes of the pre-packed weight image (prcnn_pack_weight), which is staged into LDS by a
// straight linear copy.  A rows are padded to 36 floats so the 16 rows of each ds_read_b128 lane group
// land on 16 distinct 16-byte bank slots.
// Pipeline: global loads for chunk c+1 are issued before the MFMAs of chunk c and written to the other
// LDS buffer after them (one barrier per chunk).
#include "common.h"
#include <stdlib.h>

#define MLP_BM 128
#define MLP_BN 64
#define MLP_BK 32
#define MLP_THREADS 256
#define MLP_ALD 36

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { MODE_PLAIN = 0, MODE_GROUP = 1, MODE_INTERP = 2 };

struct MlpParams {
    long rows;            // total A rows
    int K, KB, NB;        // logical K, number of 8-wide k-blocks, number of 32-wide n-blocks
    const float* wpack;
    const float* bias;
    int Nout, relu;
    float* out;
    int ld_out, col_off, pool_ns;
    // MODE_PLAIN
    const float* in;
    int ld_in;
    // MODE_GROUP
    const float* xyz;
    const float* new_xyz;
    const int32_t* idx;
    const float* feat;
    int ld_feat, N, M, ns, C;
    // MODE_INTERP
    const float* known;
    const int32_t* idx3;
    const float* w3;
    const float* skip;
    int ld_known, ld_skip, n, m, C2, C1;
    int vec_a, vec_b;     // 16-byte vector loads legal for source a (feat/known/in) / source b (skip)
    // ---- first-layer hoisting (the first conv of a SharedMLP is linear, gathers are linear: they commute) ----
    // act = 1 (MODE_GROUP): the gathered source is Z = W_f . feat (pre-computed per SOURCE point); the A row is
    //        relu(Z[idx] + act_wx . dxyz + act_bias), K = C (no appended dxyz columns)
    // act = 2 (MODE_INTERP, C1 = 0): the interpolated source is Y = W . known; the A row is relu(interp(Y) + act_bias)
    int act;
    const float* act_wx;     // (K,3) row-major: the dxyz columns of the original first-layer weight
    const float* act_bias;   // (K), padded to a multiple of 4
    // epilogue add (MODE_PLAIN, no pooling): out += sum_j w3[row,j] * addY[b*m + idx3[row,j], n]  (before ReLU)
    const float* addY;
    int ldY;
    // device-side row count (dedup.hip): the launch is sized for `rows` (the worst case), the kernel processes
    // min(rows, *rows_dev * rows_unit) and workgroups past that exit at once.  NULL: rows is exact.
    const int32_t* rows_dev;
    int rows_unit;
    // segment-prefix live rows (MODE_PLAIN): rows come in segments of seg_rows, only the first seg_cnt[s] rows of segment s
    // carry data anyone reads (roipool3d pads an RoI that holds fewer points than it samples with copies of its first
    // rows); 128-row tiles that lie entirely in the dead tail of their segment are skipped.  NULL: every row is live.
    const int32_t* seg_cnt;
    int seg_rows;
    // XCD-aware tile order (gather modes): consecutive workgroups go round-robin over the 8 XCDs, each with its own 4 MB L2.
    // With xcd_tpf = 128-row tiles per frame (> 0), XCD x works through frames x, x+8, ... one after the other, so the
    // frame's gather source (Z / Y rows, 1-2 MB) stays resident in THAT XCD's L2 instead of every XCD streaming every frame.
    int xcd_tpf;
    // v2 layer kernel, plain rows: > 0 selects the XCD-aware 1-D tile order with this many column tiles per row tile
    int wgm_cols;
    // hoisted-FP addend (addY): 0 = every workgroup builds the interpolated tile in its epilogue; 1 = workgroups alternate by
    // dispatch slot between building it BEFORE the main loop (into the accumulators) and after it; 2 = all before (default)
    int addy_phase;
    // split-bf16 variant (mlp_layer_s_kernel): the pre-split weight image (prcnn_pack_weight_split) and the number of terms (3 / 6)
    const void* wsplit;
    int split_terms;
};

// With live-row segments the 128-row tile a workgroup works on is NOT blockIdx.x: the live tiles are the first one or two
// of every segment, i.e. blockIdx.x = 0 mod (seg_rows / 128), and consecutive workgroups go round-robin over the 8 XCDs --
// with seg_rows = 512 all live tiles would land on 2 of the 8 XCDs (measured: 25 % of the tiles cost 80 % of the time).
// Tile q of segment s is taken by workgroup q * nseg + s: the live tiles form a dense prefix of the grid.
__device__ __forceinline__ long tile_of_block(const MlpParams& P, long bid) {
    if (P.xcd_tpf > 0) {
        const long xcd = bid & 7, slot = bid >> 3;
        const long k = slot / P.xcd_tpf, t = slot - k * P.xcd_tpf;
        return (k * 8 + xcd) * P.xcd_tpf + t;
    }
    if (!P.seg_cnt) return bid;
    const long nseg = P.rows / P.seg_rows, tps = P.seg_rows / 128;     // P.rows: the launch's row count (seg excludes rows_dev)
    const long q = bid / nseg, sg = bid - q * nseg;
    return sg * tps + q;
}
__device__ __forceinline__ bool tile_dead(const MlpParams& P, long row0) {
    if (!P.seg_cnt) return false;
    const long s = row0 / P.seg_rows;
    return (int)(row0 - s * P.seg_rows) >= P.seg_cnt[s];
}
__device__ __forceinline__ long effective_rows(const MlpParams& P) {
    if (!P.rows_dev) return P.rows;
    const long r = (long)(*P.rows_dev) * P.rows_unit;
    return r < P.rows ? r : P.rows;
}

template <int MODE> struct RowMeta;
template <> struct RowMeta<MODE_PLAIN> { long off; bool valid; };
template <> struct RowMeta<MODE_GROUP> { long off; float dx, dy, dz; bool valid; };
template <> struct RowMeta<MODE_INTERP> { long o0, o1, o2, os; float w0, w1, w2; bool valid; };

template <int MODE> struct Raw;
template <> struct Raw<MODE_PLAIN> { float4 a; };
template <> struct Raw<MODE_GROUP> { float4 a; };
template <> struct Raw<MODE_INTERP> { float4 a, b, c; };

__host__ __device__ static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---- per-row metadata (computed once per thread for its 4 rows) -------------------------------
template <int MODE> __device__ __forceinline__ void make_meta(const MlpParams& P, long grow, RowMeta<MODE>& r);

template <> __device__ __forceinline__ void make_meta<MODE_PLAIN>(const MlpParams& P, long grow, RowMeta<MODE_PLAIN>& r) {
    r.valid = grow < P.rows;
    r.off = r.valid ? grow * (long)P.ld_in : 0;
}
template <> __device__ __forceinline__ void make_meta<MODE_GROUP>(const MlpParams& P, long grow, RowMeta<MODE_GROUP>& r) {
    r.valid = grow < P.rows;
    r.off = 0; r.dx = r.dy = r.dz = 0.f;
    if (r.valid) {
        long per_b = (long)P.M * P.ns;
        int b = (int)(grow / per_b);
        int m_ = (int)((grow - (long)b * per_b) / P.ns);
        int p = P.idx[grow];
        long pt = (long)b * P.N + p;
        r.off = pt * (long)P.ld_feat;
        float x = P.xyz[pt * 3], y = P.xyz[pt * 3 + 1], z = P.xyz[pt * 3 + 2];
        if (P.new_xyz) {
            const float* q = P.new_xyz + ((long)b * P.M + m_) * 3;
            x = x - q[0]; y = y - q[1]; z = z - q[2];
        }
        r.dx = x; r.dy = y; r.dz = z;
    }
}
template <> __device__ __forceinline__ void make_meta<MODE_INTERP>(const MlpParams& P, long grow, RowMeta<MODE_INTERP>& r) {
    r.valid = grow < P.rows;
    r.o0 = r.o1 = r.o2 = r.os = 0; r.w0 = r.w1 = r.w2 = 0.f;
    if (r.valid) {
        int b = (int)(grow / P.n);
        const int32_t* id = P.idx3 + grow * 3;
        const float* w = P.w3 + grow * 3;
        long base = (long)b * P.m;
        r.o0 = (base + id[0]) * (long)P.ld_known;
        r.o1 = (base + id[1]) * (long)P.ld_known;
        r.o2 = (base + id[2]) * (long)P.ld_known;
        r.os = grow * (long)P.ld_skip;
        r.w0 = w[0]; r.w1 = w[1]; r.w2 = w[2];
    }
}

// ---- raw global fetch of 4 consecutive k (k % 4 == 0) for one row; finish() turns it into A values
template <int MODE> __device__ __forceinline__ void fetch(const MlpParams& P, const RowMeta<MODE>& r, int k, Raw<MODE>& v);
template <int MODE> __device__ __forceinline__ float4 finish(const MlpParams& P, const RowMeta<MODE>& r, int k, const Raw<MODE>& v);

template <> __device__ __forceinline__ void fetch<MODE_PLAIN>(const MlpParams& P, const RowMeta<MODE_PLAIN>& r, int k, Raw<MODE_PLAIN>& v) {
    v.a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!r.valid || k >= P.K) return;
    const float* s = P.in + r.off + k;
    if (P.vec_a && k + 4 <= P.K) { v.a = ld4(s); return; }
    v.a.x = s[0];
    if (k + 1 < P.K) v.a.y = s[1];
    if (k + 2 < P.K) v.a.z = s[2];
    if (k + 3 < P.K) v.a.w = s[3];
}
template <> __device__ __forceinline__ float4 finish<MODE_PLAIN>(const MlpParams&, const RowMeta<MODE_PLAIN>&, int, const Raw<MODE_PLAIN>& v) {
    return v.a;
}

__device__ __forceinline__ float group_elem(const MlpParams& P, const RowMeta<MODE_GROUP>& r, int kk) {
    if (kk < P.C) return P.feat[r.off + kk];
    if (P.act) return 0.f;                       // hoisted mode: K == C, nothing appended
    int t = kk - P.C;
    return t == 0 ? r.dx : (t == 1 ? r.dy : (t == 2 ? r.dz : 0.f));
}
template <> __device__ __forceinline__ void fetch<MODE_GROUP>(const MlpParams& P, const RowMeta<MODE_GROUP>& r, int k, Raw<MODE_GROUP>& v) {
    v.a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!r.valid || k >= P.K) return;
    if (P.vec_a && k + 4 <= P.C) { v.a = ld4(P.feat + r.off + k); return; }
    v.a.x = group_elem(P, r, k);
    v.a.y = group_elem(P, r, k + 1);
    v.a.z = group_elem(P, r, k + 2);
    v.a.w = group_elem(P, r, k + 3);
}
// relu(z + wx . d + b) for 4 consecutive channels k..k+3 (act_wx rows are 3 floats, act_bias padded to x4)
__device__ __forceinline__ float4 act_group4(const MlpParams& P, float4 z, int k, float dx, float dy, float dz) {
    const float4 w0 = ld4(P.act_wx + (long)k * 3), w1 = ld4(P.act_wx + (long)k * 3 + 4), w2 = ld4(P.act_wx + (long)k * 3 + 8);
    const float4 b = ld4(P.act_bias + k);
    float4 o;
    o.x = fmaxf(z.x + (w0.x * dx + w0.y * dy + w0.z * dz) + b.x, 0.f);
    o.y = fmaxf(z.y + (w0.w * dx + w1.x * dy + w1.y * dz) + b.y, 0.f);
    o.z = fmaxf(z.z + (w1.z * dx + w1.w * dy + w2.x * dz) + b.z, 0.f);
    o.w = fmaxf(z.w + (w2.y * dx + w2.z * dy + w2.w * dz) + b.w, 0.f);
    return o;
}
template <> __device__ __forceinline__ float4 finish<MODE_GROUP>(const MlpParams& P, const RowMeta<MODE_GROUP>& r, int k, const Raw<MODE_GROUP>& v) {
    if (!P.act || !r.valid || k >= P.K) return v.a;
    float4 o = act_group4(P, v.a, k, r.dx, r.dy, r.dz);
    if (k + 1 >= P.K) o.y = 0.f;                  // keep the zero padding beyond K
    if (k + 2 >= P.K) o.z = 0.f;
    if (k + 3 >= P.K) o.w = 0.f;
    return o;
}

__device__ __forceinline__ float interp1(float w0, float f0, float w1, float f1, float w2, float f2) {
    // (w0*f0 + w1*f1) + w2*f2, individually rounded: bit-identical to three_interpolate
    return __fadd_rn(__fadd_rn(__fmul_rn(w0, f0), __fmul_rn(w1, f1)), __fmul_rn(w2, f2));
}
template <> __device__ __forceinline__ void fetch<MODE_INTERP>(const MlpParams& P, const RowMeta<MODE_INTERP>& r, int k, Raw<MODE_INTERP>& v) {
    v.a = v.b = v.c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!r.valid || k >= P.K) return;
    if (k + 4 <= P.C2) {
        if (P.vec_a) {
            v.a = ld4(P.known + r.o0 + k); v.b = ld4(P.known + r.o1 + k); v.c = ld4(P.known + r.o2 + k);
        } else {
            const float *a = P.known + r.o0 + k, *b = P.known + r.o1 + k, *c = P.known + r.o2 + k;
            v.a = make_float4(a[0], a[1], a[2], a[3]);
            v.b = make_float4(b[0], b[1], b[2], b[3]);
            v.c = make_float4(c[0], c[1], c[2], c[3]);
        }
        return;
    }
    if (k >= P.C2) {        // skip-connection part: stored in v.a, passed through by finish()
        int ks = k - P.C2;
        const float* s = P.skip + r.os + ks;
        if (P.vec_b && ks + 4 <= P.C1) { v.a = ld4(s); return; }
        if (ks < P.C1) v.a.x = s[0];
        if (ks + 1 < P.C1) v.a.y = s[1];
        if (ks + 2 < P.C1) v.a.z = s[2];
        if (ks + 3 < P.C1) v.a.w = s[3];
        return;
    }
    // 4-group straddling the C2 boundary (C2 % 4 != 0): element-wise, v.b/v.c carry the neighbours
    auto one = [&](int kk, float& a, float& b, float& c) {
        if (kk < P.C2) { a = P.known[r.o0 + kk]; b = P.known[r.o1 + kk]; c = P.known[r.o2 + kk]; }
        else if (kk - P.C2 < P.C1) { a = P.skip[r.os + kk - P.C2]; }
    };
    one(k, v.a.x, v.b.x, v.c.x);
    one(k + 1, v.a.y, v.b.y, v.c.y);
    one(k + 2, v.a.z, v.b.z, v.c.z);
    one(k + 3, v.a.w, v.b.w, v.c.w);
}
template <> __device__ __forceinline__ float4 finish<MODE_INTERP>(const MlpParams& P, const RowMeta<MODE_INTERP>& r, int k, const Raw<MODE_INTERP>& v) {
    if (k >= P.C2) return v.a;
    float4 o;
    o.x = interp1(r.w0, v.a.x, r.w1, v.b.x, r.w2, v.c.x);
    o.y = (k + 1 < P.C2) ? interp1(r.w0, v.a.y, r.w1, v.b.y, r.w2, v.c.y) : v.a.y;
    o.z = (k + 2 < P.C2) ? interp1(r.w0, v.a.z, r.w1, v.b.z, r.w2, v.c.z) : v.a.z;
    o.w = (k + 3 < P.C2) ? interp1(r.w0, v.a.w, r.w1, v.b.w, r.w2, v.c.w) : v.a.w;
    if (P.act && r.valid) {                       // hoisted mode (C1 == 0): relu(interp(Y) + b), zero padding kept
        const float4 b = ld4(P.act_bias + k);
        o.x = fmaxf(o.x + b.x, 0.f);
        o.y = (k + 1 < P.C2) ? fmaxf(o.y + b.y, 0.f) : 0.f;
        o.z = (k + 2 < P.C2) ? fmaxf(o.z + b.z, 0.f) : 0.f;
        o.w = (k + 3 < P.C2) ? fmaxf(o.w + b.w, 0.f) : 0.f;
    }
    return o;
}

// sum_j w3[row,j] * addY[(b*m + idx3[row,j]) * ldY + n]: the interpolated pre-activation of the hoisted FP first layer
__device__ __forceinline__ float interp_gather(const MlpParams& P, long row, int n) {
    const int b = (int)(row / P.n);
    const int32_t* id = P.idx3 + row * 3;
    const float* w = P.w3 + row * 3;
    const float* y = P.addY + (long)b * P.m * P.ldY + n;
    return (w[0] * y[(long)id[0] * P.ldY] + w[1] * y[(long)id[1] * P.ldY]) + w[2] * y[(long)id[2] * P.ldY];
}

// Interpolated addend tile of the hoisted FP first layer: T[row, c] = sum_j w3[row, j] * addY[idx3[row, j], ncol0 + c] for the
// 128 rows of the tile and one 64-column half.  A thread stages the 16-byte column piece (tid % 16) of rows tid / 16 + 16 u.
__device__ __forceinline__ void addy_stage_half(const MlpParams& P, int tid, long row0, int nb0, int hf, float* T, int ldt) {
        const int ncol0 = (nb0 + hf * 2) * 32;
        const int c = (tid & 15) * 4, n = ncol0 + c, rbase = tid >> 4;
        const bool col_ok = n < P.Nout;
        // frame of a row: a 128-row tile spans at most two frames when a frame has at least 128 rows
        const long bfirst = row0 / P.n;                       // (workgroup-uniform: one scalar division)
        const long next_frame_row = (bfirst + 1) * (long)P.n;
        const bool two_frames_at_most = P.n >= MLP_BM;
#pragma unroll
        for (int q = 0; q < MLP_BM / 64; q++) {
            int32_t id[4][3];
            float w[4][3];
            const float* ybase[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const long g = row0 + rbase + 16 * (q * 4 + u);
                ok[u] = g < P.rows && col_ok;
                const long gg = ok[u] ? g : row0;               // (row0 < P.rows: the tile exists)
                const int32_t* ip = P.idx3 + gg * 3;
                const float* wp = P.w3 + gg * 3;
                id[u][0] = ip[0]; id[u][1] = ip[1]; id[u][2] = ip[2];
                w[u][0] = wp[0]; w[u][1] = wp[1]; w[u][2] = wp[2];
                const long b = two_frames_at_most ? bfirst + (gg >= next_frame_row ? 1 : 0) : gg / P.n;
                ybase[u] = P.addY + b * P.m * P.ldY + (col_ok ? n : 0);
            }
            float4 y[4][3];
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int k = 0; k < 3; k++) y[u][k] = ld4(ybase[u] + (long)id[u][k] * P.ldY);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok[u]) {
                    o.x = (w[u][0] * y[u][0].x + w[u][1] * y[u][1].x) + w[u][2] * y[u][2].x;   // same expression as interp_gather
                    o.y = (w[u][0] * y[u][0].y + w[u][1] * y[u][1].y) + w[u][2] * y[u][2].y;
                    o.z = (w[u][0] * y[u][0].z + w[u][1] * y[u][1].z) + w[u][2] * y[u][2].z;
                    o.w = (w[u][0] * y[u][0].w + w[u][1] * y[u][1].w) + w[u][2] * y[u][2].w;
                }
                *reinterpret_cast<float4*>(T + (rbase + 16 * (q * 4 + u)) * ldt + c) = o;
            }
        }
    }

// Epilogue shared by the layer kernels: bias + ReLU (+ hoisted-FP interpolated addend, + max-pool over nsample rows) and the
// stores.  As0 / Bs0: the (now idle) operand LDS, reused to stage the interpolated addend tile (128 x 72 and 128 x 64 floats).
template <int MODE, int WNB, bool ADDY>
__device__ __forceinline__ void layer_epilogue(const MlpParams& P, f32x16 (&acc)[2][WNB], float* As0, float* Bs0, int tid, long row0,
                                               int nb0, bool n_active, bool addy_done = false) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // SGPR: the full-tile test below is a scalar branch
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, j = lane & 31;
    // ---- epilogue -------------------------------------------------------------------------------
    // Hoisted FP first layer (addY): every output element needs sum_j w_j * Y[idx_j, n].  Fetching that per accumulator
    // element costs three 4-byte gathers per element; instead the workgroup builds the interpolated 128 x (64*WNB) tile
    // once -- 16-byte loads, a row's 64 columns = 256 contiguous bytes per neighbour -- into the now idle operand
    // buffers (As: 128 x 72 floats, Bs: 128 x 64) and the accumulators pick their elements up from LDS.
    // Bs0 == nullptr (v2 kernel: only the A half of the operand LDS exists): the two 64-column halves of a wide tile are staged
    // one after the other into As0, each followed by the stores of the column blocks that lie in it.
    bool staged = false;
    int npass = 1;
    // A thread stages the 16-byte column piece (tid % 16) of rows tid / 16 + 16 u, u = 0..7.  Four rows at a time: their index /
    // weight triples first, then their twelve gathers, then the arithmetic -- two dependent round trips per four rows.  (Row by
    // row, with a 64-bit division for the frame of each, the loop was eight times two exposed round trips: ~10 of the ~30 us a
    // workgroup of FP1's layer stays resident, whatever the locality of the gathers -- tools/interp_volume_probe.py.)
    auto stage_half = [&](int hf, float* T, int ldt) { addy_stage_half(P, tid, row0, nb0, hf, T, ldt); };
    if (MODE == MODE_PLAIN && ADDY && P.addY && !addy_done) {
        staged = (P.Nout % 4 == 0) && (P.ldY % 4 == 0) && aligned16(P.addY);
        if (staged && Bs0) {
#pragma unroll
            for (int hf = 0; hf < WNB; hf++) stage_half(hf, hf == 0 ? As0 : Bs0, hf == 0 ? 72 : 64);
            __syncthreads();
        } else if (staged) {
            npass = WNB;
        }
    }
    if (!n_active && npass == 1) return;              // (with several passes every wave has barriers ahead of it)
    const long wrow0 = row0 + wm * 64;
#pragma unroll
    for (int pass = 0; pass < WNB; pass++) {
    if (pass >= npass) break;
    if (staged && !Bs0) {
        if (pass > 0) __syncthreads();                // the previous half has been consumed
        stage_half(pass, As0, 72);
        __syncthreads();
    }
    if (!n_active) continue;
#pragma unroll
    for (int nn = 0; nn < WNB; nn++) {
        const int nb = nb0 + wn * WNB + nn;
        if (nb >= P.NB) continue;
        if (staged && !Bs0 && ((wn * WNB + nn) >> 1) != pass) continue;
        const int n = nb * 32 + j;
        const bool n_ok = n < P.Nout;
        const float bias = (P.bias && n_ok) ? P.bias[n] : 0.f;
        const f32x16& acc0 = acc[0][nn];
        const f32x16& acc1 = acc[1][nn];
        const bool use_addy = ADDY && P.addY && !addy_done;
        if (P.pool_ns == 0 && wrow0 + 64 <= P.rows && nb * 32 + 32 <= P.Nout && (!use_addy || staged)) {
            // The wave's whole 64 x 32 block lies inside the output (all but the last row tile / a ragged last column block):
            // no bounds test per element.  With one, every store sits in its own exec-masked region and the compiler opens each
            // with s_waitcnt vmcnt(0) -- on gfx9 that counter includes STORES, so the 32 stores of a block went out one
            // memory round trip after the other.
            const int blk = wn * WNB + nn;
            const float* T = ((blk >> 1) == 0 || !Bs0) ? As0 : Bs0;
            const int ldt = ((blk >> 1) == 0 || !Bs0) ? 72 : 64, tc = (blk & 1) * 32 + j;
            float* o = P.out + (wrow0 + 4 * h) * P.ld_out + P.col_off + n;
            const long ld = P.ld_out;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = (r & 3) + 8 * (r >> 2);
                float v0 = acc0[r] + bias, v1 = acc1[r] + bias;
                if (ADDY && use_addy) {
                    v0 += T[(wm * 64 + rr + 4 * h) * ldt + tc];
                    v1 += T[(wm * 64 + 32 + rr + 4 * h) * ldt + tc];
                }
                v0 = P.relu ? fmaxf(v0, 0.f) : v0;
                v1 = P.relu ? fmaxf(v1, 0.f) : v1;
                o[rr * ld] = v0;
                o[(32 + rr) * ld] = v1;
            }
        } else if (P.pool_ns == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
                long g0 = wrow0 + rin, g1 = wrow0 + 32 + rin;
                float v0 = acc0[r] + bias, v1 = acc1[r] + bias;
                if (ADDY && P.addY && n_ok && !addy_done) {     // hoisted FP first layer: + sum_j w_j * Y[idx_j, n]
                    if (staged) {
                        const int blk = wn * WNB + nn;
                        const float* T = ((blk >> 1) == 0 || !Bs0) ? As0 : Bs0;
                        const int ldt = ((blk >> 1) == 0 || !Bs0) ? 72 : 64, tc = (blk & 1) * 32 + j;
                        v0 += T[(wm * 64 + rin) * ldt + tc];
                        v1 += T[(wm * 64 + 32 + rin) * ldt + tc];
                    } else {
                        if (g0 < P.rows) v0 += interp_gather(P, g0, n);
                        if (g1 < P.rows) v1 += interp_gather(P, g1, n);
                    }
                }
                if (P.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                if (n_ok && g0 < P.rows) P.out[g0 * P.ld_out + P.col_off + n] = v0;
                if (n_ok && g1 < P.rows) P.out[g1 * P.ld_out + P.col_off + n] = v1;
            }
        } else {
            // max over nsample consecutive rows.  max commutes with the (monotone) bias add and ReLU, so
            // they are applied once per pooled value: bit-identical to pooling the activated rows.
            float lo0 = acc0[0], hi0 = acc0[8], lo1 = acc1[0], hi1 = acc1[8];
#pragma unroll
            for (int r = 1; r < 8; r++) {
                lo0 = fmaxf(lo0, acc0[r]); hi0 = fmaxf(hi0, acc0[8 + r]);
                lo1 = fmaxf(lo1, acc1[r]); hi1 = fmaxf(hi1, acc1[8 + r]);
            }
            lo0 = fmaxf(lo0, __shfl_xor(lo0, 32)); hi0 = fmaxf(hi0, __shfl_xor(hi0, 32));
            lo1 = fmaxf(lo1, __shfl_xor(lo1, 32)); hi1 = fmaxf(hi1, __shfl_xor(hi1, 32));
            const long groups = P.rows / P.pool_ns;
            float v[4]; long g[4]; int cnt;
            if (P.pool_ns == 16) {
                long g0 = wrow0 / 16;
                v[0] = lo0; v[1] = hi0; v[2] = lo1; v[3] = hi1; g[0] = g0; g[1] = g0 + 1; g[2] = g0 + 2; g[3] = g0 + 3; cnt = 4;
            } else if (P.pool_ns == 32) {
                long g0 = wrow0 / 32;
                v[0] = fmaxf(lo0, hi0); v[1] = fmaxf(lo1, hi1); g[0] = g0; g[1] = g0 + 1; cnt = 2;
            } else {
                v[0] = fmaxf(fmaxf(lo0, hi0), fmaxf(lo1, hi1)); g[0] = wrow0 / 64; cnt = 1;
            }
            if (h == 0 && n_ok) {
                for (int t = 0; t < cnt; t++) {
                    float o = v[t] + bias;
                    if (P.relu) o = fmaxf(o, 0.f);
                    if (g[t] < groups) P.out[g[t] * P.ld_out + P.col_off + n] = o;
                }
            }
        }
    }
    }
}

// WNB = 32-column blocks per wave: 1 -> workgroup tile 128x64 (narrow layers), 2 -> 128x128 (wide layers: twice
// the MFMAs per LDS operand read, and a gathered / interpolated A tile is rebuilt for half as many column tiles).
template <int MODE, int WNB>
__global__ __launch_bounds__(MLP_THREADS) void mlp_layer_kernel(const MlpParams Pin) {
    MlpParams P = Pin;
    P.rows = effective_rows(Pin);
    const long tile_id = tile_of_block(P, blockIdx.x);
    if (tile_id * MLP_BM >= P.rows) return;              // workgroup-uniform (device-side row count)
    if (tile_dead(P, tile_id * MLP_BM)) return;
    constexpr int QN = 2 * WNB;                          // n-blocks per workgroup
    __shared__ __attribute__((aligned(16))) float As[2][MLP_BM * MLP_ALD];
    __shared__ __attribute__((aligned(16))) float Bs[2][QN * 4 * 256];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, j = lane & 31;
    const long row0 = tile_id * MLP_BM;
    const int nb0 = blockIdx.y * QN;
    const int nchunks = (P.KB + 3) >> 2;

    // fill-phase assignment: thread -> float4 column c4 of rows r0 + 32*u
    const int c4 = tid & 7, r0 = tid >> 3;
    RowMeta<MODE> meta[4];
#pragma unroll
    for (int u = 0; u < 4; u++) make_meta<MODE>(P, row0 + r0 + 32 * u, meta[u]);

    Raw<MODE> ra[4];
    float4 rb[QN];
    // hoisted GROUP rows: the activated-gather parameters of the NEXT chunk's 4 channels travel with its gather instead of
    // being loaded (a dependent L1/L2 round trip) at the moment the chunk is written to LDS
    const bool group_act = MODE == MODE_GROUP && P.act != 0;
    float4 aw0 = make_float4(0.f, 0.f, 0.f, 0.f), aw1 = aw0, aw2 = aw0, ab = aw0;
    auto load_chunk = [&](int c) {
        const int k = c * MLP_BK + c4 * 4;
        if (group_act && k < P.K) {
            aw0 = ld4(P.act_wx + (long)k * 3); aw1 = ld4(P.act_wx + (long)k * 3 + 4); aw2 = ld4(P.act_wx + (long)k * 3 + 8);
            ab = ld4(P.act_bias + k);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) fetch<MODE>(P, meta[u], k, ra[u]);
#pragma unroll
        for (int u = 0; u < QN; u++) {
            int f = tid + 256 * u;                 // float4 index inside the [q][kbl][64] image
            int q = f >> 8, rr = f & 255, kbl = rr >> 6, o4 = rr & 63;
            int nb = nb0 + q, kb = c * 4 + kbl;
            rb[u] = (nb < P.NB && kb < P.KB) ? ld4(P.wpack + ((long)nb * P.KB + kb) * 256 + o4 * 4)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_chunk = [&](int c, int buf) {
        const int k = c * MLP_BK + c4 * 4;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float4 v;
            if constexpr (MODE == MODE_GROUP) {
                if (group_act && meta[u].valid && k < P.K) {          // == finish<MODE_GROUP> with the prefetched parameters
                    const float dx = meta[u].dx, dy = meta[u].dy, dz = meta[u].dz;
                    const float4 z = ra[u].a;
                    v.x = fmaxf(z.x + (aw0.x * dx + aw0.y * dy + aw0.z * dz) + ab.x, 0.f);
                    v.y = fmaxf(z.y + (aw0.w * dx + aw1.x * dy + aw1.y * dz) + ab.y, 0.f);
                    v.z = fmaxf(z.z + (aw1.z * dx + aw1.w * dy + aw2.x * dz) + ab.z, 0.f);
                    v.w = fmaxf(z.w + (aw2.y * dx + aw2.z * dy + aw2.w * dz) + ab.w, 0.f);
                    if (k + 1 >= P.K) v.y = 0.f;
                    if (k + 2 >= P.K) v.z = 0.f;
                    if (k + 3 >= P.K) v.w = 0.f;
                } else {
                    v = finish<MODE>(P, meta[u], k, ra[u]);
                }
            } else {
                v = finish<MODE>(P, meta[u], k, ra[u]);
            }
            *reinterpret_cast<float4*>(&As[buf][(r0 + 32 * u) * MLP_ALD + c4 * 4]) = v;
        }
#pragma unroll
        for (int u = 0; u < QN; u++) *reinterpret_cast<float4*>(&Bs[buf][(tid + 256 * u) * 4]) = rb[u];
    };

    f32x16 acc[2][WNB];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int n = 0; n < WNB; n++) acc[r][n] = (f32x16){0};
    const bool n_active = (nb0 + wn * WNB) < P.NB;      // at least this wave's first column block exists

    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk(c + 1);
        if (n_active) {
            const int nkb = min(4, P.KB - c * 4);
            const float* a_base = &As[buf][(wm * 64 + j) * MLP_ALD + 4 * h];
            const float* b_base = &Bs[buf][wn * WNB * 1024 + lane * 4];
            for (int kbl = 0; kbl < nkb; kbl++) {
                float4 a[2], bq[WNB];
                a[0] = *reinterpret_cast<const float4*>(a_base + kbl * 8);
                a[1] = *reinterpret_cast<const float4*>(a_base + 32 * MLP_ALD + kbl * 8);
#pragma unroll
                for (int n = 0; n < WNB; n++) bq[n] = *reinterpret_cast<const float4*>(b_base + n * 1024 + kbl * 256);
                // k-step outermost: consecutive MFMAs hit different accumulators (no dependent-accumulator stall)
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].x, bq[n].x, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].y, bq[n].y, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].z, bq[n].z, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].w, bq[n].w, acc[r][n], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) store_chunk(c + 1, buf ^ 1);
        __syncthreads();
    }

    layer_epilogue<MODE, WNB, true>(P, acc, &As[0][0], &Bs[0][0], tid, row0, nb0, n_active);
}

// ---- v2 of the layer kernel: B operand straight from L2 ------------------------------------------------------------------
// The packed weight image IS the MFMA B-operand layout (lane l of (n-block, k-block) owns 16 contiguous bytes), so a wave can
// load its own B operands directly into registers; only the A tile (gathered / interpolated / plain rows, shared by the
// two waves of a column pair) goes through LDS.  What that buys over mlp_layer_kernel:
//   * LDS per workgroup 69 -> 37 KB and no B staging registers / ds_writes: three or four workgroups per CU instead of two
//     (LayerBOcc below), so a workgroup that is in its prologue (row metadata, first chunk's global -> LDS round trip) or
//     its store epilogue leaves the others to feed the MFMA pipe -- the un-overlapped per-tile fixed costs were what
//     separated the v1 kernel (90-100 TFLOP/s on plain rows) from its own inner loop (150);
//   * the B stream is a 4-slot register ring requested four k-blocks (4096 MFMA cycles) ahead: its L2 latency never shows.
// The weights of a layer are <= 1 MB and every workgroup reads all of them: they stay L2 / L1 resident.
// Plain rows use an XCD-aware 1-D tile order: workgroup b runs on XCD b % 8; the column tiles of one row tile are handed to
// consecutive slots of ONE XCD, so the A rows they share are fetched from HBM once and hit that XCD's L2 afterwards.
// waves per SIMD the register allocation is held to: 3 where it fits without spilling (plain rows; the narrow grouped tile),
// 2 for the gather modes whose row metadata + raw gather registers need the room (they still gain the LDS and the B ring)
#ifndef PRCNN_ABL
#define PRCNN_ABL 0            // ablation builds of mlp_layer_b_kernel (tools/build_ablation.py): bit 0 no epilogue, 1 no A loads in
#endif                         // the loop, 2 no B loads in the loop, 3 no LDS refill + barrier in the loop -- never set in the product
                               // (mlp_layer_s_kernel: 16 no epilogue, 32 no B loads in the loop, 64 no A loads / split / LDS writes, 128 no MFMAs, 256 no LDS reads / barrier in the loop)
template <int MODE, int WNB, bool FAST, bool ADDY> struct LayerBOcc {
    // plain rows, straight-line form: FOUR workgroups per CU (<= 128 registers).  The plain GEMMs of the graph have power-of-two
    // tile counts (256 .. 4096): 1024 resident workgroups run them in whole rounds, whereas three per CU (768) leaves a
    // ragged last round of one workgroup per CU with nothing to hide its latencies behind (measured: 32768 x 512 -> 512 took
    // 213 us at three per CU against 170 us at two).  With four waves sharing a SIMD's matrix pipe a k-block lasts ~4096
    // cycles of wall time, so a two-slot B ring (two k-blocks ahead) is ample.
    // ADDY (hoisted-FP first layer: the epilogue stages the interpolated addend tile through LDS while the accumulators are
    // live): three per CU -- at 128 registers that epilogue spills
    static constexpr int W = ADDY ? (WNB == 1 ? 4 : 3) : (MODE == MODE_PLAIN && FAST) ? 4 : ((MODE == MODE_PLAIN || (MODE == MODE_GROUP && WNB == 1)) ? 3 : 2);
    static constexpr int RING = (MODE == MODE_PLAIN && FAST && W == 4) ? 2 : 4;
};
// FAST: K a multiple of the 32-wide chunk, 16-byte aligned source rows (and, grouped, the hoisted activated-gather form):
// rows are CLAMPED to the last live row instead of bounds-checked (their results are never stored), so the whole main loop
// is straight-line code -- no exec-mask branches around the loads, which is what lets the compiler keep the A chunk and the B
// ring in flight with counted waits instead of draining vmcnt to 0 at every join.
template <int MODE, int WNB, bool FAST, bool ADDY>
__global__ __launch_bounds__(MLP_THREADS, (LayerBOcc<MODE, WNB, FAST, ADDY>::W)) void mlp_layer_b_kernel(const MlpParams Pin) {
    MlpParams P = Pin;
    P.rows = effective_rows(Pin);
    constexpr int QN = 2 * WNB;                          // n-blocks per workgroup
    long tile_id;
    int nb0;
    if (P.wgm_cols > 0) {                                // XCD-aware 1-D order (plain rows)
        const long bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const long rl = slot / P.wgm_cols;
        tile_id = rl * 8 + xcd;
        nb0 = (int)(slot - rl * P.wgm_cols) * QN;
    } else {
        tile_id = tile_of_block(P, blockIdx.x);
        nb0 = blockIdx.y * QN;
    }
    if (tile_id * MLP_BM >= P.rows) return;              // workgroup-uniform (device-side row count / grid padding)
    if (tile_dead(P, tile_id * MLP_BM)) return;
    __shared__ __attribute__((aligned(16))) float As[2][MLP_BM * MLP_ALD];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave index in an SGPR: uniform branches
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, j = lane & 31;
    const long row0 = tile_id * MLP_BM;
    const int nchunks = (P.KB + 3) >> 2;

    const int c4 = tid & 7, r0 = tid >> 3;
    RowMeta<MODE> meta[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        long grow = row0 + r0 + 32 * u;
        if (FAST && grow >= P.rows) grow = P.rows - 1;
        make_meta<MODE>(P, grow, meta[u]);
    }

    Raw<MODE> ra[4];
    const bool group_act = MODE == MODE_GROUP && P.act != 0;
    float4 aw0 = make_float4(0.f, 0.f, 0.f, 0.f), aw1 = aw0, aw2 = aw0, ab = aw0;
    auto load_chunk = [&](int c) {
        const int k = c * MLP_BK + c4 * 4;
        if constexpr (FAST) {
            if constexpr (MODE == MODE_GROUP) {
                aw0 = ld4(P.act_wx + (long)k * 3); aw1 = ld4(P.act_wx + (long)k * 3 + 4); aw2 = ld4(P.act_wx + (long)k * 3 + 8);
                ab = ld4(P.act_bias + k);
#pragma unroll
                for (int u = 0; u < 4; u++) ra[u].a = ld4(P.feat + meta[u].off + k);
            } else {
#pragma unroll
                for (int u = 0; u < 4; u++) ra[u].a = ld4(P.in + meta[u].off + k);
            }
        } else {
            if (group_act && k < P.K) {
                aw0 = ld4(P.act_wx + (long)k * 3); aw1 = ld4(P.act_wx + (long)k * 3 + 4); aw2 = ld4(P.act_wx + (long)k * 3 + 8);
                ab = ld4(P.act_bias + k);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) fetch<MODE>(P, meta[u], k, ra[u]);
        }
    };
    auto store_chunk = [&](int c, int buf) {
        const int k = c * MLP_BK + c4 * 4;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float4 v;
            if constexpr (MODE == MODE_GROUP) {
                if (FAST || (group_act && meta[u].valid && k < P.K)) {          // == finish<MODE_GROUP> with the prefetched parameters
                    const float dx = meta[u].dx, dy = meta[u].dy, dz = meta[u].dz;
                    const float4 z = ra[u].a;
                    v.x = fmaxf(z.x + (aw0.x * dx + aw0.y * dy + aw0.z * dz) + ab.x, 0.f);
                    v.y = fmaxf(z.y + (aw0.w * dx + aw1.x * dy + aw1.y * dz) + ab.y, 0.f);
                    v.z = fmaxf(z.z + (aw1.z * dx + aw1.w * dy + aw2.x * dz) + ab.z, 0.f);
                    v.w = fmaxf(z.w + (aw2.y * dx + aw2.z * dy + aw2.w * dz) + ab.w, 0.f);
                    if (!FAST) {
                        if (k + 1 >= P.K) v.y = 0.f;
                        if (k + 2 >= P.K) v.z = 0.f;
                        if (k + 3 >= P.K) v.w = 0.f;
                    }
                } else {
                    v = finish<MODE>(P, meta[u], k, ra[u]);
                }
            } else if constexpr (MODE == MODE_PLAIN) {
                v = ra[u].a;
            } else {
                v = finish<MODE>(P, meta[u], k, ra[u]);
            }
            *reinterpret_cast<float4*>(&As[buf][(r0 + 32 * u) * MLP_ALD + c4 * 4]) = v;
        }
    };

    f32x16 acc[2][WNB];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int n = 0; n < WNB; n++) acc[r][n] = (f32x16){0};
    const bool n_active = (nb0 + wn * WNB) < P.NB;      // at least this wave's first column block exists (wave-uniform)

    // B ring: slot (g & 3) holds this wave's operands of k-block g, requested four k-blocks ahead.  Loads are unconditional:
    // column blocks past the layer's width re-read the last block (their results are never stored), k-blocks past K re-read
    // the last k-block -- the A tile is zero there, so the extra MFMAs of a ragged last chunk add exact zeros.
    constexpr int RING = LayerBOcc<MODE, WNB, FAST, ADDY>::RING;      // slots = prefetch distance in k-blocks (4: one chunk, 2: half)
    float4 bq[RING][WNB];
    const float* bptr[WNB];
#pragma unroll
    for (int n = 0; n < WNB; n++) bptr[n] = P.wpack + ((long)min(nb0 + wn * WNB + n, P.NB - 1) * P.KB) * 256 + lane * 4;
    const int kb_last = P.KB - 1;
    auto load_b = [&](int g, int slot) {
        const long off = (long)min(g, kb_last) * 256;
#pragma unroll
        for (int n = 0; n < WNB; n++) bq[slot][n] = ld4(bptr[n] + off);
    };
#pragma unroll
    for (int g = 0; g < RING; g++) load_b(g, g);

    load_chunk(0);
    // Hoisted-FP addend, built FIRST.  In the epilogue the interpolated tile (per 16-byte piece three gathers, their addressing,
    // 20 multiply-adds, an LDS store: ~400 instructions per thread) ran when the main loop was over, its gathers exposed.  Built
    // before the main loop, straight into the accumulators (through the still idle operand LDS), its gathers are in flight
    // together with the B ring and the first A chunk requested above, and the tail of the tile is the plain store epilogue.
    // Measured (A/B by PRCNN_ADDY_PHASE, same box): FP1's 131072 x 96 -> 256 launch 119 -> 108 us; alternating first / last by
    // dispatch slot (phase 1) gives the same 108 us, so it is the overlap with the prologue loads, not de-phasing, that pays;
    // the two 512-wide launches are unchanged (their staging is 4 % of their time).
    bool addy_done = false;
    if constexpr (ADDY && WNB == 1) {            // (the wide tile has no registers to spare for the second code path: it spills)
        const bool first = P.addY && (P.addy_phase == 2 || (P.addy_phase == 1 && ((blockIdx.x >> 8) & 1)));
        const bool stageable = (P.Nout % 4 == 0) && (P.ldY % 4 == 0) && aligned16(P.addY);
        if (first && stageable) {                           // workgroup-uniform
            float* T = &As[0][0];
#pragma unroll
            for (int pass = 0; pass < WNB; pass++) {
                if (pass > 0) __syncthreads();
                addy_stage_half(P, tid, row0, nb0, pass, T, 72);
                __syncthreads();
#pragma unroll
                for (int nn = 0; nn < WNB; nn++) {
                    const int blk = wn * WNB + nn;
                    if ((blk >> 1) != pass) continue;
                    const int tc = (blk & 1) * 32 + j;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
                        acc[0][nn][r] = T[(wm * 64 + rin) * 72 + tc];
                        acc[1][nn][r] = T[(wm * 64 + 32 + rin) * 72 + tc];
                    }
                }
            }
            __syncthreads();
            addy_done = true;
        }
    }
    store_chunk(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        const int buf = c & 1;
        // unconditional (the last iteration re-requests the last chunk and stores it into the idle buffer): a branch around
        // these loads would make the compiler's wait counts assume the path WITHOUT them, i.e. force them early on the other
        if (!(PRCNN_ABL & 2)) load_chunk(min(c + 1, nchunks - 1));
        const float* a_base = &As[(PRCNN_ABL & 8) ? 0 : buf][(wm * 64 + j) * MLP_ALD + 4 * h];
#pragma unroll
        for (int kbl = 0; kbl < 4; kbl++) {
            if (n_active) {
                float4 a[2];
                a[0] = *reinterpret_cast<const float4*>(a_base + kbl * 8);
                a[1] = *reinterpret_cast<const float4*>(a_base + 32 * MLP_ALD + kbl * 8);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].x, bq[kbl % RING][n].x, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].y, bq[kbl % RING][n].y, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].z, bq[kbl % RING][n].z, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].w, bq[kbl % RING][n].w, acc[r][n], 0, 0, 0);
            }
            // Refill the slot just consumed with k-block g + 4.  Every B operand a chunk uses was therefore requested during
            // the PREVIOUS chunk, i.e. before this chunk's A loads: vmcnt retires in order, so waiting for a B operand never
            // drags the younger A loads along -- they have the whole chunk (4096 MFMA cycles) to arrive.
            if (!(PRCNN_ABL & 4)) load_b(c * 4 + kbl + RING, kbl % RING);
        }
        if (!(PRCNN_ABL & 8)) {
            store_chunk(min(c + 1, nchunks - 1), buf ^ 1);
            __syncthreads();
        }
    }
    if (PRCNN_ABL & 1) {        // keep the accumulators alive with a store that (practically) never happens
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int n = 0; n < WNB; n++)
#pragma unroll
                for (int e = 0; e < 16; e++) sum += acc[r][n][e];
        if (sum == 123.456f) P.out[tid] = sum;
        return;
    }
    layer_epilogue<MODE, WNB, ADDY>(P, acc, &As[0][0], nullptr, tid, row0, nb0, n_active, addy_done);
}

// =====================================================================================================
// Split-bf16 plain layer (the DEFAULT arithmetic of the plain-row layers since round 4, TERMS = 6; PRCNN_MLP_SPLIT=0 selects the
// fp32-MFMA kernels above, TERMS = 3 is an explicitly requested variant outside the 1e-5 contract): every fp32 operand is cut into three
// bf16 pieces x = x0 + x1 + x2 -- EXACTLY: x0 = the top 16 bits of x, x1 = the top 16 bits of x - x0, x2 = x - x0 - x1, which
// has at most 8 significant bits left -- and the product is rebuilt from bf16 MFMAs (v_mfma_f32_32x32x16_bf16, fp32
// accumulate, 16x the fp32-MFMA rate):
//   TERMS = 6:  x0w0 + x0w1 + x1w0 + x1w1 + x0w2 + x2w0   (drops terms below 2^-24 of |x||w|: fp32-grade results)
//   TERMS = 3:  x0w0 + x0w1 + x1w0                        (drops terms below 2^-16 of |x||w|)
// Six bf16 MFMAs per 16-wide k-step cost 192 matrix-pipe cycles where the eight fp32 MFMAs of the same 16 k cost 512.
// The A rows are split ONCE, on their way from HBM into LDS (three bf16 planes per buffer, 80-byte rows: the 16 rows of a
// ds_read_b128 lane group land on 16 distinct 16-byte bank slots); the weights come pre-split (prcnn_pack_weight_split: per
// (32-column block, 16-wide k-step, piece) the 64 lanes' 16-byte B operands, contiguous) and stream from L2 through a two-step
// register ring, as in mlp_layer_b_kernel.  Accumulator layout = mlp_layer_b_kernel's: the epilogue is shared.
// Non-finite inputs: the split of an infinity is (inf, NaN, NaN) and a NaN piece poisons every product of its row, where the fp32
// kernels return what IEEE arithmetic returns (inf * w = +-inf, NaN only from inf - inf, inf * 0 or a NaN operand).  So a wave whose
// accumulators hold a non-finite value after the main loop (one fma per accumulator register to find out) recomputes its 64-row x
// 32 WNB-column block with fp32 MFMAs straight from global memory -- split_redo_f32, mlp_layer_b_kernel's k order, no LDS, no
// barrier: the block then holds exactly the fp32 kernel's bits.  Finite products that overflow take the same path (same result).
// Shapes: K a multiple of 32, 16-byte aligned rows.
// =====================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#ifndef SPL_VALU_PER_MFMA
#define SPL_VALU_PER_MFMA 5                    // split / address instructions placed behind each MFMA of a chunk's second k-step
#endif
#define SPL_LDB 80                               // bytes per LDS row of one piece plane: 32 bf16 + 16 bytes of padding
#define SPL_PLANE (MLP_BM * SPL_LDB)

__device__ __forceinline__ void split3(float x, uint32_t& b0, uint32_t& b1, uint32_t& b2) {
    b0 = __float_as_uint(x);
    const float r1 = x - __uint_as_float(b0 & 0xFFFF0000u);        // exact: <= 16 significant bits
    b1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(b1 & 0xFFFF0000u);       // exact: <= 8 significant bits, i.e. a bf16 value
    b2 = __float_as_uint(r2);
}
// the top halves of two fp32 bit patterns as a bf16 pair (element 0 in the low half)
__device__ __forceinline__ uint32_t bf16_pair(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// chain = 0: the layer kernel's image, [column block][k-step][piece][lane], lane (h, j) element e <-> k = 16 ks + 8 h + e
// chain = 1: the chain kernel's image, [k-step][output block][piece][lane], element e <-> k = 16 ks + 8 (e >> 2) + 4 h + (e & 3)
__global__ void pack_weight_split_kernel(const float* __restrict__ w, int Nout, int K, int KS, int NB, int chain, uint4* __restrict__ img) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;            // one thread per (column block, k-step, lane)
    if (e >= (long)NB * KS * 64) return;
    const int lane = (int)(e & 63);
    long blk = e >> 6;
    const int ks = (int)(blk % KS), nb = (int)(blk / KS);
    const int n = nb * 32 + (lane & 31), hh = lane >> 5;
    uint32_t b[3][8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int k = chain ? ks * 16 + 8 * (i >> 2) + 4 * hh + (i & 3) : ks * 16 + 8 * hh + i;
        const float v = (n < Nout && k < K) ? w[(long)n * K + k] : 0.f;
        split3(v, b[0][i], b[1][i], b[2][i]);
    }
    if (chain) blk = (long)ks * NB + nb;
#pragma unroll
    for (int p = 0; p < 3; p++)
        img[(blk * 3 + p) * 64 + lane] = make_uint4(bf16_pair(b[p][0], b[p][1]), bf16_pair(b[p][2], b[p][3]),
                                                    bf16_pair(b[p][4], b[p][5]), bf16_pair(b[p][6], b[p][7]));
}

// true (wave-uniform) when any of the wave's accumulator registers is not finite: x * 0 is NaN for x = +-inf / NaN, 0 otherwise
template <int NR>
__device__ __forceinline__ bool wave_has_nonfinite(const f32x16 (&acc)[NR]) {
    float chk = 0.f;
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
        for (int e = 0; e < 16; e++) chk = __builtin_fmaf(acc[r][e], 0.f, chk);
    return __builtin_amdgcn_ballot_w64(chk != chk) != 0;
}

// the wave's 64 x (32 WNB) block again, with fp32 MFMAs and operands read straight from global memory (rare path: a non-finite
// operand).  Same products in the same order as mlp_layer_b_kernel: k-block by k-block, the four k-pairs of a block one after the
// other -- bit-identical to the fp32 layer kernel.
template <int MODE, int WNB>
__device__ __forceinline__ void split_redo_f32(const MlpParams& P, f32x16 (&acc)[2][WNB], long row0, int wm, int wn, int nb0, int lane) {
    const int h = lane >> 5, j = lane & 31;
    const float* ar[2];
    const float* bp[WNB];
    float gd[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // MODE_GROUP: the rows' dxyz (hoisted form: row = relu(Z[idx] + wx . dxyz + b))
#pragma unroll
    for (int r = 0; r < 2; r++) {
        long g = row0 + wm * 64 + r * 32 + j;
        if (g >= P.rows) g = P.rows - 1;                 // clamped, never stored
        if constexpr (MODE == MODE_GROUP) {
            RowMeta<MODE_GROUP> gm;
            make_meta<MODE_GROUP>(P, g, gm);
            ar[r] = P.feat + gm.off + 4 * h;
            gd[r][0] = gm.dx; gd[r][1] = gm.dy; gd[r][2] = gm.dz;
        } else {
            ar[r] = P.in + g * P.ld_in + 4 * h;
        }
    }
#pragma unroll
    for (int n = 0; n < WNB; n++) bp[n] = P.wpack + ((long)min(nb0 + wn * WNB + n, P.NB - 1) * P.KB) * 256 + lane * 4;
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int n = 0; n < WNB; n++) acc[r][n] = (f32x16){0};
    for (int kb = 0; kb < P.KB; kb++) {
        float4 a[2], b[WNB];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            a[r] = ld4(ar[r] + kb * 8);
            if constexpr (MODE == MODE_GROUP) a[r] = act_group4(P, a[r], kb * 8 + 4 * h, gd[r][0], gd[r][1], gd[r][2]);
        }
#pragma unroll
        for (int n = 0; n < WNB; n++) b[n] = ld4(bp[n] + (long)kb * 256);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].x, b[n].x, acc[r][n], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].y, b[n].y, acc[r][n], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].z, b[n].z, acc[r][n], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].w, b[n].w, acc[r][n], 0, 0, 0);
    }
}

// One 128-row tile of the split layer: the body of mlp_layer_s_kernel.  (bx, by) = the workgroup's grid position, or, in the
// bounded-grid form, the position the tile loop stands in for; tid = the thread's index, passed in so that the tile loop can hand in
// an opaque copy per tile (see mlp_layer_s_kernel).
// MODE_GROUP (round 5): the HOISTED grouped first layer -- the A row is relu(Z[idx] + act_wx . dxyz + act_bias)
// with Z = W_f . feat per source point (K = C, a multiple of 32, 16-byte rows): the gathered chunk is activated on its way into the
// split, everything else is the plain kernel (the RCNN stage's three grouped layers ran on the fp32 pipe until then).
template <int MODE, int WNB, int TERMS, bool ADDY>
__device__ __forceinline__ void layer_s_tile(const MlpParams& P, const long bx, const int by, const int tid, unsigned char* __restrict__ Ls) {
    constexpr int QN = 2 * WNB;
    constexpr int NP = TERMS == 6 ? 3 : 2;               // pieces in use
    long tile_id;
    int nb0;
    if (P.wgm_cols > 0) {
        const long bid = bx, xcd = bid & 7, slot = bid >> 3;
        const long rl = slot / P.wgm_cols;
        tile_id = rl * 8 + xcd;
        nb0 = (int)(slot - rl * P.wgm_cols) * QN;
    } else {
        tile_id = tile_of_block(P, bx);
        nb0 = by * QN;
    }
    if (tile_id * MLP_BM >= P.rows) return;
    if (tile_dead(P, tile_id * MLP_BM)) return;

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, j = lane & 31;
    const long row0 = tile_id * MLP_BM;
    const int nchunks = P.K / MLP_BK, KS = P.K / 16;

    static_assert(MODE == MODE_PLAIN || (MODE == MODE_GROUP && !ADDY), "split layer kernel: plain rows or the hoisted grouped form");
    const int c4 = tid & 7, r0 = tid >> 3;
    const float* arow[4];
    float gdx[4], gdy[4], gdz[4];                        // MODE_GROUP: the rows' dxyz
#pragma unroll
    for (int u = 0; u < 4; u++) {
        long grow = row0 + r0 + 32 * u;
        if (grow >= P.rows) grow = P.rows - 1;           // clamped, never stored
        if constexpr (MODE == MODE_GROUP) {
            RowMeta<MODE_GROUP> gm;
            make_meta<MODE_GROUP>(P, grow, gm);
            arow[u] = P.feat + gm.off + c4 * 4;
            gdx[u] = gm.dx; gdy[u] = gm.dy; gdz[u] = gm.dz;
        } else {
            arow[u] = P.in + grow * P.ld_in + c4 * 4;
            gdx[u] = gdy[u] = gdz[u] = 0.f;
        }
    }
    float4 raA[4], raB[4];                               // two chunks of A rows in flight (see the main loop)
    auto load_chunk = [&](int c, float4 (&ra)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) ra[u] = ld4(arow[u] + c * MLP_BK);
    };
    auto store_chunk = [&](int buf, const float4 (&ra)[4], int c) {          // c: the chunk the values belong to (MODE_GROUP)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float4 v = ra[u];
            if constexpr (MODE == MODE_GROUP) v = act_group4(P, v, c * MLP_BK + c4 * 4, gdx[u], gdy[u], gdz[u]);
            uint32_t b[3][4];
            split3(v.x, b[0][0], b[1][0], b[2][0]);
            split3(v.y, b[0][1], b[1][1], b[2][1]);
            split3(v.z, b[0][2], b[1][2], b[2][2]);
            split3(v.w, b[0][3], b[1][3], b[2][3]);
#pragma unroll
            for (int p = 0; p < NP; p++)
                *reinterpret_cast<uint2*>(Ls + (buf * 3 + p) * SPL_PLANE + (r0 + 32 * u) * SPL_LDB + c4 * 8) =
                    make_uint2(bf16_pair(b[p][0], b[p][1]), bf16_pair(b[p][2], b[p][3]));
        }
    };

    f32x16 acc[2][WNB];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int n = 0; n < WNB; n++) acc[r][n] = (f32x16){0};
    const bool n_active = (nb0 + wn * WNB) < P.NB;      // (a wave without a column block computes the last block again, unstored)

    // B ring: slot (g & 1) holds this wave's operands of k-step g, requested two steps (one chunk) ahead; unconditional loads
    // (column blocks past the width re-read the last block, k-steps past K the last step -- neither result is stored / reached)
    uint4 bq[2][WNB][NP];
    const uint4* bptr[WNB];
#pragma unroll
    for (int n = 0; n < WNB; n++) bptr[n] = reinterpret_cast<const uint4*>(P.wsplit) + ((long)min(nb0 + wn * WNB + n, P.NB - 1) * KS) * 192 + lane;
    auto load_b = [&](int g, int slot) {
        const long off = (long)min(g, KS - 1) * 192;
#pragma unroll
        for (int n = 0; n < WNB; n++)
#pragma unroll
            for (int p = 0; p < NP; p++) bq[slot][n][p] = bptr[n][off + p * 64];
    };
    bf16x8 a0[2][NP], a1[2][NP];                          // the A operands of the chunk's two k-steps
    auto read_a = [&](int buf, int st, bf16x8 (&a)[2][NP]) {
        const unsigned char* a_base = Ls + buf * 3 * SPL_PLANE + (wm * 64 + j) * SPL_LDB + h * 16 + st * 32;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int p = 0; p < NP; p++)
                a[r][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_base + p * SPL_PLANE + r * 32 * SPL_LDB));
    };
#define SPL_TERM(A, SL, PA, PB)                                                                                                   \
    _Pragma("unroll") for (int r = 0; r < 2; r++) _Pragma("unroll") for (int n = 0; n < WNB; n++)                                  \
        acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[r][PA], __builtin_bit_cast(bf16x8, bq[SL][n][PB]), acc[r][n], 0, 0, 0)
#define SPL_STEP(A, SL)                      /* smallest terms first */                                                           \
    do {                                                                                                                          \
        if constexpr (TERMS == 6) { SPL_TERM(A, SL, 0, 2); SPL_TERM(A, SL, 2, 0); SPL_TERM(A, SL, 1, 1); }                         \
        SPL_TERM(A, SL, 0, 1); SPL_TERM(A, SL, 1, 0); SPL_TERM(A, SL, 0, 0);                                                       \
    } while (0)
    constexpr int NM = 2 * WNB * TERMS;                   // MFMAs per k-step
    load_b(0, 0);
    load_b(1, 1);
    load_chunk(0, raA);
    load_chunk(min(1, nchunks - 1), raB);
    // Hoisted-FP addend, built FIRST (as mlp_layer_b_kernel does): the interpolated tile -- per 16-byte piece three gathers, ~400
    // instructions per thread -- goes straight into the accumulators through the still idle operand LDS while the B ring and the first
    // two A chunks requested above are in flight; in the epilogue its gathers were exposed behind a main loop of K / 32 chunks (three
    // for FP1's 96-wide skip features).  PRCNN_ADDY_PHASE=0 keeps it in the epilogue (A/B switch; last-bit different summation order).
    bool addy_done = false;
    if constexpr (ADDY) {
        const bool stageable = P.addY && P.addy_phase != 0 && (P.Nout % 4 == 0) && (P.ldY % 4 == 0) && aligned16(P.addY);
        if (stageable) {                                     // workgroup-uniform
            float* T = reinterpret_cast<float*>(Ls);
#pragma unroll
            for (int pass = 0; pass < WNB; pass++) {
                if (pass > 0) __syncthreads();
                addy_stage_half(P, tid, row0, nb0, pass, T, 72);
                __syncthreads();
#pragma unroll
                for (int nn = 0; nn < WNB; nn++) {
                    const int blk = wn * WNB + nn;
                    if ((blk >> 1) != pass) continue;
                    const int tc = (blk & 1) * 32 + j;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
                        acc[0][nn][r] = T[(wm * 64 + rin) * 72 + tc];
                        acc[1][nn][r] = T[(wm * 64 + 32 + rin) * 72 + tc];
                    }
                }
            }
            __syncthreads();
            addy_done = true;
        }
    }
    store_chunk(0, raA, 0);
    __syncthreads();
    read_a(0, 0, a0);
    if (PRCNN_ABL & 256) read_a(0, 1, a1);
    // One chunk = two k-steps.  The matrix pipe takes a bf16 MFMA every 32 cycles; a wave's other instructions issue in the gaps.
    // Step 0 carries the LDS reads of step 1 and the global loads (A rows TWO chunks ahead -- a chunk is ~1500 matrix-pipe cycles,
    // less than an HBM round trip under load -- and B two steps ahead); step 1 carries the split of the next chunk and the LDS
    // writes into the idle buffer.  Left to itself the compiler issues each step's MFMAs as one run and the ~100 split
    // instructions after it, with the pipe idle.
    auto chunk_body = [&](int c, const float4 (&ra_split)[4], float4 (&ra_load)[4]) {
        const int buf = c & 1;
        if (!(PRCNN_ABL & 64)) load_chunk(min(c + 2, nchunks - 1), ra_load);
        if (!(PRCNN_ABL & 256)) read_a(buf, 1, a1);
        if (!(PRCNN_ABL & 128)) SPL_STEP(a0, 0);
        if (!(PRCNN_ABL & 32)) load_b(c * 2 + 2, 0);
#pragma unroll
        for (int q = 0; q < 2 * NP; q++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int q = 0; q < 4 + WNB * NP; q++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(PRCNN_ABL & 64)) store_chunk(buf ^ 1, ra_split, min(c + 1, nchunks - 1));
        if (!(PRCNN_ABL & 128)) SPL_STEP(a1, 1);
        if (!(PRCNN_ABL & 32)) load_b(c * 2 + 3, 1);
#pragma unroll
        for (int q = 0; q < NM; q++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, SPL_VALU_PER_MFMA, 0);
            if (q % 2 == 1 && q / 2 < 4 * NP) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(PRCNN_ABL & 256)) {
            __syncthreads();
            read_a(buf ^ 1, 0, a0);
        }
    };
    for (int c = 0; c < nchunks; c += 2) {
        chunk_body(c, raB, raA);                          // raA held chunk c (already in LDS): it takes chunk c + 2
        if (c + 1 < nchunks) chunk_body(c + 1, raA, raB);
    }
#undef SPL_STEP
#undef SPL_TERM
    if (PRCNN_ABL & (16 | 128)) {           // ablation builds only: keep the operands / accumulators alive with a store that never happens
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int n = 0; n < WNB; n++)
#pragma unroll
                for (int e = 0; e < 16; e++) sum += acc[r][n][e];
        if (PRCNN_ABL & 128) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int p = 0; p < NP; p++) sum += (float)a0[r][p][0] + (float)a1[r][p][3] + __uint_as_float(bq[0][0][p].x) + __uint_as_float(bq[1][WNB - 1][p].w);
        }
        if (sum == 123.456f) P.out[tid] = sum;
        if (PRCNN_ABL & 16) return;
    }
    if (!(PRCNN_ABL & (16 | 128))) {
        // non-finite operands (see the header): this wave's block again on the fp32 pipe
        // (workgroup-uniform when the addend sits in the accumulators: the epilogue then stages it again for all four waves; a
        //  non-finite ADDEND takes this path too and comes out as the fp32 kernel's sum)
        bool bad = wave_has_nonfinite<2 * WNB>(reinterpret_cast<const f32x16 (&)[2 * WNB]>(acc));
        if (ADDY && addy_done) bad = __syncthreads_or(bad) != 0;
        if (bad && n_active) split_redo_f32<MODE, WNB>(P, acc, row0, wm, wn, nb0, lane);
        if (bad) addy_done = false;
    }
    layer_epilogue<MODE, WNB, ADDY>(P, acc, reinterpret_cast<float*>(Ls), nullptr, tid, row0, nb0, n_active, addy_done);
}

// LOOP = false: one workgroup per tile (the grid covers the launch's row count).
// LOOP = true (round 5): the BOUNDED-GRID form for launches sized for the capacity of a compacted list (device-side row count: the
// RCNN stage's lists hold 0.4-0.6 M live rows of 6.5-26 M): a fixed number of workgroups walk the LIVE tiles, where the one-tile form
// retired one empty workgroup per 128 rows of CAPACITY (90-100 us of pure dispatch per launch, seven launches per two-stage step).
// A plain loop around the body cost 20-48 VGPRs (round 4: the optimiser keeps every lane-derived address of the body live across
// the back edge); here each trip derives everything from an OPAQUE copy of the thread index (a volatile v_mov the optimiser cannot
// hoist or merge), so nothing but the loop counter lives across tiles and the body is allocated as in the one-tile form.
template <int MODE, int WNB, int TERMS, bool ADDY, bool LOOP>
__global__ __launch_bounds__(MLP_THREADS, 2) void mlp_layer_s_kernel(const MlpParams Pin) {
    MlpParams P = Pin;
    P.rows = effective_rows(Pin);
    __shared__ __attribute__((aligned(16))) unsigned char Ls[2 * 3 * SPL_PLANE];
    if constexpr (LOOP) {
        const long nbid = ((P.rows + MLP_BM - 1) / MLP_BM + 7) / 8 * 8 * P.wgm_cols;       // 1-D XCD-aware order only (launch_mlp)
        for (long bid = blockIdx.x; bid < nbid; bid += gridDim.x) {
            int tid;
            asm volatile("v_mov_b32 %0, %1" : "=v"(tid) : "v"(threadIdx.x));
            layer_s_tile<MODE, WNB, TERMS, ADDY>(P, bid, 0, tid, Ls);
            __syncthreads();                             // the next tile's LDS writes vs this tile's last reads
        }
    } else {
        layer_s_tile<MODE, WNB, TERMS, ADDY>(P, blockIdx.x, blockIdx.y, threadIdx.x, Ls);
    }
}

// =====================================================================================================
// Register-resident layer CHAIN (whole SharedMLP in one kernel, one wave = 32 rows, activations never in LDS/HBM).
//
// The layer is computed transposed: out^T[n][row] = sum_k W[n][k] * act^T[k][row], i.e. the MFMA A operand is
// the packed weight (lane (h,i) -> W[32*ob+i][8*kb+4*h+s], exactly the prcnn_pack_weight image) and the B
// operand is the activation of the lane's own row (lane (h,j) -> act[row j][8*kb+4*h+s]).  The 32x32 MFMA
// returns D[i][j] with lane (h,j) holding i = (r&3) + 8*(r>>2) + 4*h in register r = 4*q+s, i.e. channel
// 32*ob + 8*q + 4*h + s of row j -- which IS the B operand element (k-block 4*ob+q, step s) of the next
// layer.  So accumulators feed the next layer's MFMAs directly: the chain never leaves the register file.
// Bias+ReLU are applied in place; the max-pool over nsample rows is a DPP max over the 16-lane rows of the
// row (= lane) axis.  Intermediate activations never touch LDS or HBM, and an interpolated / gathered input
// row is built exactly once (the per-layer kernel rebuilds it for every 64-column tile).
// Limits: every layer width <= 128 (4 blocks of 32), pool_ns in {0,16,32}; wider layers use mlp_layer_kernel.
// =====================================================================================================
struct ChainParams {
    MlpParams a;               // prologue description + layer 0 (wpack/bias/K/KB/Nout/relu) + output/pool
    const float* wpack1; const float* bias1; int KB1, N1, relu1;
    const float* wpack2; const float* bias2; int KB2, N2, relu2;
    int nlayers;
    int stack_split;           // mlp_stack2_kernel: most workgroups per row tile (set by dispatch_chain)
    const void* wsplit1;       // mlp_chain_s_kernel: layer 1's split image (layer 0's is a.wsplit)
};

#define CH_DPP_FMAX(v, ctrl) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl : "+v"(v))
#define CH_DPP_FMAX4(v, ctrl)                                                                                   \
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl               \
                 "\n\tv_max_f32_dpp %2, %2, %2 " ctrl "\n\tv_max_f32_dpp %3, %3, %3 " ctrl                       \
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]))

__device__ __forceinline__ float4 ldw(const float* __restrict__ wpack, int KB, int ob, int kb, int lane) {
    return ld4(wpack + (((long)ob * KB + kb) * 64 + lane) * 4);
}

// bias (padded to 32*NB by the host) + optional ReLU, in the D-register layout
template <int NB>
__device__ __forceinline__ void bias_act(f32x16 (&acc)[NB], const float* __restrict__ bias, int relu, int h) {
#pragma unroll
    for (int ob = 0; ob < NB; ob++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 b = bias ? ld4(bias + ob * 32 + 8 * q + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
            float v0 = acc[ob][4 * q + 0] + b.x, v1 = acc[ob][4 * q + 1] + b.y;
            float v2 = acc[ob][4 * q + 2] + b.z, v3 = acc[ob][4 * q + 3] + b.w;
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            acc[ob][4 * q + 0] = v0; acc[ob][4 * q + 1] = v1; acc[ob][4 * q + 2] = v2; acc[ob][4 * q + 3] = v3;
        }
}

// ---- weight staging: the 4 waves of a workgroup walk the same (layer, k-block) sequence, so each packed weight
// tile (1 KB = one (ob,kb) MFMA A-operand for 64 lanes) is fetched from L2 ONCE per workgroup into LDS and read by
// all waves with conflict-free ds_read_b128 (lane-linear image).  A stage = G k-blocks x NB output blocks
// (<= 16 tiles = 16 KB), double buffered: one s_barrier per stage (~64 MFMAs per wave).
#define CH_WAVES 4
#define CH_STAGE_TILES 16
template <int NB> struct ChStage { static constexpr int G = CH_STAGE_TILES / NB; static constexpr int TPW = (G * NB + CH_WAVES - 1) / CH_WAVES; };

template <int NB>
__device__ __forceinline__ void stage_load(const float* __restrict__ wpack, int KB, int st, int wave, int lane, float4 (&r)[4]) {
    constexpr int G = ChStage<NB>::G;
#pragma unroll
    for (int u = 0; u < ChStage<NB>::TPW; u++) {
        int tt = wave + CH_WAVES * u, kbl = tt / NB, ob = tt - kbl * NB, kb = st * G + kbl;
        r[u] = (tt < G * NB && kb < KB) ? ldw(wpack, KB, ob, kb, lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int NB>
__device__ __forceinline__ void stage_store(float* ws, int wave, int lane, const float4 (&r)[4]) {
#pragma unroll
    for (int u = 0; u < ChStage<NB>::TPW; u++) {
        int tt = wave + CH_WAVES * u;
        if (tt < ChStage<NB>::G * NB) *reinterpret_cast<float4*>(ws + (tt * 64 + lane) * 4) = r[u];
    }
}
__device__ __forceinline__ float4 lds_w(const float* ws, int tile, int lane) {
    return *reinterpret_cast<const float4*>(ws + (tile * 64 + lane) * 4);
}

// next layer from register-resident activations: out[ob] += W[ob, kb] * in[kb/4][4*(kb%4)+s].
// Fully unrolled (register indices must be compile-time); every branch below is workgroup-uniform.
template <int NBI, int NBO>
__device__ __forceinline__ void chain_layer(const f32x16 (&in)[NBI], f32x16 (&out)[NBO], const float* __restrict__ wpack,
                                            int KB, float (*Ws)[CH_STAGE_TILES * 256], int wave, int lane) {
    constexpr int G = ChStage<NBO>::G;
    constexpr int NST = (NBI * 4 + G - 1) / G;
#pragma unroll
    for (int ob = 0; ob < NBO; ob++) out[ob] = (f32x16){0};
    float4 wr[4];
    stage_load<NBO>(wpack, KB, 0, wave, lane, wr);
    stage_store<NBO>(Ws[0], wave, lane, wr);
    __syncthreads();
#pragma unroll
    for (int st = 0; st < NST; st++) {
        if (st * G < KB) {
            const bool more = (st + 1) * G < KB;
            if (more) stage_load<NBO>(wpack, KB, st + 1, wave, lane, wr);
            const float* ws = Ws[st & 1];
#pragma unroll
            for (int kbl = 0; kbl < G; kbl++) {
                const int kb = st * G + kbl;
                if (kb < NBI * 4 && kb < KB) {
                    const int pb = kb < NBI * 4 ? kb / 4 : 0, q = kb % 4;
                    float4 w[NBO];
#pragma unroll
                    for (int ob = 0; ob < NBO; ob++) w[ob] = lds_w(ws, kbl * NBO + ob, lane);
                    // k-step outermost: consecutive MFMAs hit different accumulators
#pragma unroll
                    for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].x, in[pb][4 * q + 0], out[ob], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].y, in[pb][4 * q + 1], out[ob], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].z, in[pb][4 * q + 2], out[ob], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].w, in[pb][4 * q + 3], out[ob], 0, 0, 0);
                }
            }
            if (more) stage_store<NBO>(Ws[(st + 1) & 1], wave, lane, wr);
            __syncthreads();
        }
    }
}

// epilogue of the last layer: (activated) acc -> out, plain store or max over nsample rows (= lanes)
template <int NB>
__device__ __forceinline__ void chain_store(const ChainParams& C, f32x16 (&acc)[NB], int Nlast, long row, bool valid,
                                            int lane, int h) {
    const MlpParams& P = C.a;
    const bool vec = ((P.ld_out | P.col_off) % 4 == 0) && aligned16(P.out);
    if (P.pool_ns == 0) {
        if (!valid) return;
        float* o = P.out + row * P.ld_out + P.col_off;
#pragma unroll
        for (int ob = 0; ob < NB; ob++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int c = ob * 32 + 8 * q + 4 * h;
                if (c >= Nlast) continue;
                if (vec && c + 4 <= Nlast) {
                    *reinterpret_cast<float4*>(o + c) = make_float4(acc[ob][4 * q], acc[ob][4 * q + 1], acc[ob][4 * q + 2], acc[ob][4 * q + 3]);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; s++)
                        if (c + s < Nlast) o[c + s] = acc[ob][4 * q + s];
                }
            }
        return;
    }
    // max over the 16-lane DPP rows of the row axis (lanes 0-15 / 16-31 of each half are 16 consecutive rows);
    // ReLU'd values are >= 0 and rows past the end hold 0 contributions only in groups that are not stored
    const int j = lane & 31;
    const bool writer = P.pool_ns == 16 ? ((j & 15) == 15) : (j == 31);
    const long group = P.pool_ns == 16 ? (row / 16) : (row / 32);
    const long groups = P.rows / P.pool_ns;
    float* o = P.out + group * P.ld_out + P.col_off;
#pragma unroll
    for (int ob = 0; ob < NB; ob++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            // the four registers of a (block, q) group walk the DPP steps in lock-step: each value's next step is four
            // VALU instructions after its previous one, so the DPP read-after-write hazard needs no s_nop per step
            float v[4] = {acc[ob][4 * q], acc[ob][4 * q + 1], acc[ob][4 * q + 2], acc[ob][4 * q + 3]};
            CH_DPP_FMAX4(v, "row_shr:1 row_mask:0xf bank_mask:0xf");
            CH_DPP_FMAX4(v, "row_shr:2 row_mask:0xf bank_mask:0xf");
            CH_DPP_FMAX4(v, "row_shr:4 row_mask:0xf bank_mask:0xf");
            CH_DPP_FMAX4(v, "row_shr:8 row_mask:0xf bank_mask:0xf");
            if (P.pool_ns == 32) CH_DPP_FMAX4(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
            asm volatile("s_nop 1");
            int c = ob * 32 + 8 * q + 4 * h;
            if (writer && valid && group < groups && c < Nlast) {
                if (vec && c + 4 <= Nlast) *reinterpret_cast<float4*>(o + c) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int s = 0; s < 4; s++)
                        if (c + s < Nlast) o[c + s] = v[s];
                }
            }
        }
}

// ONE output channel after a hidden layer (the classification head, 128 -> 128 -> 1): the last layer as a dot product on the VALU.
// On the matrix pipe the single column occupies a whole 32-column block: 4 NB MFMAs per k-block = 4096 pipe cycles per 32 rows for
// 1/32 of their result -- a fifth of that launch's MFMA time.  The lane already holds its row's 16 NB activations of each half
// (channel 32 ob + 8 q + 4 h + s), so it is 16 NB multiply-adds and one exchange between the halves.  Shared by every chain variant
// (bit-identical among them).  wpack: the pack image of the (1, 32 NB) weight, whose row 0 sits at lane 32 h of k-block 4 ob + q.
template <int NB>
__device__ __forceinline__ void chain_out1(const ChainParams& C, const f32x16 (&a)[NB], long row, bool valid, int lane, int h) {
    const MlpParams& P = C.a;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int ob = 0; ob < NB; ob++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 w = ld4(C.wpack1 + ((4 * ob + q) * 64 + h * 32) * 4);
            s0 = fmaf(a[ob][4 * q + 0], w.x, s0); s1 = fmaf(a[ob][4 * q + 1], w.y, s1);
            s2 = fmaf(a[ob][4 * q + 2], w.z, s2); s3 = fmaf(a[ob][4 * q + 3], w.w, s3);
        }
    float sum = (s0 + s1) + (s2 + s3);
    sum += __shfl_xor(sum, 32);
    sum += C.bias1 ? C.bias1[0] : 0.f;
    if (C.relu1) sum = fmaxf(sum, 0.f);
    if (valid && h == 0) P.out[row * P.ld_out + P.col_off] = sum;
}
template <int NB1, int NB2> __device__ __forceinline__ bool chain_out1_applies(const ChainParams& C) {
    return NB1 == 1 && NB2 == 0 && C.N1 == 1 && C.a.pool_ns == 0;
}

template <int MODE, int NB0, int NB1, int NB2>
__global__ __launch_bounds__(256) void mlp_chain_kernel(const ChainParams Cin) {
    ChainParams C = Cin;
    C.a.rows = effective_rows(Cin.a);
    const MlpParams& P = C.a;
    const long tile_id = tile_of_block(P, blockIdx.x);
    if (tile_id * 128 >= P.rows) return;
    if (tile_dead(P, tile_id * 128)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const long row = (tile_id * 4 + wave) * 32 + j;
    RowMeta<MODE> meta;
    make_meta<MODE>(P, row, meta);

    __shared__ __attribute__((aligned(16))) float Ws[2][CH_STAGE_TILES * 256];

    // ---- layer 0: the B operand is the gathered / interpolated / plain input row, streamed over K ----
    f32x16 a0[NB0];
#pragma unroll
    for (int ob = 0; ob < NB0; ob++) a0[ob] = (f32x16){0};
    {
        constexpr int G = ChStage<NB0>::G;
        const int nst = (P.KB + G - 1) / G;
        float4 wr[4];
        stage_load<NB0>(P.wpack, P.KB, 0, wave, lane, wr);
        Raw<MODE> cur, nxt;
        fetch<MODE>(P, meta, 4 * h, cur);
        stage_store<NB0>(Ws[0], wave, lane, wr);
        __syncthreads();
        for (int st = 0; st < nst; st++) {
            const bool more_st = st + 1 < nst;
            if (more_st) stage_load<NB0>(P.wpack, P.KB, st + 1, wave, lane, wr);
            const float* ws = Ws[st & 1];
            for (int kbl = 0; kbl < G; kbl++) {
                const int kb = st * G + kbl;
                if (kb >= P.KB) break;
                const bool more = kb + 1 < P.KB;
                if (more) fetch<MODE>(P, meta, 8 * (kb + 1) + 4 * h, nxt);
                float4 b = finish<MODE>(P, meta, 8 * kb + 4 * h, cur);
                float4 w[NB0];
#pragma unroll
                for (int ob = 0; ob < NB0; ob++) w[ob] = lds_w(ws, kbl * NB0 + ob, lane);
#pragma unroll
                for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].x, b.x, a0[ob], 0, 0, 0);
#pragma unroll
                for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].y, b.y, a0[ob], 0, 0, 0);
#pragma unroll
                for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].z, b.z, a0[ob], 0, 0, 0);
#pragma unroll
                for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].w, b.w, a0[ob], 0, 0, 0);
                if (more) cur = nxt;
            }
            if (more_st) stage_store<NB0>(Ws[(st + 1) & 1], wave, lane, wr);
            __syncthreads();
        }
    }
    bias_act<NB0>(a0, P.bias, P.relu, h);
    if (NB1 == 0) { chain_store<NB0>(C, a0, P.Nout, row, meta.valid, lane, h); return; }
    if (chain_out1_applies<NB1, NB2>(C)) { chain_out1<NB0>(C, a0, row, meta.valid, lane, h); return; }

    f32x16 a1[NB1 ? NB1 : 1];
    chain_layer<NB0, (NB1 ? NB1 : 1)>(a0, a1, C.wpack1, C.KB1, Ws, wave, lane);
    bias_act<(NB1 ? NB1 : 1)>(a1, C.bias1, C.relu1, h);
    if (NB2 == 0) { chain_store<(NB1 ? NB1 : 1)>(C, a1, C.N1, row, meta.valid, lane, h); return; }

    f32x16 a2[NB2 ? NB2 : 1];
    chain_layer<(NB1 ? NB1 : 1), (NB2 ? NB2 : 1)>(a1, a2, C.wpack2, C.KB2, Ws, wave, lane);
    bias_act<(NB2 ? NB2 : 1)>(a2, C.bias2, C.relu2, h);
    chain_store<(NB2 ? NB2 : 1)>(C, a2, C.N2, row, meta.valid, lane, h);
}

// =====================================================================================================
// FAST chain variant: the same computation as mlp_chain_kernel for the shapes that dominate the RPN graph (hoisted SA
// stacks, the hoisted FP0 layer, the heads), written so that the layer-0 loop body is ONE straight-line block:
//   * rows past the end are clamped to the last row (their results are simply not stored), every K is a multiple
//     of the stage (host-checked), sources are 16-byte aligned: fetch / finish have no branches at all;
//   * the activated-gather parameters (W_x, bias) sit in LDS;
//   * the B operand of k-block kb+1 is produced (VALU) and the raw pieces of kb+2 are requested (VMEM) in the same
//     block as the MFMAs of kb, so the scheduler runs them in the MFMA shadow instead of between MFMA bursts (the
//     generic kernel's per-piece bounds checks split the loop into dozens of basic blocks and serialise the three);
//   * the first weight stage of layer l+1 is requested while layer l computes.
// Results are bit-identical to the generic kernel (same MFMA order, same activation arithmetic).
// =====================================================================================================
template <int NB> struct FStage { static constexpr int G = NB == 3 ? 4 : CH_STAGE_TILES / NB; static constexpr int TPW = (G * NB + CH_WAVES - 1) / CH_WAVES; };

template <int NB>
__device__ __forceinline__ void fstage_load(const float* __restrict__ wpack, int KB, int st, int wave, int lane, float4 (&r)[4]) {
    constexpr int G = FStage<NB>::G;
#pragma unroll
    for (int u = 0; u < FStage<NB>::TPW; u++) {
        const int tt = wave + CH_WAVES * u, kbl = tt / NB, ob = tt - kbl * NB;
        if (tt < G * NB) r[u] = ldw(wpack, KB, ob, st * G + kbl, lane);            // compile-time condition after unrolling
    }
}
template <int NB>
__device__ __forceinline__ void fstage_store(float* ws, int wave, int lane, const float4 (&r)[4]) {
#pragma unroll
    for (int u = 0; u < FStage<NB>::TPW; u++) {
        const int tt = wave + CH_WAVES * u;
        if (tt < FStage<NB>::G * NB) *reinterpret_cast<float4*>(ws + (tt * 64 + lane) * 4) = r[u];
    }
}

template <int MODE> __device__ __forceinline__ void fast_fetch(const MlpParams& P, const RowMeta<MODE>& r, int k, Raw<MODE>& v);
template <> __device__ __forceinline__ void fast_fetch<MODE_PLAIN>(const MlpParams& P, const RowMeta<MODE_PLAIN>& r, int k, Raw<MODE_PLAIN>& v) {
    v.a = ld4(P.in + r.off + k);
}
template <> __device__ __forceinline__ void fast_fetch<MODE_GROUP>(const MlpParams& P, const RowMeta<MODE_GROUP>& r, int k, Raw<MODE_GROUP>& v) {
    v.a = ld4(P.feat + r.off + k);
}
template <> __device__ __forceinline__ void fast_fetch<MODE_INTERP>(const MlpParams& P, const RowMeta<MODE_INTERP>& r, int k, Raw<MODE_INTERP>& v) {
    v.a = ld4(P.known + r.o0 + k); v.b = ld4(P.known + r.o1 + k); v.c = ld4(P.known + r.o2 + k);
}
// s_wx: act_wx (K,3) copied to LDS; s_b: act_bias (K)
template <int MODE> __device__ __forceinline__ float4 fast_finish(const RowMeta<MODE>& r, int k, const Raw<MODE>& v, const float* s_wx, const float* s_b);
template <> __device__ __forceinline__ float4 fast_finish<MODE_PLAIN>(const RowMeta<MODE_PLAIN>&, int, const Raw<MODE_PLAIN>& v, const float*, const float*) {
    return v.a;
}
template <> __device__ __forceinline__ float4 fast_finish<MODE_GROUP>(const RowMeta<MODE_GROUP>& r, int k, const Raw<MODE_GROUP>& v, const float* s_wx, const float* s_b) {
    const float4 w0 = ld4(s_wx + k * 3), w1 = ld4(s_wx + k * 3 + 4), w2 = ld4(s_wx + k * 3 + 8), b = ld4(s_b + k);
    float4 o;                                        // same expression tree as act_group4
    o.x = fmaxf(v.a.x + (w0.x * r.dx + w0.y * r.dy + w0.z * r.dz) + b.x, 0.f);
    o.y = fmaxf(v.a.y + (w0.w * r.dx + w1.x * r.dy + w1.y * r.dz) + b.y, 0.f);
    o.z = fmaxf(v.a.z + (w1.z * r.dx + w1.w * r.dy + w2.x * r.dz) + b.z, 0.f);
    o.w = fmaxf(v.a.w + (w2.y * r.dx + w2.z * r.dy + w2.w * r.dz) + b.w, 0.f);
    return o;
}
template <> __device__ __forceinline__ float4 fast_finish<MODE_INTERP>(const RowMeta<MODE_INTERP>& r, int k, const Raw<MODE_INTERP>& v, const float*, const float* s_b) {
    const float4 b = ld4(s_b + k);
    float4 o;
    o.x = fmaxf(interp1(r.w0, v.a.x, r.w1, v.b.x, r.w2, v.c.x) + b.x, 0.f);
    o.y = fmaxf(interp1(r.w0, v.a.y, r.w1, v.b.y, r.w2, v.c.y) + b.y, 0.f);
    o.z = fmaxf(interp1(r.w0, v.a.z, r.w1, v.b.z, r.w2, v.c.z) + b.z, 0.f);
    o.w = fmaxf(interp1(r.w0, v.a.w, r.w1, v.b.w, r.w2, v.c.w) + b.w, 0.f);
    return o;
}

// layers >= 1 of the fast chain: KB == 4*NBI exactly (host-checked), first stage pre-loaded by the caller
template <int NBI, int NBO, int NBN>
__device__ __forceinline__ void fchain_layer(const f32x16 (&in)[NBI], f32x16 (&out)[NBO], const float* __restrict__ wpack,
                                             float (*Ws)[CH_STAGE_TILES * 256], int wave, int lane, float4 (&wr)[4],
                                             const float* __restrict__ wpack_next, float4 (&wnext)[4]) {
    constexpr int G = FStage<NBO>::G, KB = NBI * 4;
    static_assert(KB % G == 0, "stage size must divide the k-blocks");
    constexpr int NST = KB / G;
#pragma unroll
    for (int ob = 0; ob < NBO; ob++) out[ob] = (f32x16){0};
    fstage_store<NBO>(Ws[0], wave, lane, wr);
    __syncthreads();
    if (NBN > 0) fstage_load<(NBN > 0 ? NBN : 1)>(wpack_next, NBO * 4, 0, wave, lane, wnext);
#pragma unroll
    for (int st = 0; st < NST; st++) {
        if (st + 1 < NST) fstage_load<NBO>(wpack, KB, st + 1, wave, lane, wr);
        __builtin_amdgcn_sched_barrier(0);                 // (as in layer 0: keep the weight request ahead of the MFMAs)
        const float* ws = Ws[st & 1];
#pragma unroll
        for (int kbl = 0; kbl < G; kbl++) {
            const int kb = st * G + kbl, pb = kb / 4, q = kb % 4;
            float4 w[NBO];
#pragma unroll
            for (int ob = 0; ob < NBO; ob++) w[ob] = lds_w(ws, kbl * NBO + ob, lane);
#pragma unroll
            for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].x, in[pb][4 * q + 0], out[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].y, in[pb][4 * q + 1], out[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].z, in[pb][4 * q + 2], out[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].w, in[pb][4 * q + 3], out[ob], 0, 0, 0);
        }
        if (st + 1 < NST) fstage_store<NBO>(Ws[(st + 1) & 1], wave, lane, wr);
        __syncthreads();
    }
}

#define FAST_MAX_K 128
#ifndef FAST_RING_PLAIN
#define FAST_RING_PLAIN 3
#endif
#ifndef FAST_RING_GROUP
#define FAST_RING_GROUP 2
#endif
#ifndef FAST_RING_INTERP
#define FAST_RING_INTERP 2
#endif
template <int MODE> struct FAST_RING { static constexpr int D = MODE == MODE_PLAIN ? FAST_RING_PLAIN : (MODE == MODE_GROUP ? FAST_RING_GROUP : FAST_RING_INTERP); };
template <int MODE, int NB0, int NB1, int NB2>
// (the plain-row instances with a narrow second layer -- the heads -- are asked for three waves per SIMD: 164 registers without a spill where the compiler
//  settles at 180 unprompted, -3 % per launch, and a wave still fits next to a resident FPS workgroup; the interpolating
//  instance spills at that budget, 235 -> 277 us, and keeps two)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MODE == MODE_PLAIN && NB1 <= 3) ? 3 : 2, 8))) void mlp_chain_fast_kernel(const ChainParams Cin) {
    ChainParams C = Cin;
    C.a.rows = effective_rows(Cin.a);
    const MlpParams& P = C.a;
    const long tile_id = tile_of_block(P, blockIdx.x);
    if (tile_id * 128 >= P.rows) return;
    if (tile_dead(P, tile_id * 128)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const long row = (tile_id * 4 + wave) * 32 + j;
    const bool valid = row < P.rows;
    RowMeta<MODE> meta;
    make_meta<MODE>(P, valid ? row : P.rows - 1, meta);

    __shared__ __attribute__((aligned(16))) float Ws[2][CH_STAGE_TILES * 256];
    __shared__ __attribute__((aligned(16))) float s_wx[(FAST_MAX_K + 16) * 3];
    __shared__ __attribute__((aligned(16))) float s_b[FAST_MAX_K + 16];
    __shared__ __attribute__((aligned(16))) float s_bias[3][128];      // layer biases: an LDS read at use instead of an L1/L2 trip
    if (MODE == MODE_GROUP)
        for (int t = threadIdx.x; t < P.K * 3; t += 256) s_wx[t] = P.act_wx[t];
    if (MODE != MODE_PLAIN)
        for (int t = threadIdx.x; t < P.K; t += 256) s_b[t] = P.act_bias[t];
    if (threadIdx.x < 128) {
        const int t = threadIdx.x;
        s_bias[0][t] = (P.bias && t < NB0 * 32) ? P.bias[t] : 0.f;
        s_bias[1][t] = (NB1 > 0 && C.bias1 && t < NB1 * 32) ? C.bias1[t] : 0.f;
        s_bias[2][t] = (NB2 > 0 && C.bias2 && t < NB2 * 32) ? C.bias2[t] : 0.f;
    }

    f32x16 a0[NB0];
#pragma unroll
    for (int ob = 0; ob < NB0; ob++) a0[ob] = (f32x16){0};
    float4 w1s[4], w2s[4];
    {
        constexpr int G = FStage<NB0>::G;
        const int nst = P.KB / G;                              // KB % G == 0 (host-checked)
        const int klast = P.K - 8 + 4 * h;                     // last legal piece of this lane's k sub-range
        float4 wr[4];
        fstage_load<NB0>(P.wpack, P.KB, 0, wave, lane, wr);
        // RD pieces ahead: ring[0] = raw row piece of kb+1 (arrived) ... ring[RD-1] = kb+RD; kb+RD+1 is requested in this block.
        // Plain rows stream from HBM (~2 us away under load) and cost 4 registers a piece; gathered rows come from L2 / MALL
        // and cost 4 (group) or 12 (interp) registers a piece.
        constexpr int RD = FAST_RING<MODE>::D;
        Raw<MODE> r0, ring[RD];
        fast_fetch<MODE>(P, meta, 4 * h, r0);
#pragma unroll
        for (int q = 0; q < RD; q++) fast_fetch<MODE>(P, meta, min(8 * (q + 1) + 4 * h, klast), ring[q]);
        fstage_store<NB0>(Ws[0], wave, lane, wr);
        __syncthreads();                                       // Ws[0], s_wx, s_b visible
        if (NB1 > 0) fstage_load<(NB1 ? NB1 : 1)>(C.wpack1, NB0 * 4, 0, wave, lane, w1s);
        float4 bcur = fast_finish<MODE>(meta, 4 * h, r0, s_wx, s_b);
        for (int st = 0; st < nst; st++) {
            const bool more_st = st + 1 < nst;
            if (more_st) fstage_load<NB0>(P.wpack, P.KB, st + 1, wave, lane, wr);
            __builtin_amdgcn_sched_barrier(0);
            const float* ws = Ws[st & 1];
#pragma unroll
            for (int kbl = 0; kbl < G; kbl++) {
                const int kb = st * G + kbl;
                Raw<MODE> rn;
                fast_fetch<MODE>(P, meta, min(8 * (kb + 1 + RD) + 4 * h, klast), rn);
                // keep the request HERE: left alone, the scheduler sinks these loads to just before their first use (one
                // k-block later), which turns the two-piece ring into a load-use stall of a full L2 round trip per k-block
                __builtin_amdgcn_sched_barrier(0);
                float4 w[NB0];
#pragma unroll
                for (int ob = 0; ob < NB0; ob++) w[ob] = lds_w(ws, kbl * NB0 + ob, lane);
#pragma unroll
                for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].x, bcur.x, a0[ob], 0, 0, 0);
#pragma unroll
                for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].y, bcur.y, a0[ob], 0, 0, 0);
#pragma unroll
                for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].z, bcur.z, a0[ob], 0, 0, 0);
#pragma unroll
                for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].w, bcur.w, a0[ob], 0, 0, 0);
                bcur = fast_finish<MODE>(meta, min(8 * (kb + 1) + 4 * h, klast), ring[0], s_wx, s_b);   // B operand of kb+1
#pragma unroll
                for (int q = 0; q + 1 < RD; q++) ring[q] = ring[q + 1];
                ring[RD - 1] = rn;
            }
            if (more_st) fstage_store<NB0>(Ws[(st + 1) & 1], wave, lane, wr);
            __syncthreads();
        }
    }
    bias_act<NB0>(a0, s_bias[0], P.relu, h);
    if (chain_out1_applies<NB1, NB2>(C)) { chain_out1<NB0>(C, a0, row, valid, lane, h); return; }
    if constexpr (NB1 == 0) {
        chain_store<NB0>(C, a0, P.Nout, row, valid, lane, h);
    } else {
        f32x16 a1[NB1];
        fchain_layer<NB0, NB1, NB2>(a0, a1, C.wpack1, Ws, wave, lane, w1s, C.wpack2, w2s);
        bias_act<NB1>(a1, s_bias[1], C.relu1, h);
        if constexpr (NB2 == 0) {
            chain_store<NB1>(C, a1, C.N1, row, valid, lane, h);
        } else {
            f32x16 a2[NB2];
            fchain_layer<NB1, NB2, 0>(a1, a2, C.wpack2, Ws, wave, lane, w2s, nullptr, w1s);
            bias_act<NB2>(a2, s_bias[2], C.relu2, h);
            chain_store<NB2>(C, a2, C.N2, row, valid, lane, h);
        }
    }
}

// =====================================================================================================
// Split-bf16 register-resident chain, plain rows, two layers (the RPN heads: 128 -> 128 -> 76 and 128 -> 128 -> 1) -- the opt-in
// VARIANT of mlp_chain_fast_kernel<MODE_PLAIN, 4, NB1, 0> (see mlp_layer_s_kernel for the arithmetic: exact three-way bf16 split of
// both operands, TERMS bf16 MFMA products per fp32 product, fp32 accumulate).  Same transposed formulation as the fp32 chain: the
// weights are the MFMA A operand, the lane's own row the B operand, the D layout of layer l is the B layout of layer l+1 -- with
// the 16-wide k-steps of the bf16 MFMA a lane half's eight operand elements are accumulator registers 8 sh .. 8 sh + 7 of block
// pb, i.e. k = 32 pb + 16 sh + 8 (e >> 2) + 4 h + (e & 3): the chain's weight image (prcnn_pack_weight_split, chain = 1) is packed
// in that k order, [k-step][output block][piece][lane], so a stage of two k-steps is one contiguous run.  Activations are split in
// registers (lane-local), weights go L2 -> LDS once per workgroup and stage (two k-steps x NBO blocks x 3 pieces, double
// buffered) and are read by the four waves with lane-linear ds_read_b128.  bias / ReLU / store / single-channel output are the
// fp32 chain's own helpers (same D layout).
// =====================================================================================================
template <int NBO> struct SStage { static constexpr int U4 = 2 * NBO * 3 * 64; static constexpr int PT = (U4 + 255) / 256; };

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // staging registers as a first-class vector (not HIP's uint4 struct)
template <int NBO>
__device__ __forceinline__ void sstage_load(const uint4* __restrict__ img, int st, int tid, u32x4 (&r)[SStage<NBO>::PT]) {
#pragma unroll
    for (int u = 0; u < SStage<NBO>::PT; u++) {
        const int e = min(tid + 256 * u, SStage<NBO>::U4 - 1);        // clamped, not guarded: straight-line loads (a ragged last
        r[u] = reinterpret_cast<const u32x4*>(img)[(long)st * SStage<NBO>::U4 + e];   // round re-reads / re-writes the stage's last element)
    }
}
template <int NBO>
__device__ __forceinline__ void sstage_store(uint4* ws, int tid, const u32x4 (&r)[SStage<NBO>::PT]) {
#pragma unroll
    for (int u = 0; u < SStage<NBO>::PT; u++) reinterpret_cast<u32x4*>(ws)[min(tid + 256 * u, SStage<NBO>::U4 - 1)] = r[u];
}
// eight fp32 values (one lane half's k-step) -> the three bf16 operand pieces
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&piece)[3]) {
    uint32_t b[3][8];
#pragma unroll
    for (int i = 0; i < 8; i++) split3(v[i], b[0][i], b[1][i], b[2][i]);
#pragma unroll
    for (int p = 0; p < 3; p++)
        piece[p] = __builtin_bit_cast(bf16x8, make_uint4(bf16_pair(b[p][0], b[p][1]), bf16_pair(b[p][2], b[p][3]),
                                                        bf16_pair(b[p][4], b[p][5]), bf16_pair(b[p][6], b[p][7])));
}
// one k-step of a chain layer: out[ob] += sum over the TERMS piece pairs of W[ob, step] (LDS tiles ksl * NBO + ob) x bp
template <int NBO, int TERMS>
__device__ __forceinline__ void schain_step(f32x16 (&out)[NBO], const uint4* ws, int ksl, int lane, const bf16x8 (&bp)[3]) {
    constexpr int NP = TERMS == 6 ? 3 : 2;
    bf16x8 w[NBO][NP];
#pragma unroll
    for (int ob = 0; ob < NBO; ob++)
#pragma unroll
        for (int p = 0; p < NP; p++) w[ob][p] = __builtin_bit_cast(bf16x8, ws[((ksl * NBO + ob) * 3 + p) * 64 + lane]);
#define SCH_TERM(PA, PB)                                                                                                          \
    _Pragma("unroll") for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ob][PA], bp[PB], out[ob], 0, 0, 0)
    if constexpr (TERMS == 6) { SCH_TERM(2, 0); SCH_TERM(0, 2); SCH_TERM(1, 1); }
    SCH_TERM(1, 0); SCH_TERM(0, 1); SCH_TERM(0, 0);
#undef SCH_TERM
}

// the same k-step two output blocks at a time: 24 operand registers alive instead of 48 (each accumulator sees its terms in the same
// order: same bits); for the persistent kernel's interpolation variant, which is at the register limit of two waves per SIMD
template <int NBO, int TERMS>
__device__ __forceinline__ void schain_step_pairs(f32x16 (&out)[NBO], const uint4* ws, int ksl, int lane, const bf16x8 (&bp)[3]) {
    static_assert(NBO % 2 == 0 && TERMS == 6, "pairs of output blocks, six terms");
#pragma unroll
    for (int o2 = 0; o2 < NBO; o2 += 2) {
        bf16x8 w[2][3];
#pragma unroll
        for (int ob = 0; ob < 2; ob++)
#pragma unroll
            for (int p = 0; p < 3; p++) w[ob][p] = __builtin_bit_cast(bf16x8, ws[((ksl * NBO + o2 + ob) * 3 + p) * 64 + lane]);
#define SCH_TERM2(PA, PB)                                                                                                         \
        _Pragma("unroll") for (int ob = 0; ob < 2; ob++) out[o2 + ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ob][PA], bp[PB], out[o2 + ob], 0, 0, 0)
        SCH_TERM2(2, 0); SCH_TERM2(0, 2); SCH_TERM2(1, 1); SCH_TERM2(1, 0); SCH_TERM2(0, 1); SCH_TERM2(0, 0);
#undef SCH_TERM2
    }
}

// The wave's 32 rows through the whole chain again on the fp32 pipe (rare path: a non-finite value among the rows' inputs, see
// mlp_layer_s_kernel's header).  Weights straight from the fp32 pack images in global memory (a.wpack, wpack1), the lane's own
// row as the B operand, layer 1 fed from layer 0's accumulators -- the fp32 chain kernels' products in their k order, no LDS
// staging, no barrier (the other waves of the workgroup are past theirs).  Stores go to the same addresses as the split path's,
// later in program order.
template <int MODE, int NB1>
__device__ __forceinline__ void schain_redo_f32(const ChainParams& C, long row, bool valid, int lane, int h, const float* s_b,
                                                const float (*s_bias)[128], bool out1) {
    constexpr int NB0 = 4, KB = 16;
    const MlpParams& P = C.a;
    RowMeta<MODE> meta;
    make_meta<MODE>(P, valid ? row : P.rows - 1, meta);
    f32x16 a0[NB0];
#pragma unroll
    for (int ob = 0; ob < NB0; ob++) a0[ob] = (f32x16){0};
    for (int kb = 0; kb < KB; kb++) {
        Raw<MODE> x;
        fast_fetch<MODE>(P, meta, 8 * kb + 4 * h, x);
        const float4 v = fast_finish<MODE>(meta, 8 * kb + 4 * h, x, nullptr, s_b);
#pragma unroll
        for (int ob = 0; ob < NB0; ob++) {
            const float4 w = ldw(P.wpack, KB, ob, kb, lane);
            a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, v.x, a0[ob], 0, 0, 0);
            a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, v.y, a0[ob], 0, 0, 0);
            a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, v.z, a0[ob], 0, 0, 0);
            a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, v.w, a0[ob], 0, 0, 0);
        }
    }
    bias_act<NB0>(a0, s_bias[0], P.relu, h);
    if (out1) { chain_out1<NB0>(C, a0, row, valid, lane, h); return; }
    if constexpr (NB1 == 0) chain_store<NB0>(C, a0, P.Nout, row, valid, lane, h);
    if constexpr (NB1 > 1) {
        f32x16 a1[NB1];
#pragma unroll
        for (int ob = 0; ob < NB1; ob++) a1[ob] = (f32x16){0};
#pragma unroll
        for (int kb = 0; kb < KB; kb++)                       // (register indices must be compile-time)
#pragma unroll
            for (int ob = 0; ob < NB1; ob++) {
                const float4 w = ldw(C.wpack1, KB, ob, kb, lane);
                a1[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, a0[kb >> 2][4 * (kb & 3) + 0], a1[ob], 0, 0, 0);
                a1[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, a0[kb >> 2][4 * (kb & 3) + 1], a1[ob], 0, 0, 0);
                a1[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, a0[kb >> 2][4 * (kb & 3) + 2], a1[ob], 0, 0, 0);
                a1[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, a0[kb >> 2][4 * (kb & 3) + 3], a1[ob], 0, 0, 0);
            }
        bias_act<NB1>(a1, s_bias[1], C.relu1, h);
        chain_store<NB1>(C, a1, C.N1, row, valid, lane, h);
    }
}

// MODE_PLAIN: the rows are read as they are; MODE_INTERP (hoisted FP0: act = 2, C1 = 0): row = relu(interp(Y) + act_bias), built in
// registers from the three neighbours' rows exactly as mlp_chain_fast_kernel builds it.  NB1 = 0: one layer.
template <int MODE, int NB1, int TERMS>
__global__ __launch_bounds__(256, 2) void mlp_chain_s_kernel(const ChainParams Cin) {
    constexpr int NB0 = 4, KS = 8;                           // K = 128 -> 128 (-> N1)
    constexpr int RD = MODE == MODE_PLAIN ? 4 : 2;           // k-steps of input requested ahead (plain rows stream from HBM)
    ChainParams C = Cin;
    C.a.rows = effective_rows(Cin.a);
    const MlpParams& P = C.a;
    const long tile_id = tile_of_block(P, blockIdx.x);
    if (tile_id * 128 >= P.rows) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const long row = (tile_id * 4 + wave) * 32 + j;
    const bool valid = row < P.rows;
    RowMeta<MODE> meta;
    make_meta<MODE>(P, valid ? row : P.rows - 1, meta);

    __shared__ __attribute__((aligned(16))) uint4 Ws[2][SStage<NB0>::U4];
    __shared__ __attribute__((aligned(16))) float s_bias[2][128];
    __shared__ __attribute__((aligned(16))) float s_b[128];
    if (tid < 128) {
        s_bias[0][tid] = P.bias ? P.bias[tid] : 0.f;
        s_bias[1][tid] = (NB1 > 0 && C.bias1 && tid < NB1 * 32) ? C.bias1[tid] : 0.f;
        s_b[tid] = MODE != MODE_PLAIN ? P.act_bias[tid] : 0.f;
    }
    const uint4* img0 = reinterpret_cast<const uint4*>(P.wsplit);
    const uint4* img1 = reinterpret_cast<const uint4*>(C.wsplit1);
    const bool out1 = chain_out1_applies<NB1, 0>(C);

    // the lane's row: k-step ks needs elements 16 ks + 4 h .. +4 and 16 ks + 8 + 4 h .. +4
    Raw<MODE> xa[KS], xb[KS];
#pragma unroll
    for (int ks = 0; ks < RD; ks++) { fast_fetch<MODE>(P, meta, 16 * ks + 4 * h, xa[ks]); fast_fetch<MODE>(P, meta, 16 * ks + 8 + 4 * h, xb[ks]); }
    u32x4 wr[SStage<NB0>::PT];
    sstage_load<NB0>(img0, 0, tid, wr);
    sstage_store<NB0>(Ws[0], tid, wr);
    __syncthreads();

    f32x16 a0[NB0];
#pragma unroll
    for (int ob = 0; ob < NB0; ob++) a0[ob] = (f32x16){0};
#pragma unroll
    for (int st = 0; st < KS / 2; st++) {
        if (st + 1 < KS / 2) sstage_load<NB0>(img0, st + 1, tid, wr);
#pragma unroll
        for (int ksl = 0; ksl < 2; ksl++) {
            const int ks = 2 * st + ksl;
            if (ks + RD < KS) {
                fast_fetch<MODE>(P, meta, 16 * (ks + RD) + 4 * h, xa[ks + RD]);
                fast_fetch<MODE>(P, meta, 16 * (ks + RD) + 8 + 4 * h, xb[ks + RD]);
            }
            const float4 va = fast_finish<MODE>(meta, 16 * ks + 4 * h, xa[ks], nullptr, s_b);
            const float4 vb = fast_finish<MODE>(meta, 16 * ks + 8 + 4 * h, xb[ks], nullptr, s_b);
            const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
            bf16x8 bp[3];
            split8(v, bp);
            schain_step<NB0, TERMS>(a0, Ws[st & 1], ksl, lane, bp);
        }
        if (st + 1 < KS / 2) sstage_store<NB0>(Ws[(st + 1) & 1], tid, wr);
        __syncthreads();
    }
    // a non-finite input value (its split holds NaN pieces, which reach every output of the row): the wave redoes its rows on the
    // fp32 pipe once the split path is through its barriers
    const bool bad = P.wpack != nullptr && wave_has_nonfinite<NB0>(a0);
    bias_act<NB0>(a0, s_bias[0], P.relu, h);
    if (out1) {
        chain_out1<NB0>(C, a0, row, valid, lane, h);
        if (bad) schain_redo_f32<MODE, NB1>(C, row, valid, lane, h, s_b, s_bias, true);
        return;
    }
    if constexpr (NB1 == 0) {
        if (!bad) chain_store<NB0>(C, a0, P.Nout, row, valid, lane, h);
        else schain_redo_f32<MODE, NB1>(C, row, valid, lane, h, s_b, s_bias, false);
    }
    if constexpr (NB1 > 1) {
        u32x4 w1[SStage<NB1>::PT];
        uint4* W1s = &Ws[0][0];                              // (a stage of NB1 <= 4 blocks fits a stage of four)
        sstage_load<NB1>(img1, 0, tid, w1);
        sstage_store<NB1>(W1s, tid, w1);
        __syncthreads();
        f32x16 a1[NB1];
#pragma unroll
        for (int ob = 0; ob < NB1; ob++) a1[ob] = (f32x16){0};
#pragma unroll
        for (int st = 0; st < NB0 * 2 / 2; st++) {           // NB0 * 2 k-steps, two per stage: stage st = input block st
            if (st + 1 < NB0) sstage_load<NB1>(img1, st + 1, tid, w1);
#pragma unroll
            for (int sh = 0; sh < 2; sh++) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = a0[st][8 * sh + e];
                bf16x8 bp[3];
                split8(v, bp);
                schain_step<NB1, TERMS>(a1, &Ws[st & 1][0], sh, lane, bp);
            }
            if (st + 1 < NB0) sstage_store<NB1>(&Ws[(st + 1) & 1][0], tid, w1);
            __syncthreads();
        }
        // ... or a finite layer-0 output whose layer-1 product overflows: the split of the overflowing sum holds NaN pieces where the fp32
        // chain returns +-inf (round-4 advisor finding: only the layer-0 accumulators were tested)
        const bool bad1 = bad || (P.wpack != nullptr && wave_has_nonfinite<NB1>(a1));
        bias_act<NB1>(a1, s_bias[1], C.relu1, h);
        if (!bad1) chain_store<NB1>(C, a1, C.N1, row, valid, lane, h);
        else schain_redo_f32<MODE, NB1>(C, row, valid, lane, h, s_b, s_bias, false);
    }
}

// =====================================================================================================
// mlp_chain_s_kernel with COOPERATIVE row access (round 6).  Same arithmetic, same products in the same order, same results bit for
// bit; what changes is who touches memory.  In the transposed formulation a lane IS a row: it used to fetch its own row 16 bytes
// at a time, so one wave-wide load touched 32-64 different cache lines for 32 bytes each, and the vector L1 processes a wave's load
// line by line -- the hoisted FP0 launch kept the L1 busy 76 % of its time (TCP_TOTAL_CACHE_ACCESSES / SQ_BUSY_CU_CYCLES, 74 M
// accesses for 1.2 GB of useful traffic) while the matrix pipe was busy 24 %; the heads the same way at 53 %.  Here eight lanes
// fetch 128 contiguous bytes of one row (one full line), a wave-wide load covers 8 rows x one line, 4 x fewer L1 accesses; the
// interpolation / bias / ReLU of the hoisted FP first layer happens in THAT layout (the three neighbours' values of a k-quad are in
// one lane), the finished 32-row x 32-k chunk goes through a 4.5 KB per-wave LDS tile and comes back in the lane-is-a-row layout the
// MFMA B operand wants (conflict-free both ways: row stride 36 dwords).  The output takes the same route backwards: D registers ->
// LDS tile -> 8 rows x 128 contiguous bytes per store.  A chunk is exactly one weight stage (two k-steps), so the stage loop and its
// barriers are unchanged.  PRCNN_CHAIN_COOP=0 selects the round-5 kernel (A/B switch).
// =====================================================================================================
#define CC_LD 36
// -DMLP_TIMING (tools/build_variant.py mlp.hip mlptiming -DMLP_TIMING; tools/mlp_timing.py): cycle counters of wave 0 of every workgroup of
// mlp_chain_c_kernel by phase, summed into a device array read back through prcnn_debug_mlp_timing; never defined in the product build
#ifdef MLP_TIMING
__device__ unsigned long long g_mlp_t[16];
#define MLP_T(k) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; }
#else
#define MLP_T(k)
#endif

template <int MODE> struct CoopRows;
template <> struct CoopRows<MODE_PLAIN> { const float* p[4]; };
template <> struct CoopRows<MODE_INTERP> { const float* p0[4]; const float* p1[4]; const float* p2[4]; float w0[4], w1[4], w2[4]; };
template <int MODE> struct CoopRaw;
template <> struct CoopRaw<MODE_PLAIN> { float4 a[4]; };
template <> struct CoopRaw<MODE_INTERP> { float4 a[4], b[4], c[4]; };

// rows rw0 + 8u + g (u = 0..3) of the wave's 32, k-quad q of every 32-float chunk: rows past the end are clamped (read, never stored)
__device__ __forceinline__ void coop_rows(const MlpParams& P, long rw0, int g, int q, CoopRows<MODE_PLAIN>& cr) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long r = min(rw0 + 8 * u + g, P.rows - 1);
        cr.p[u] = P.in + r * (long)P.ld_in + 4 * q;
    }
}
__device__ __forceinline__ void coop_rows(const MlpParams& P, long rw0, int g, int q, CoopRows<MODE_INTERP>& cr) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long r = min(rw0 + 8 * u + g, P.rows - 1);
        const long base = (r / P.n) * (long)P.m;
        const int32_t* id = P.idx3 + r * 3;
        const float* w = P.w3 + r * 3;
        cr.p0[u] = P.known + (base + id[0]) * (long)P.ld_known + 4 * q;
        cr.p1[u] = P.known + (base + id[1]) * (long)P.ld_known + 4 * q;
        cr.p2[u] = P.known + (base + id[2]) * (long)P.ld_known + 4 * q;
        cr.w0[u] = w[0]; cr.w1[u] = w[1]; cr.w2[u] = w[2];
    }
}
__device__ __forceinline__ void coop_fetch(const CoopRows<MODE_PLAIN>& cr, int c, CoopRaw<MODE_PLAIN>& v) {
#pragma unroll
    for (int u = 0; u < 4; u++) v.a[u] = ld4(cr.p[u] + 32 * c);
}
__device__ __forceinline__ void coop_fetch(const CoopRows<MODE_INTERP>& cr, int c, CoopRaw<MODE_INTERP>& v) {
#pragma unroll
    for (int u = 0; u < 4; u++) { v.a[u] = ld4(cr.p0[u] + 32 * c); v.b[u] = ld4(cr.p1[u] + 32 * c); v.c[u] = ld4(cr.p2[u] + 32 * c); }
}
// relu(interp1(w, a, b, c) + bias) on a quad, two components per instruction (v_pk_mul_f32 / v_pk_add_f32: the only fp32 VALU forms that
// issue at full rate on gfx950; every component goes through the same individually rounded operations as interp1 -- no FMA: the
// library is built with -ffp-contract=off -- so the values are those of fast_finish bit for bit)
typedef float mlp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 interp_bias_relu4(float w0, float w1, float w2, float4 a, float4 b, float4 c, float4 bias) {
    const mlp_f32x2 W0 = {w0, w0}, W1 = {w1, w1}, W2 = {w2, w2};
    const mlp_f32x2 lo = ((mlp_f32x2){a.x, a.y} * W0 + (mlp_f32x2){b.x, b.y} * W1) + (mlp_f32x2){c.x, c.y} * W2 + (mlp_f32x2){bias.x, bias.y};
    const mlp_f32x2 hi = ((mlp_f32x2){a.z, a.w} * W0 + (mlp_f32x2){b.z, b.w} * W1) + (mlp_f32x2){c.z, c.w} * W2 + (mlp_f32x2){bias.z, bias.w};
    return make_float4(fmaxf(lo.x, 0.f), fmaxf(lo.y, 0.f), fmaxf(hi.x, 0.f), fmaxf(hi.y, 0.f));
}
// the finished A values of row 8u + g, k = 32 c + 4 q .. + 4 (the expressions of fast_finish, element for element)
__device__ __forceinline__ float4 coop_finish(const CoopRows<MODE_PLAIN>&, const CoopRaw<MODE_PLAIN>& v, int u, float4) { return v.a[u]; }
__device__ __forceinline__ float4 coop_finish(const CoopRows<MODE_INTERP>& cr, const CoopRaw<MODE_INTERP>& v, int u, float4 b) {
    return interp_bias_relu4(cr.w0[u], cr.w1[u], cr.w2[u], v.a[u], v.b[u], v.c[u], b);
}

// D registers -> the wave's LDS tile -> 8 rows x 128 contiguous bytes per store instruction (32 channels at a time)
template <int NB>
__device__ __forceinline__ void coop_store(const ChainParams& C, f32x16 (&acc)[NB], int Nlast, long rw0, int lane, float* Tw) {
    const MlpParams& P = C.a;
    const int h = lane >> 5, j = lane & 31, g = lane >> 3, q = lane & 7;
#pragma unroll
    for (int ob = 0; ob < NB; ob++) {
        if (ob * 32 >= Nlast) continue;                  // (no break: the loop must unroll, acc[ob] is a register index)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int qq = 0; qq < 4; qq++)
            *reinterpret_cast<float4*>(Tw + j * CC_LD + 8 * qq + 4 * h) =
                make_float4(acc[ob][4 * qq], acc[ob][4 * qq + 1], acc[ob][4 * qq + 2], acc[ob][4 * qq + 3]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float4 v = *reinterpret_cast<const float4*>(Tw + (8 * u + g) * CC_LD + 4 * q);
            const long r = rw0 + 8 * u + g;
            const int c = ob * 32 + 4 * q;
            if (r < P.rows && c < Nlast) *reinterpret_cast<float4*>(P.out + r * (long)P.ld_out + P.col_off + c) = v;
        }
    }
}

template <int MODE, int NB1, int TERMS>
__global__ __launch_bounds__(256, 2) void mlp_chain_c_kernel(const ChainParams Cin) {
    constexpr int NB0 = 4, NST = 4;                          // K = 128 -> 128 (-> N1); four stages of two k-steps = four 32-float chunks
    ChainParams C = Cin;
    C.a.rows = effective_rows(Cin.a);
    const MlpParams& P = C.a;
    const long tile_id = tile_of_block(P, blockIdx.x);
    if (tile_id * 128 >= P.rows) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31, g = lane >> 3, q = lane & 7;
    const long rw0 = (tile_id * 4 + wave) * 32;              // the wave's first row
    const long row = rw0 + j;
    const bool valid = row < P.rows;

    __shared__ __attribute__((aligned(16))) uint4 Ws[2][SStage<NB0>::U4];
    __shared__ __attribute__((aligned(16))) float Tbuf[4][32 * CC_LD];
    __shared__ __attribute__((aligned(16))) float s_bias[2][128];
    __shared__ __attribute__((aligned(16))) float s_b[128];
    float* Tw = Tbuf[wave];
    if (tid < 128) {
        s_bias[0][tid] = P.bias ? P.bias[tid] : 0.f;
        s_bias[1][tid] = (NB1 > 0 && C.bias1 && tid < NB1 * 32) ? C.bias1[tid] : 0.f;
        s_b[tid] = MODE != MODE_PLAIN ? P.act_bias[tid] : 0.f;
    }
    const uint4* img0 = reinterpret_cast<const uint4*>(P.wsplit);
    const uint4* img1 = reinterpret_cast<const uint4*>(C.wsplit1);
    const bool out1 = chain_out1_applies<NB1, 0>(C);

#ifdef MLP_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    CoopRows<MODE> cr;
    coop_rows(P, rw0, g, q, cr);
    CoopRaw<MODE> x[NST];
    constexpr int RD = MODE == MODE_PLAIN ? 2 : 1;           // chunks requested ahead (a chunk of interpolation sources is 48 registers)
#pragma unroll
    for (int c = 0; c < RD; c++) coop_fetch(cr, c, x[c]);
    u32x4 wr[SStage<NB0>::PT];
    sstage_load<NB0>(img0, 0, tid, wr);
    sstage_store<NB0>(Ws[0], tid, wr);
    __syncthreads();
    MLP_T(0)

    f32x16 a0[NB0];
#pragma unroll
    for (int ob = 0; ob < NB0; ob++) a0[ob] = (f32x16){0};
#pragma unroll
    for (int st = 0; st < NST; st++) {
        if (st + 1 < NST) sstage_load<NB0>(img0, st + 1, tid, wr);
        if (st + RD < NST) coop_fetch(cr, st + RD, x[st + RD]);
        const float4 b4 = MODE != MODE_PLAIN ? ld4(s_b + 32 * st + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; u++) *reinterpret_cast<float4*>(Tw + (8 * u + g) * CC_LD + 4 * q) = coop_finish(cr, x[st], u, b4);
        MLP_T(1)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ksl = 0; ksl < 2; ksl++) {
            const float4 va = *reinterpret_cast<const float4*>(Tw + j * CC_LD + 16 * ksl + 4 * h);
            const float4 vb = *reinterpret_cast<const float4*>(Tw + j * CC_LD + 16 * ksl + 8 + 4 * h);
            const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
            bf16x8 bp[3];
            split8(v, bp);
            schain_step<NB0, TERMS>(a0, Ws[st & 1], ksl, lane, bp);
        }
        MLP_T(2)
        __builtin_amdgcn_wave_barrier();
        if (st + 1 < NST) sstage_store<NB0>(Ws[(st + 1) & 1], tid, wr);
        MLP_T(3)
        __syncthreads();
        MLP_T(4)
    }
    const bool coop_out = P.pool_ns == 0 && ((P.ld_out | P.col_off) % 4 == 0) && aligned16(P.out);
    const bool bad = P.wpack != nullptr && wave_has_nonfinite<NB0>(a0);
    bias_act<NB0>(a0, s_bias[0], P.relu, h);
    if (out1) {
        chain_out1<NB0>(C, a0, row, valid, lane, h);
        if (bad) schain_redo_f32<MODE, NB1>(C, row, valid, lane, h, s_b, s_bias, true);
        return;
    }
    if constexpr (NB1 == 0) {
        if (bad) schain_redo_f32<MODE, NB1>(C, row, valid, lane, h, s_b, s_bias, false);
        else if (coop_out && P.Nout % 4 == 0) coop_store<NB0>(C, a0, P.Nout, rw0, lane, Tw);
        else chain_store<NB0>(C, a0, P.Nout, row, valid, lane, h);
    }
    if constexpr (NB1 > 1) {
        u32x4 w1[SStage<NB1>::PT];
        uint4* W1s = &Ws[0][0];                              // (a stage of NB1 <= 4 blocks fits a stage of four)
        sstage_load<NB1>(img1, 0, tid, w1);
        sstage_store<NB1>(W1s, tid, w1);
        __syncthreads();
        MLP_T(5)
        f32x16 a1[NB1];
#pragma unroll
        for (int ob = 0; ob < NB1; ob++) a1[ob] = (f32x16){0};
#pragma unroll
        for (int st = 0; st < NB0 * 2 / 2; st++) {
            if (st + 1 < NB0) sstage_load<NB1>(img1, st + 1, tid, w1);
#pragma unroll
            for (int sh = 0; sh < 2; sh++) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = a0[st][8 * sh + e];
                bf16x8 bp[3];
                split8(v, bp);
                schain_step<NB1, TERMS>(a1, &Ws[st & 1][0], sh, lane, bp);
            }
            MLP_T(6)
            if (st + 1 < NB0) sstage_store<NB1>(&Ws[(st + 1) & 1][0], tid, w1);
            __syncthreads();
            MLP_T(4)
        }
        const bool bad1 = bad || (P.wpack != nullptr && wave_has_nonfinite<NB1>(a1));
        bias_act<NB1>(a1, s_bias[1], C.relu1, h);
        if (bad1) schain_redo_f32<MODE, NB1>(C, row, valid, lane, h, s_b, s_bias, false);
        else if (coop_out && C.N1 % 4 == 0) coop_store<NB1>(C, a1, C.N1, rw0, lane, Tw);
        else chain_store<NB1>(C, a1, C.N1, row, valid, lane, h);
    }
#ifdef MLP_TIMING
    MLP_T(7)
    if (tid == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_mlp_t[k], tacc[k]);
    if (tid == 0) atomicAdd(&g_mlp_t[8], 1ULL);
#endif
}
#ifdef MLP_TIMING
PRCNN_API int prcnn_debug_mlp_timing(unsigned long long* out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_mlp_t), sizeof(unsigned long long) * 16) != hipSuccess) return PRCNN_EHIP;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_t), z, sizeof(z)) != hipSuccess) return PRCNN_EHIP; }
    return PRCNN_OK;
}
#endif
// =====================================================================================================
// PERSISTENT form of the cooperative-row split chain (round 6): ONE layer 128 -> 128 (+ the single-channel dot product of the
// classification head) with the WHOLE split weight image resident in LDS (96 KB) for the lifetime of an 8-wave workgroup, one per CU.
// Cycle counters of mlp_chain_c_kernel (tools/mlp_timing.py, FP0): of 55.8 k cycles a workgroup spends on its 128 rows, the matrix pipe
// is busy 6.1 k per wave; 13 k go to the prologue -- neighbour indices from HBM, then the rows they point to, then the first weight
// stage, three dependent round trips before the first product -- and every stage ends in a barrier.  Here each WAVE walks its own
// sequence of 32-row tiles with no barrier at all (weights never move), and the next tile's neighbour indices, addresses and first
// chunk are requested while the current tile multiplies, so the round trips of tile t+1 run under the products of tile t.
// Same chunks, same transposition tile, same products in the same order as mlp_chain_c_kernel / mlp_chain_s_kernel: same bits.
// Tile order: with xcd_tpf > 0 (gather mode, frames a multiple of 8) XCD x takes frames x, x+8, ... one after the other and spreads
// each frame's tiles over its workgroups' waves, so a frame's gather source stays in that XCD's L2 (as tile_of_block does for the
// per-tile kernels).  Row offsets are 32-bit (host-checked: the source array is smaller than 4 GB).
// =====================================================================================================
#define CP_WAVES 8
template <int MODE> struct CoopRows32;                   // per-lane: rows 8u + g of the wave's tile, as float offsets from the source base
template <> struct CoopRows32<MODE_PLAIN> { unsigned o[4]; };
template <> struct CoopRows32<MODE_INTERP> { unsigned o0[4], o1[4], o2[4]; float w0[4], w1[4], w2[4]; };
template <int MODE> struct CoopMeta;                     // the raw words a tile's addresses are computed from (requested a tile ahead)
template <> struct CoopMeta<MODE_PLAIN> { int unused; };
template <> struct CoopMeta<MODE_INTERP> { int i0[4], i1[4], i2[4]; float w0[4], w1[4], w2[4]; };

__device__ __forceinline__ void coop_meta_fetch(const MlpParams&, long, int, CoopMeta<MODE_PLAIN>&) {}
__device__ __forceinline__ void coop_meta_fetch(const MlpParams& P, long rw0, int g, CoopMeta<MODE_INTERP>& mt) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long r = min(rw0 + 8 * u + g, P.rows - 1);
        const int32_t* id = P.idx3 + r * 3;
        const float* w = P.w3 + r * 3;
        mt.i0[u] = id[0]; mt.i1[u] = id[1]; mt.i2[u] = id[2];
        mt.w0[u] = w[0]; mt.w1[u] = w[1]; mt.w2[u] = w[2];
    }
}
__device__ __forceinline__ void coop_rows32(const MlpParams& P, long rw0, int g, int q, const CoopMeta<MODE_PLAIN>&, CoopRows32<MODE_PLAIN>& cr) {
#pragma unroll
    for (int u = 0; u < 4; u++) cr.o[u] = (unsigned)(min(rw0 + 8 * u + g, P.rows - 1) * (long)P.ld_in) + 4u * q;
}
__device__ __forceinline__ void coop_rows32(const MlpParams& P, long rw0, int g, int q, const CoopMeta<MODE_INTERP>& mt, CoopRows32<MODE_INTERP>& cr) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long r = min(rw0 + 8 * u + g, P.rows - 1);
        const long base = (r / P.n) * (long)P.m;
        cr.o0[u] = (unsigned)((base + mt.i0[u]) * (long)P.ld_known) + 4u * q;
        cr.o1[u] = (unsigned)((base + mt.i1[u]) * (long)P.ld_known) + 4u * q;
        cr.o2[u] = (unsigned)((base + mt.i2[u]) * (long)P.ld_known) + 4u * q;
        cr.w0[u] = mt.w0[u]; cr.w1[u] = mt.w1[u]; cr.w2[u] = mt.w2[u];
    }
}
__device__ __forceinline__ void coop_fetch32(const MlpParams& P, const CoopRows32<MODE_PLAIN>& cr, int c, CoopRaw<MODE_PLAIN>& v) {
#pragma unroll
    for (int u = 0; u < 4; u++) v.a[u] = ld4(P.in + cr.o[u] + 32 * c);
}
__device__ __forceinline__ void coop_fetch32(const MlpParams& P, const CoopRows32<MODE_INTERP>& cr, int c, CoopRaw<MODE_INTERP>& v) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        v.a[u] = ld4(P.known + cr.o0[u] + 32 * c); v.b[u] = ld4(P.known + cr.o1[u] + 32 * c); v.c[u] = ld4(P.known + cr.o2[u] + 32 * c);
    }
}
__device__ __forceinline__ float4 coop_finish32(const CoopRows32<MODE_PLAIN>&, const CoopRaw<MODE_PLAIN>& v, int u, float4) { return v.a[u]; }
__device__ __forceinline__ float4 coop_finish32(const CoopRows32<MODE_INTERP>& cr, const CoopRaw<MODE_INTERP>& v, int u, float4 b) {
    return interp_bias_relu4(cr.w0[u], cr.w1[u], cr.w2[u], v.a[u], v.b[u], v.c[u], b);
}

// the wave's tile sequence (32-row tiles)
struct CoopTiles {
    long t;                 // current tile, -1: none left
    long step, end;         // plain order: t += step while t < end
    long f, i, tpf, wx, WX, nframes;   // XCD order: frame f (step 8), tile i of the frame (step WX)
    bool xcd;
};
__device__ __forceinline__ void coop_tiles_init(CoopTiles& it, const MlpParams& P, int wave) {
    const long ntiles = (P.rows + 31) / 32;
    it.xcd = P.xcd_tpf > 0 && (gridDim.x & 7) == 0;
    if (it.xcd) {
        it.tpf = 4L * P.xcd_tpf; it.nframes = ntiles / it.tpf;
        it.WX = (long)(gridDim.x >> 3) * CP_WAVES; it.wx = (long)(blockIdx.x >> 3) * CP_WAVES + wave;
        it.f = blockIdx.x & 7; it.i = it.wx;
        it.t = (it.i < it.tpf && it.f < it.nframes) ? it.f * it.tpf + it.i : -1;
    } else {
        it.step = (long)gridDim.x * CP_WAVES; it.end = ntiles;
        it.t = (long)blockIdx.x * CP_WAVES + wave;
        if (it.t >= it.end) it.t = -1;
    }
}
__device__ __forceinline__ long coop_tiles_next(CoopTiles& it) {        // -> the tile after the current one (-1: none); advances
    if (it.t < 0) return -1;
    if (it.xcd) {
        it.i += it.WX;
        if (it.i >= it.tpf) { it.i = it.wx; it.f += 8; }
        it.t = (it.f < it.nframes) ? it.f * it.tpf + it.i : -1;
    } else {
        it.t += it.step;
        if (it.t >= it.end) it.t = -1;
    }
    return it.t;
}

template <int MODE, int NB1, int TERMS>
__global__ __launch_bounds__(CP_WAVES * 64, 1) void mlp_chain_p_kernel(const ChainParams Cin) {
    static_assert(NB1 == 0 || NB1 == 1, "one layer, or one layer + the single-channel output");
    constexpr int NB0 = 4, NST = 4;
    constexpr int RD = MODE == MODE_PLAIN ? 2 : 1;           // chunks requested ahead
    ChainParams C = Cin;
    C.a.rows = effective_rows(Cin.a);
    const MlpParams& P = C.a;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31, g = lane >> 3, q = lane & 7;

    extern __shared__ __attribute__((aligned(16))) unsigned char cp_lds[];
    uint4* Wl = reinterpret_cast<uint4*>(cp_lds);                                            // [NST][U4]: the whole layer-0 image
    float* Tw = reinterpret_cast<float*>(cp_lds + (size_t)NST * SStage<NB0>::U4 * 16) + wave * 32 * CC_LD;
    float* s_bias = reinterpret_cast<float*>(cp_lds + (size_t)NST * SStage<NB0>::U4 * 16) + CP_WAVES * 32 * CC_LD;   // [128]
    float* s_b = s_bias + 128;                                                               // [128]
    {
        const u32x4* img = reinterpret_cast<const u32x4*>(P.wsplit);
        for (int e = tid; e < NST * SStage<NB0>::U4; e += CP_WAVES * 64) reinterpret_cast<u32x4*>(Wl)[e] = img[e];
        if (tid < 128) { s_bias[tid] = P.bias ? P.bias[tid] : 0.f; s_b[tid] = MODE != MODE_PLAIN ? P.act_bias[tid] : 0.f; }
    }
    __syncthreads();                                         // the only barrier of the kernel
    const bool out1 = NB1 == 1;
    const bool coop_out = P.pool_ns == 0 && ((P.ld_out | P.col_off) % 4 == 0) && aligned16(P.out) && P.Nout % 4 == 0;

#ifdef MLP_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    CoopTiles it;
    coop_tiles_init(it, P, wave);
    if (it.t < 0) return;
    long cur = it.t;
    CoopMeta<MODE> mt;
    CoopRows32<MODE> cr;
    CoopRaw<MODE> x[NST];
    coop_meta_fetch(P, cur * 32, g, mt);
    coop_rows32(P, cur * 32, g, q, mt, cr);
#pragma unroll
    for (int c = 0; c < RD; c++) coop_fetch32(P, cr, c, x[c]);
    MLP_T(0)

    while (cur >= 0) {
        const long rw0 = cur * 32;
        const long nxt = coop_tiles_next(it);
        const long nrw0 = (nxt >= 0 ? nxt : cur) * 32;       // (no next tile: the requests repeat this tile's, results unused)
        CoopRows32<MODE> crn;
        f32x16 a0[NB0];
#pragma unroll
        for (int ob = 0; ob < NB0; ob++) a0[ob] = (f32x16){0};
#pragma unroll
        for (int st = 0; st < NST; st++) {
            if (st == 0) coop_meta_fetch(P, nrw0, g, mt);                      // next tile's neighbour indices / weights: in flight for >= 2 chunks
            // (the offset goes through an opaque asm so that the four bias quads are re-read from LDS every tile: hoisted out of the
            //  tile loop they are 16 more live registers in a kernel at the limit of two waves per SIMD)
            int qo = 4 * q;
            asm volatile("" : "+v"(qo));
            const float4 b4 = MODE != MODE_PLAIN ? ld4(s_b + 32 * st + qo) : make_float4(0.f, 0.f, 0.f, 0.f);
            // PLAIN (16 registers a chunk): the request runs two chunks ahead, issued before this chunk is consumed; INTERP (48 a chunk):
            // one chunk ahead and issued AFTER this chunk's values are finished into the LDS tile, so that only one chunk of raw
            // neighbour rows is alive at a time (with two the kernel spilled 46 registers at two waves per SIMD)
            if (MODE == MODE_PLAIN) {
                if (st + RD == NST) coop_rows32(P, nrw0, g, q, mt, crn);
                if (st + RD < NST) coop_fetch32(P, cr, st + RD, x[st + RD]);
                else coop_fetch32(P, crn, st + RD - NST, x[st + RD - NST]);    // the next tile's first chunks, under this tile's last products
            }
#pragma unroll
            for (int u = 0; u < 4; u++) *reinterpret_cast<float4*>(Tw + (8 * u + g) * CC_LD + 4 * q) = coop_finish32(cr, x[st], u, b4);
            if (MODE != MODE_PLAIN) {
                if (st + RD == NST) coop_rows32(P, nrw0, g, q, mt, crn);       // the next tile's addresses, just before its first chunk is requested
                if (st + RD < NST) coop_fetch32(P, cr, st + RD, x[st + RD]);
                else coop_fetch32(P, crn, st + RD - NST, x[st + RD - NST]);
            }
            MLP_T(1)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ksl = 0; ksl < 2; ksl++) {
                const float4 va = *reinterpret_cast<const float4*>(Tw + j * CC_LD + 16 * ksl + 4 * h);
                const float4 vb = *reinterpret_cast<const float4*>(Tw + j * CC_LD + 16 * ksl + 8 + 4 * h);
                const float v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
                bf16x8 bp[3];
                split8(v, bp);
                // the stage's lane-adjusted base goes through an opaque asm: one address register + immediate offsets per stage; left to
                // itself the compiler hoists one address register per weight tile beyond the 64 KB immediate range out of the tile loop
                // (~20 registers: spills at two waves per SIMD)
                unsigned wo = (unsigned)(st * SStage<NB0>::U4 + lane) * 16u;
                asm volatile("" : "+v"(wo));
                const uint4* wst = reinterpret_cast<const uint4*>(cp_lds + wo);
                if (MODE == MODE_PLAIN) schain_step<NB0, TERMS>(a0, wst, ksl, 0, bp);
                else schain_step_pairs<NB0, TERMS>(a0, wst, ksl, 0, bp);
            }
            MLP_T(2)
            __builtin_amdgcn_wave_barrier();
        }
        const long row = rw0 + j;
        const bool valid = row < P.rows;
        const bool bad = P.wpack != nullptr && wave_has_nonfinite<NB0>(a0);
        bias_act<NB0>(a0, s_bias, P.relu, h);
        if (out1) {
            chain_out1<NB0>(C, a0, row, valid, lane, h);
            if (bad) schain_redo_f32<MODE, NB1>(C, row, valid, lane, h, s_b, reinterpret_cast<const float (*)[128]>(s_bias), true);
        } else {
            if (bad) schain_redo_f32<MODE, NB1>(C, row, valid, lane, h, s_b, reinterpret_cast<const float (*)[128]>(s_bias), false);
            else if (coop_out) coop_store<NB0>(C, a0, P.Nout, rw0, lane, Tw);
            else chain_store<NB0>(C, a0, P.Nout, row, valid, lane, h);
        }
        cr = crn;
        cur = nxt;
        MLP_T(7)
#ifdef MLP_TIMING
        if (lane == 0 && wave == 0) atomicAdd(&g_mlp_t[8], 1ULL);
#endif
    }
#ifdef MLP_TIMING
    if (lane == 0 && wave == 0) for (int k = 0; k < 8; k++) atomicAdd(&g_mlp_t[k], tacc[k]);
#endif
}
template <int MODE, int NB1> static constexpr size_t chain_p_lds_bytes() {
    return (size_t)4 * SStage<4>::U4 * 16 + (size_t)CP_WAVES * 32 * CC_LD * 4 + 2 * 128 * 4;
}
static int chain_p_grid() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n & ~7;                                        // a multiple of 8: the XCD-aware tile order counts on it
    }();
    return cus > 0 ? cus : 8;
}
// PRCNN_CHAIN_PERSIST=0: the per-tile kernels (A/B switch, same bits)
static bool chain_persist_on() {                           // (read per launch: the tests flip it inside one process)
    const char* e = getenv("PRCNN_CHAIN_PERSIST");
    return !(e && atoi(e) == 0);
}
template <int MODE, int NB1>
static int launch_chain_p(const ChainParams& C, hipStream_t s) {
    constexpr size_t lds = chain_p_lds_bytes<MODE, NB1>();
    static PrcnnLdsLimit attr;
    if (!attr.raise((const void*)mlp_chain_p_kernel<MODE, NB1, 6>, (int)lds))
        return prcnn_fail(PRCNN_EHIP, "prcnn_mlp_chain(persistent split): cannot raise the dynamic LDS limit");
    const long tiles = (C.a.rows + 31) / 32;
    const int grid = (int)min((long)chain_p_grid(), (tiles + CP_WAVES - 1) / CP_WAVES);
    hipLaunchKernelGGL((mlp_chain_p_kernel<MODE, NB1, 6>), dim3(grid), dim3(CP_WAVES * 64), lds, s, C);
    return PRCNN_OK;
}
static bool chain_coop_on() {                              // (read per launch: the tests flip it inside one process)
    const char* e = getenv("PRCNN_CHAIN_COOP");
    return !(e && atoi(e) == 0);
}
static bool chain_coop_forced() {                          // 2: the cooperative form also where the lane-is-a-row kernel is the default
    const char* e = getenv("PRCNN_CHAIN_COOP");
    return e && atoi(e) == 2;
}

// =====================================================================================================
// PERSISTENT chain: the fast chain's arithmetic with ALL layers' packed weights resident in LDS for the lifetime of the
// workgroup (48-128 KB of the CU's 160 KB; one 8-wave workgroup per CU = 2 waves per SIMD sharing one copy).  Each
// wave then loops over 32-row tiles on its own: no weight staging, no barriers, no workgroup launch per 128 rows, and
// the next tile's neighbour metadata and first input pieces are requested while the current tile multiplies.
// Same MFMA order and activation arithmetic as mlp_chain_fast_kernel / mlp_chain_kernel: bit-identical results.
// =====================================================================================================
#define PERS_WAVES 8
template <int KB, int NB>
__device__ __forceinline__ void pers_load_layer(float* dst, const float* __restrict__ wpack, int wave, int lane) {
    for (int t = wave; t < KB * NB; t += PERS_WAVES) {          // LDS tile index = kb * NB + ob (what lds_w expects)
        const int kb = t / NB, ob = t - kb * NB;
        *reinterpret_cast<float4*>(dst + (t * 64 + lane) * 4) = ldw(wpack, KB, ob, kb, lane);
    }
}
template <int NBI, int NBO>
__device__ __forceinline__ void pers_layer(const f32x16 (&in)[NBI], f32x16 (&out)[NBO], const float* wl, int lane) {
#pragma unroll
    for (int ob = 0; ob < NBO; ob++) out[ob] = (f32x16){0};
#pragma unroll
    for (int kb = 0; kb < NBI * 4; kb++) {
        const int pb = kb / 4, q = kb % 4;
        float4 w[NBO];
#pragma unroll
        for (int ob = 0; ob < NBO; ob++) w[ob] = lds_w(wl, kb * NBO + ob, lane);
#pragma unroll
        for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].x, in[pb][4 * q + 0], out[ob], 0, 0, 0);
#pragma unroll
        for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].y, in[pb][4 * q + 1], out[ob], 0, 0, 0);
#pragma unroll
        for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].z, in[pb][4 * q + 2], out[ob], 0, 0, 0);
#pragma unroll
        for (int ob = 0; ob < NBO; ob++) out[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].w, in[pb][4 * q + 3], out[ob], 0, 0, 0);
    }
}

template <int MODE, int KB0, int NB0, int NB1, int NB2>
static constexpr size_t pers_lds_bytes() {
    return ((size_t)(KB0 * NB0 + 4 * NB0 * NB1 + 4 * NB1 * NB2) * 256 + (MODE == MODE_PLAIN ? 0 : (FAST_MAX_K + 16) * 4)) * sizeof(float);
}

template <int MODE, int KB0, int NB0, int NB1, int NB2>
__global__ __launch_bounds__(PERS_WAVES * 64) void mlp_chain_pers_kernel(const ChainParams Cin) {
    ChainParams C = Cin;
    C.a.rows = effective_rows(Cin.a);
    const MlpParams& P = C.a;
    extern __shared__ __attribute__((aligned(16))) float Wl[];
    constexpr int T0 = KB0 * NB0, T1 = 4 * NB0 * NB1, T2 = 4 * NB1 * NB2;
    float* W0 = Wl;
    float* W1 = W0 + T0 * 256;
    float* W2 = W1 + T1 * 256;
    float* s_wx = W2 + T2 * 256;                        // (K,3) act_wx, then act_bias (MODE_GROUP / MODE_INTERP only)
    float* s_b = s_wx + (FAST_MAX_K + 16) * 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    pers_load_layer<KB0, NB0>(W0, P.wpack, wave, lane);
    if (NB1 > 0) pers_load_layer<NB0 * 4, (NB1 ? NB1 : 1)>(W1, C.wpack1, wave, lane);
    if (NB2 > 0) pers_load_layer<(NB1 ? NB1 : 1) * 4, (NB2 ? NB2 : 1)>(W2, C.wpack2, wave, lane);
    if (MODE == MODE_GROUP)
        for (int t = threadIdx.x; t < P.K * 3; t += PERS_WAVES * 64) s_wx[t] = P.act_wx[t];
    if (MODE != MODE_PLAIN)
        for (int t = threadIdx.x; t < P.K; t += PERS_WAVES * 64) s_b[t] = P.act_bias[t];
    __syncthreads();

    const long ntiles = (P.rows + 31) >> 5;
    const long stride = (long)gridDim.x * PERS_WAVES;
    long tile = (long)blockIdx.x * PERS_WAVES + wave;
    if (tile >= ntiles) return;
    auto row_of = [&](long t) { long r = (t < ntiles ? t : ntiles - 1) * 32 + j; return r < P.rows ? r : P.rows - 1; };
    constexpr int K0 = KB0 * 8;
    const int klast = K0 - 8 + 4 * h;
    RowMeta<MODE> meta, meta_n;
    Raw<MODE> r0, r1;
    make_meta<MODE>(P, row_of(tile), meta);
    fast_fetch<MODE>(P, meta, 4 * h, r0);
    fast_fetch<MODE>(P, meta, 8 + 4 * h, r1);
    for (; tile < ntiles; tile += stride) {
        const long row = tile * 32 + j;
        const bool valid = row < P.rows;
        make_meta<MODE>(P, row_of(tile + stride), meta_n);           // next tile: index -> coordinates, in flight during layer 0
        f32x16 a0[NB0];
#pragma unroll
        for (int ob = 0; ob < NB0; ob++) a0[ob] = (f32x16){0};
        float4 bcur = fast_finish<MODE>(meta, 4 * h, r0, s_wx, s_b);
#pragma unroll
        for (int kb = 0; kb < KB0; kb++) {
            Raw<MODE> r2;
            fast_fetch<MODE>(P, meta, min(8 * (kb + 2) + 4 * h, klast), r2);
            float4 w[NB0];
#pragma unroll
            for (int ob = 0; ob < NB0; ob++) w[ob] = lds_w(W0, kb * NB0 + ob, lane);
#pragma unroll
            for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].x, bcur.x, a0[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].y, bcur.y, a0[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].z, bcur.z, a0[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NB0; ob++) a0[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ob].w, bcur.w, a0[ob], 0, 0, 0);
            bcur = fast_finish<MODE>(meta, min(8 * (kb + 1) + 4 * h, klast), r1, s_wx, s_b);
            r1 = r2;
        }
        // first two input pieces of the next tile: requested now, consumed at the top of the next iteration
        fast_fetch<MODE>(P, meta_n, 4 * h, r0);
        fast_fetch<MODE>(P, meta_n, 8 + 4 * h, r1);
        bias_act<NB0>(a0, P.bias, P.relu, h);
        if constexpr (NB1 == 0) {
            chain_store<NB0>(C, a0, P.Nout, row, valid, lane, h);
        } else if (chain_out1_applies<NB1, NB2>(C)) {
            chain_out1<NB0>(C, a0, row, valid, lane, h);
        } else {
            f32x16 a1[NB1];
            pers_layer<NB0, NB1>(a0, a1, W1, lane);
            bias_act<NB1>(a1, C.bias1, C.relu1, h);
            if constexpr (NB2 == 0) {
                chain_store<NB1>(C, a1, C.N1, row, valid, lane, h);
            } else {
                f32x16 a2[NB2];
                pers_layer<NB1, NB2>(a1, a2, W2, lane);
                bias_act<NB2>(a2, C.bias2, C.relu2, h);
                chain_store<NB2>(C, a2, C.N2, row, valid, lane, h);
            }
        }
        meta = meta_n;
    }
}

// =====================================================================================================
// SA level 0 (no input features: the grouped row is just the centred xyz, K = 3): widths 16/16/32 and 32/32/64.
// The generic chain kernel spends ~4 700 instructions around 52 MFMAs per 32-row tile here (bounds-checked fetches, a
// weight stage + barrier per layer, one workgroup launch per 128 rows).  All weights and biases of such a stack fit in
// registers (<= 52 + 64 VGPRs), so this variant keeps them there, runs a PERSISTENT loop over 32-row tiles per wave --
// no LDS, no barriers -- and requests the next tile's neighbour index / coordinates while the current tile multiplies.
// Same MFMA order, bias/ReLU arithmetic and pooling as mlp_chain_kernel: bit-identical results.
// =====================================================================================================
template <int KB1, int KB2, int NB2, int NS>
__global__ __launch_bounds__(256) void sa_xyz_chain_kernel(const ChainParams Cin) {
    ChainParams C = Cin;
    C.a.rows = effective_rows(Cin.a);
    const MlpParams& P = C.a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, j = lane & 31;
    const float4 w0 = ldw(P.wpack, 1, 0, 0, lane);
    float4 w1[KB1], w2[NB2][KB2], b0[4], b1[4], b2[NB2][4];
#pragma unroll
    for (int kb = 0; kb < KB1; kb++) w1[kb] = ldw(C.wpack1, KB1, 0, kb, lane);
#pragma unroll
    for (int ob = 0; ob < NB2; ob++)
#pragma unroll
        for (int kb = 0; kb < KB2; kb++) w2[ob][kb] = ldw(C.wpack2, KB2, ob, kb, lane);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        b0[q] = P.bias ? ld4(P.bias + 8 * q + 4 * h) : zero4;
        b1[q] = C.bias1 ? ld4(C.bias1 + 8 * q + 4 * h) : zero4;
#pragma unroll
        for (int ob = 0; ob < NB2; ob++) b2[ob][q] = C.bias2 ? ld4(C.bias2 + ob * 32 + 8 * q + 4 * h) : zero4;
    }
    const long ntiles = (P.rows + 31) >> 5;
    const long stride = (long)gridDim.x * 4;
    long tile = (long)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    // neighbour index two tiles ahead, coordinates one tile ahead
    auto row_of = [&](long t) { long r = (t < ntiles ? t : ntiles - 1) * 32 + j; return r < P.rows ? r : P.rows - 1; };
    auto load_pt = [&](long r, int p, float (&v)[6]) {
        const int cen = (int)(r / NS), b = cen / P.M;
        const float* x = P.xyz + ((long)b * P.N + p) * 3;
        const float* q = P.new_xyz + (long)cen * 3;
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = q[0]; v[4] = q[1]; v[5] = q[2];
    };
    float cur[6], nxt[6];
    long r_cur = row_of(tile), r_nxt = row_of(tile + stride);
    int p_nxt = P.idx[r_nxt];
    load_pt(r_cur, P.idx[r_cur], cur);
    for (; tile < ntiles; tile += stride) {
        const long r_nn = row_of(tile + 2 * stride);
        const int p_nn = P.idx[r_nn];
        load_pt(r_nxt, p_nxt, nxt);
        const long row = tile * 32 + j;
        const bool valid = row < P.rows;
        // ---- layer 0: B operand = (dx, dy, dz, 0) in the h = 0 half, zeros in the h = 1 half (k = 4..7 >= K) ----
        const float dx = cur[0] - cur[3], dy = cur[1] - cur[4], dz = cur[2] - cur[5];
        f32x16 a0 = (f32x16){0};
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.x, h ? 0.f : dx, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.y, h ? 0.f : dy, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.z, h ? 0.f : dz, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0.w, 0.f, a0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float v0 = a0[4 * q] + b0[q].x, v1 = a0[4 * q + 1] + b0[q].y, v2 = a0[4 * q + 2] + b0[q].z, v3 = a0[4 * q + 3] + b0[q].w;
            if (P.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            a0[4 * q] = v0; a0[4 * q + 1] = v1; a0[4 * q + 2] = v2; a0[4 * q + 3] = v3;
        }
        // ---- layer 1 ----
        f32x16 a1 = (f32x16){0};
#pragma unroll
        for (int kb = 0; kb < KB1; kb++) {
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[kb].x, a0[4 * kb + 0], a1, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[kb].y, a0[4 * kb + 1], a1, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[kb].z, a0[4 * kb + 2], a1, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[kb].w, a0[4 * kb + 3], a1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float v0 = a1[4 * q] + b1[q].x, v1 = a1[4 * q + 1] + b1[q].y, v2 = a1[4 * q + 2] + b1[q].z, v3 = a1[4 * q + 3] + b1[q].w;
            if (C.relu1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            a1[4 * q] = v0; a1[4 * q + 1] = v1; a1[4 * q + 2] = v2; a1[4 * q + 3] = v3;
        }
        // ---- layer 2 ----
        f32x16 a2[NB2];
#pragma unroll
        for (int ob = 0; ob < NB2; ob++) a2[ob] = (f32x16){0};
#pragma unroll
        for (int kb = 0; kb < KB2; kb++) {
#pragma unroll
            for (int ob = 0; ob < NB2; ob++) a2[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[ob][kb].x, a1[4 * kb + 0], a2[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NB2; ob++) a2[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[ob][kb].y, a1[4 * kb + 1], a2[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NB2; ob++) a2[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[ob][kb].z, a1[4 * kb + 2], a2[ob], 0, 0, 0);
#pragma unroll
            for (int ob = 0; ob < NB2; ob++) a2[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[ob][kb].w, a1[4 * kb + 3], a2[ob], 0, 0, 0);
        }
#pragma unroll
        for (int ob = 0; ob < NB2; ob++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float v0 = a2[ob][4 * q] + b2[ob][q].x, v1 = a2[ob][4 * q + 1] + b2[ob][q].y;
                float v2 = a2[ob][4 * q + 2] + b2[ob][q].z, v3 = a2[ob][4 * q + 3] + b2[ob][q].w;
                if (C.relu2) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                a2[ob][4 * q] = v0; a2[ob][4 * q + 1] = v1; a2[ob][4 * q + 2] = v2; a2[ob][4 * q + 3] = v3;
            }
        chain_store<NB2>(C, a2, C.N2, row, valid, lane, h);
        // rotate the prefetch pipeline
#pragma unroll
        for (int e = 0; e < 6; e++) cur[e] = nxt[e];
        r_nxt = r_nn; p_nxt = p_nn;
    }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, int Nout, int K, int k_rot, int KB, int NB,
                                   float* __restrict__ wpack) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)NB * KB * 256;
    if (e >= total) return;
    int s = e & 3, jj = (e >> 2) & 31, hh = (e >> 7) & 1;
    long blk = e >> 8;
    int kb = (int)(blk % KB), nb = (int)(blk / KB);
    int n = nb * 32 + jj, kp = kb * 8 + 4 * hh + s;
    float val = 0.f;
    if (n < Nout && kp < K) {
        int ko = kp < K - k_rot ? kp + k_rot : kp - (K - k_rot);
        val = w[(long)n * K + ko];
    }
    wpack[e] = val;
}

__global__ void maxpool_rows_kernel(const float* __restrict__ in, int ld_in, long rows_out, int ns, int C,
                                    float* __restrict__ out, int ld_out, int col_off) {
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows_out * C) return;
    long r = e / C; int c = (int)(e - r * C);
    const float* p = in + r * ns * (long)ld_in + c;
    float m = p[0];
    for (int s = 1; s < ns; s++) m = fmaxf(m, p[(long)s * ld_in]);
    out[r * ld_out + col_off + c] = m;
}


static bool rows32_ok(int mode, const MlpParams& P);
static int launch_rows32(MlpParams& P, hipStream_t s);

static int launch_mlp(int mode, MlpParams& P, hipStream_t s) {
    PRCNN_REQUIRE(P.wpack && P.out, "prcnn_mlp: null weight/output pointer");
    PRCNN_REQUIRE(P.rows >= 0 && P.K > 0 && P.Nout > 0, "prcnn_mlp: bad shape rows=%ld K=%d Nout=%d", P.rows, P.K, P.Nout);
    PRCNN_REQUIRE(P.pool_ns == 0 || P.pool_ns == 16 || P.pool_ns == 32 || P.pool_ns == 64,
                  "prcnn_mlp: pool_ns=%d unsupported (use 16/32/64, or store + prcnn_maxpool_rows)", P.pool_ns);
    PRCNN_REQUIRE(P.pool_ns == 0 || P.rows % P.pool_ns == 0, "prcnn_mlp: rows %ld not a multiple of pool_ns %d", P.rows, P.pool_ns);
    PRCNN_REQUIRE(aligned16(P.wpack), "prcnn_mlp: wpack must be 16-byte aligned");
    if (P.rows == 0) return PRCNN_OK;
    P.KB = (P.K + 7) / 8;
    P.NB = (P.Nout + 31) / 32;
    // split-bf16 variant: 128 x 128 tiles while they give most CUs a workgroup (two are resident per CU), else 128 x 64.  Every
    // launch of a supported shape takes it, however few its rows: which arithmetic a layer is computed in must not depend on the
    // batch size (a frame's result is the same bits in a batch of 1 and of 32).  PRCNN_SPLIT_MIN_TILES (dev A/B) sends launches of
    // fewer tiles to the fp32 kernels, which have forms for few rows.
    const long split_tiles_wide = (long)prcnn_divup(P.rows, MLP_BM) * prcnn_divup(P.NB, 4), split_tiles_narrow = (long)prcnn_divup(P.rows, MLP_BM) * prcnn_divup(P.NB, 2);
    static const long split_min = getenv("PRCNN_SPLIT_MIN_TILES") ? atol(getenv("PRCNN_SPLIT_MIN_TILES")) : 0;
    // the hoisted grouped form (prcnn_mlp_group_split); PRCNN_GROUP_SPLIT=0: A/B switch back to the fp32 layer kernel
    static const bool group_split_on = !(getenv("PRCNN_GROUP_SPLIT") && atoi(getenv("PRCNN_GROUP_SPLIT")) == 0);
    const bool split_group = group_split_on && mode == MODE_GROUP && P.act == 1 && P.C == P.K && !P.addY;
    if (P.wsplit && (mode == MODE_PLAIN || split_group) && P.K % MLP_BK == 0 && P.vec_a && split_tiles_narrow >= split_min) {
        PRCNN_REQUIRE(aligned16(P.wsplit) && (P.split_terms == 3 || P.split_terms == 6), "prcnn_mlp: bad split image / terms=%d", P.split_terms);
        static const long split_wide_min = getenv("PRCNN_SPLIT_WIDE_MIN") ? atol(getenv("PRCNN_SPLIT_WIDE_MIN")) : 192;
        const bool wide = P.NB >= 4 && split_tiles_wide >= split_wide_min;
        dim3 grid(prcnn_divup(P.rows, MLP_BM), prcnn_divup(P.NB, wide ? 4 : 2));
        if (!P.seg_cnt) {
            P.wgm_cols = (int)grid.y;
            grid = dim3((unsigned)(prcnn_divup(grid.x, 8) * 8 * grid.y), 1);
        }
        // a launch sized for the CAPACITY of a compacted list (device-side row count): 2048 workgroups (four rounds of the 512 resident
        // ones, a multiple of the 8 XCDs) walk the live tiles instead of one workgroup per tile of capacity; PRCNN_BOUNDED_GRID=0: A/B
        static const bool bounded_on = !(getenv("PRCNN_BOUNDED_GRID") && atoi(getenv("PRCNN_BOUNDED_GRID")) == 0);
        const bool bounded = bounded_on && P.rows_dev && !P.seg_cnt && !P.addY && grid.x > 2048u;
        if (bounded) grid = dim3(2048u, 1);
#define SPL_LAUNCH(W, T)                                                                                                  \
    do {                                                                                                                  \
        if (split_group && bounded) hipLaunchKernelGGL((mlp_layer_s_kernel<MODE_GROUP, W, T, false, true>), grid, dim3(MLP_THREADS), 0, s, P);   \
        else if (split_group) hipLaunchKernelGGL((mlp_layer_s_kernel<MODE_GROUP, W, T, false, false>), grid, dim3(MLP_THREADS), 0, s, P);        \
        else if (P.addY) hipLaunchKernelGGL((mlp_layer_s_kernel<MODE_PLAIN, W, T, true, false>), grid, dim3(MLP_THREADS), 0, s, P);        \
        else if (bounded) hipLaunchKernelGGL((mlp_layer_s_kernel<MODE_PLAIN, W, T, false, true>), grid, dim3(MLP_THREADS), 0, s, P);  \
        else hipLaunchKernelGGL((mlp_layer_s_kernel<MODE_PLAIN, W, T, false, false>), grid, dim3(MLP_THREADS), 0, s, P);              \
    } while (0)
        if (wide) { if (P.split_terms == 6) SPL_LAUNCH(2, 6); else SPL_LAUNCH(2, 3); }
        else { if (P.split_terms == 6) SPL_LAUNCH(1, 6); else SPL_LAUNCH(1, 3); }
#undef SPL_LAUNCH
        PRCNN_LAUNCH_CHECK("prcnn_mlp (split-bf16)");
        return PRCNN_OK;
    }
    if (rows32_ok(mode, P)) return launch_rows32(P, s);
    // >= 97 output channels: 128x128 workgroup tile -- unless that leaves most of the 256 CUs without a workgroup
    // (few rows, e.g. FP3's 2048 known points): then the 128x64 tile doubles the number of workgroups
    // (a device-side row count means a compacted list: P.rows is its worst case, the live part is expected to be small)
    static const long wide_min = getenv("PRCNN_WIDE_MIN_TILES") ? atol(getenv("PRCNN_WIDE_MIN_TILES")) : 192;
    static const bool wide_lists = getenv("PRCNN_WIDE_LISTS") != nullptr;
    bool wide = P.NB >= 4 && (!P.rows_dev || wide_lists) && (long)prcnn_divup(P.rows, MLP_BM) * prcnn_divup(P.NB, 4) >= wide_min;
    dim3 grid(prcnn_divup(P.rows, MLP_BM), prcnn_divup(P.NB, wide ? 4 : 2));
    // v2 (B operand straight from L2, up to four workgroups per CU) is the default; PRCNN_LAYER_V1=1 is the A/B switch (same bits).
    static const bool force_v1 = getenv("PRCNN_LAYER_V1") != nullptr;
    const bool v2 = !force_v1;
    const bool fast = P.K % MLP_BK == 0 && P.vec_a && (mode == MODE_PLAIN || (mode == MODE_GROUP && P.act == 1 && P.C == P.K));
    if (v2 && fast && mode == MODE_PLAIN && wide && getenv("PRCNN_WIDE_MIN_TILES") == nullptr) {
        // four workgroups per CU = 1024 resident tiles: with >= 1024 wide tiles the narrow tile (twice as many, half as long)
        // runs in more, staggered rounds, so one round's store epilogue overlaps the next one's main loop (measured: 32768 x
        // 512 -> 512 171 -> 160 us, 131072 x 256 -> 256 173 -> 167 us); between 384 and 1023 wide tiles the wide tile's better
        // MFMA : LDS-read ratio wins (32768 x 512 -> 256: 74 vs 81 us)
        const long tiles_wide = (long)prcnn_divup(P.rows, MLP_BM) * prcnn_divup(P.NB, 4);
        wide = tiles_wide >= 384 && tiles_wide < 1024;
    }
    grid = dim3(prcnn_divup(P.rows, MLP_BM), prcnn_divup(P.NB, wide ? 4 : 2));
    if (v2 && mode == MODE_PLAIN && !P.seg_cnt && P.xcd_tpf == 0 && getenv("PRCNN_NO_WGM") == nullptr) {
        P.wgm_cols = (int)grid.y;
        grid = dim3((unsigned)(prcnn_divup(grid.x, 8) * 8 * grid.y), 1);
    }
    // (fast = straight-line main loop, see mlp_layer_b_kernel: whole 32-wide chunks, 16-byte rows; grouped: the hoisted form only)
#define MLP_LAUNCH_B(M, W, F)                                                                                            \
    do {                                                                                                                 \
        if (M == MODE_PLAIN && P.addY) hipLaunchKernelGGL((mlp_layer_b_kernel<M, W, F, (M == MODE_PLAIN)>), grid, dim3(MLP_THREADS), 0, s, P); \
        else hipLaunchKernelGGL((mlp_layer_b_kernel<M, W, F, false>), grid, dim3(MLP_THREADS), 0, s, P);              \
    } while (0)
#define MLP_LAUNCH(M)                                                                                         \
    do {                                                                                                      \
        if (v2) {                                                                                             \
            if (M != MODE_INTERP && fast) { if (wide) MLP_LAUNCH_B(M, 2, (M != MODE_INTERP)); else MLP_LAUNCH_B(M, 1, (M != MODE_INTERP)); } \
            else if (wide) MLP_LAUNCH_B(M, 2, false);                                                         \
            else MLP_LAUNCH_B(M, 1, false);                                                                   \
        } else if (wide) hipLaunchKernelGGL((mlp_layer_kernel<M, 2>), grid, dim3(MLP_THREADS), 0, s, P);      \
        else hipLaunchKernelGGL((mlp_layer_kernel<M, 1>), grid, dim3(MLP_THREADS), 0, s, P);                  \
    } while (0)
    if (mode == MODE_PLAIN) MLP_LAUNCH(MODE_PLAIN);
    else if (mode == MODE_GROUP) MLP_LAUNCH(MODE_GROUP);
    else MLP_LAUNCH(MODE_INTERP);
#undef MLP_LAUNCH
#undef MLP_LAUNCH_B
    PRCNN_LAUNCH_CHECK("prcnn_mlp");
    return PRCNN_OK;
}

PRCNN_API size_t prcnn_wpack_floats(int Nout, int K) {
    if (Nout <= 0 || K <= 0) return 0;
    return (size_t)((Nout + 31) / 32) * ((K + 7) / 8) * 256;
}

PRCNN_API int prcnn_pack_weight(const float* w, int Nout, int K, int k_rot, float* wpack, prcnn_stream_t stream) {
    PRCNN_REQUIRE(w && wpack, "prcnn_pack_weight: null pointer");
    PRCNN_REQUIRE(Nout > 0 && K > 0 && k_rot >= 0 && k_rot <= K, "prcnn_pack_weight: bad shape Nout=%d K=%d k_rot=%d", Nout, K, k_rot);
    int KB = (K + 7) / 8, NB = (Nout + 31) / 32;
    long total = (long)NB * KB * 256;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(prcnn_divup(total, 256)), dim3(256), 0, (hipStream_t)stream, w, Nout, K,
                       k_rot, KB, NB, wpack);
    PRCNN_LAUNCH_CHECK("prcnn_pack_weight");
    return PRCNN_OK;
}

PRCNN_API size_t prcnn_wsplit_bytes(int Nout, int K) {
    if (Nout <= 0 || K <= 0) return 0;
    return (size_t)((Nout + 31) / 32) * ((K + 15) / 16) * 3 * 1024;
}

PRCNN_API int prcnn_pack_weight_split(const float* w, int Nout, int K, int chain, void* wsplit, prcnn_stream_t stream) {
    PRCNN_REQUIRE(w && wsplit && aligned16(wsplit), "prcnn_pack_weight_split: null / misaligned pointer");
    PRCNN_REQUIRE(Nout > 0 && K > 0 && (chain == 0 || chain == 1), "prcnn_pack_weight_split: bad shape Nout=%d K=%d chain=%d", Nout, K, chain);
    const int KS = (K + 15) / 16, NB = (Nout + 31) / 32;
    hipLaunchKernelGGL(pack_weight_split_kernel, dim3(prcnn_divup((long)NB * KS * 64, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       Nout, K, KS, NB, chain, reinterpret_cast<uint4*>(wsplit));
    PRCNN_LAUNCH_CHECK("prcnn_pack_weight_split");
    return PRCNN_OK;
}

// Two-layer plain-row chain on the split kernels; shapes: K = 128, nout[0] = 128, nout[1] = 1 or 65..128 (the RPN heads).
// wchain[l]: prcnn_pack_weight_split(chain = 1) images; wpack1: the fp32 pack image of layer 1 (read by the single-channel output).
// PRCNN_EUNSUPPORTED for any other shape: the caller issues prcnn_mlp_chain_rows.
PRCNN_API int prcnn_mlp_chain_rows_split(const float* in, int ld_in, int64_t rows, int K, const void* const* wchain, const float* const* wpack,
                                         const float* const* bias, const int* nout, const int* relu, int terms, float* out, int ld_out,
                                         int col_off, prcnn_stream_t stream) {
    PRCNN_REQUIRE(in && wchain && wpack && bias && nout && relu && out, "prcnn_mlp_chain_rows_split: null pointer");
    PRCNN_REQUIRE(terms == 3 || terms == 6, "prcnn_mlp_chain_rows_split: terms=%d (3 or 6)", terms);
    PRCNN_REQUIRE(wpack[0] && wpack[1] && aligned16(wpack[0]) && aligned16(wpack[1]), "prcnn_mlp_chain_rows_split: the fp32 pack images of both layers are needed (non-finite rows, single-channel output)");
    const float* wpack1 = wpack[1];
    const bool ok = K == 128 && nout[0] == 128 && (nout[1] == 1 || (nout[1] > 64 && nout[1] <= 128)) && aligned16(in) && ld_in % 4 == 0 &&
                    ld_in >= K && wchain[0] && (nout[1] == 1 || wchain[1] != nullptr);
    if (!ok) return PRCNN_EUNSUPPORTED;
    PRCNN_REQUIRE(ld_out >= col_off + nout[1], "prcnn_mlp_chain_rows_split: ld_out=%d < col_off+Nout", ld_out);
    if (rows == 0) return PRCNN_OK;
    ChainParams C = {};
    MlpParams& P = C.a;
    P.rows = rows; P.K = K; P.in = in; P.ld_in = ld_in; P.bias = bias[0]; P.Nout = nout[0]; P.relu = relu[0];
    P.out = out; P.ld_out = ld_out; P.col_off = col_off; P.rows_unit = 1;
    P.wsplit = wchain[0]; P.split_terms = terms; P.wpack = wpack[0];
    C.wsplit1 = wchain[1]; C.wpack1 = wpack1; C.bias1 = bias[1]; C.N1 = nout[1]; C.relu1 = relu[1]; C.KB1 = 16; C.nlayers = 2;
    const dim3 grid(prcnn_divup(rows, 128));
    const hipStream_t s = (hipStream_t)stream;
#define SCH_LAUNCH(NB1)                                                                                        \
    do {                                                                                                       \
        /* plain rows, two layers: the lane-is-a-row kernel stays the default (its 32 rows are 16 contiguous KB that the 16 k-steps */ \
        /* re-read from L1: 168 vs 179 us for the reg head); PRCNN_CHAIN_COOP=2 selects the cooperative form (same bits)            */ \
        if (terms == 6 && chain_coop_forced()) hipLaunchKernelGGL((mlp_chain_c_kernel<MODE_PLAIN, NB1, 6>), grid, dim3(256), 0, s, C); \
        else if (terms == 6) hipLaunchKernelGGL((mlp_chain_s_kernel<MODE_PLAIN, NB1, 6>), grid, dim3(256), 0, s, C); \
        else hipLaunchKernelGGL((mlp_chain_s_kernel<MODE_PLAIN, NB1, 3>), grid, dim3(256), 0, s, C);            \
    } while (0)
    if (nout[1] == 1 && terms == 6 && chain_coop_on() && chain_persist_on() && (long)rows * ld_in < (1L << 30)) {
        const int rc = launch_chain_p<MODE_PLAIN, 1>(C, s);                     // (row offsets in floats fit 32 bits)
        if (rc) return rc;
    } else if (nout[1] == 1) SCH_LAUNCH(1);
    else if (nout[1] <= 96) SCH_LAUNCH(3);
    else SCH_LAUNCH(4);
#undef SCH_LAUNCH
    PRCNN_LAUNCH_CHECK("prcnn_mlp_chain_rows_split");
    return PRCNN_OK;
}

PRCNN_API int prcnn_mlp_rows_split(const float* in, int ld_in, int64_t rows, int K, const float* wpack, const void* wsplit, int terms,
                                   const float* bias, int Nout, int relu, float* out, int ld_out, int col_off, int pool_ns,
                                   const int32_t* rows_dev, int rows_unit, const int32_t* seg_cnt, int seg_rows, prcnn_stream_t stream) {
    PRCNN_REQUIRE(in && wsplit, "prcnn_mlp_rows_split: null pointer");
    PRCNN_REQUIRE(ld_in >= K && ld_out >= col_off + Nout, "prcnn_mlp_rows_split: bad strides ld_in=%d K=%d ld_out=%d", ld_in, K, ld_out);
    PRCNN_REQUIRE(terms == 3 || terms == 6, "prcnn_mlp_rows_split: terms=%d (3 or 6)", terms);
    MlpParams P = {};
    P.rows = rows; P.K = K; P.wpack = wpack; P.bias = bias; P.Nout = Nout; P.relu = relu;
    P.out = out; P.ld_out = ld_out; P.col_off = col_off; P.pool_ns = pool_ns;
    P.in = in; P.ld_in = ld_in;
    P.vec_a = aligned16(in) && (ld_in % 4 == 0);
    P.rows_dev = rows_dev; P.rows_unit = rows_unit > 0 ? rows_unit : 1;
    PRCNN_REQUIRE(!seg_cnt || (seg_rows > 0 && seg_rows % MLP_BM == 0 && rows % seg_rows == 0 && pool_ns == 0 && !rows_dev),
                  "prcnn_mlp_rows_split: seg_rows=%d must be a multiple of %d dividing rows (no pooling, no rows_dev)", seg_rows, MLP_BM);
    P.seg_cnt = seg_cnt; P.seg_rows = seg_rows;
    P.wsplit = wsplit; P.split_terms = terms;
    return launch_mlp(MODE_PLAIN, P, (hipStream_t)stream);
}

PRCNN_API int prcnn_mlp_rows_addinterp_split(const float* in, int ld_in, int K, const float* wpack, const void* wsplit, int terms,
                                             const float* bias, int Nout, int relu, const float* y_cl, int ld_y, const int32_t* idx3,
                                             const float* w3, int B, int n, int m, float* out, int ld_out, int col_off,
                                             prcnn_stream_t stream) {
    PRCNN_REQUIRE(in && y_cl && idx3 && w3 && wsplit, "prcnn_mlp_rows_addinterp_split: null pointer");
    PRCNN_REQUIRE(B >= 0 && n > 0 && m > 0 && ld_in >= K && ld_y >= Nout && ld_out >= col_off + Nout,
                  "prcnn_mlp_rows_addinterp_split: bad shape B=%d n=%d m=%d K=%d Nout=%d", B, n, m, K, Nout);
    PRCNN_REQUIRE(terms == 3 || terms == 6, "prcnn_mlp_rows_addinterp_split: terms=%d (3 or 6)", terms);
    MlpParams P = {};
    P.rows = (long)B * n; P.K = K; P.wpack = wpack; P.bias = bias; P.Nout = Nout; P.relu = relu;
    P.out = out; P.ld_out = ld_out; P.col_off = col_off; P.pool_ns = 0;
    P.in = in; P.ld_in = ld_in;
    P.vec_a = aligned16(in) && (ld_in % 4 == 0);
    P.addY = y_cl; P.ldY = ld_y; P.idx3 = idx3; P.w3 = w3; P.n = n; P.m = m;
    static const int addy_phase = getenv("PRCNN_ADDY_PHASE") ? atoi(getenv("PRCNN_ADDY_PHASE")) : 2;      // A/B switch: 0 = all in the epilogue
    P.addy_phase = addy_phase;
    P.rows_unit = 1;
    P.wsplit = wsplit; P.split_terms = terms;
    return launch_mlp(MODE_PLAIN, P, (hipStream_t)stream);
}

PRCNN_API int prcnn_mlp_rows(const float* in, int ld_in, int64_t rows, int K, const float* wpack, const float* bias,
                             int Nout, int relu, float* out, int ld_out, int col_off, int pool_ns,
                             const int32_t* rows_dev, int rows_unit, const int32_t* seg_cnt, int seg_rows,
                             prcnn_stream_t stream) {
    PRCNN_REQUIRE(in, "prcnn_mlp_rows: null input");
    PRCNN_REQUIRE(ld_in >= K && ld_out >= col_off + Nout, "prcnn_mlp_rows: bad strides ld_in=%d K=%d ld_out=%d", ld_in, K, ld_out);
    MlpParams P = {};
    P.rows = rows; P.K = K; P.wpack = wpack; P.bias = bias; P.Nout = Nout; P.relu = relu;
    P.out = out; P.ld_out = ld_out; P.col_off = col_off; P.pool_ns = pool_ns;
    P.in = in; P.ld_in = ld_in;
    P.vec_a = aligned16(in) && (ld_in % 4 == 0);
    P.rows_dev = rows_dev; P.rows_unit = rows_unit > 0 ? rows_unit : 1;
    PRCNN_REQUIRE(!seg_cnt || (seg_rows > 0 && seg_rows % MLP_BM == 0 && rows % seg_rows == 0 && pool_ns == 0 && !rows_dev),
                  "prcnn_mlp_rows: seg_rows=%d must be a multiple of %d dividing rows (no pooling, no rows_dev)", seg_rows, MLP_BM);
    P.seg_cnt = seg_cnt; P.seg_rows = seg_rows;
    return launch_mlp(MODE_PLAIN, P, (hipStream_t)stream);
}

// act_wx / act_bias non-null: hoisted first layer -- feat_cl is Z = W_f . feat per source point (C = width of that
// layer), the A row is relu(Z[idx] + act_wx . dxyz + act_bias) and K = C.
static int set_group_act(MlpParams& P, const float* act_wx, const float* act_bias, int C) {
    if (!act_wx && !act_bias) return PRCNN_OK;
    PRCNN_REQUIRE(act_wx && act_bias && C > 0, "prcnn_mlp_group: act_wx and act_bias must both be given (C > 0)");
    PRCNN_REQUIRE(aligned16(act_wx) && aligned16(act_bias) && C % 4 == 0,
                  "prcnn_mlp_group: hoisted mode needs 16-byte aligned act_wx/act_bias and C %% 4 == 0 (C=%d)", C);
    P.act = 1; P.act_wx = act_wx; P.act_bias = act_bias; P.K = C;
    return PRCNN_OK;
}

PRCNN_API int prcnn_mlp_rows_addinterp(const float* in, int ld_in, int K, const float* wpack, const float* bias, int Nout,
                                       int relu, const float* y_cl, int ld_y, const int32_t* idx3, const float* w3, int B,
                                       int n, int m, float* out, int ld_out, int col_off, prcnn_stream_t stream) {
    PRCNN_REQUIRE(in && y_cl && idx3 && w3, "prcnn_mlp_rows_addinterp: null pointer");
    PRCNN_REQUIRE(B >= 0 && n > 0 && m > 0 && ld_in >= K && ld_y >= Nout && ld_out >= col_off + Nout,
                  "prcnn_mlp_rows_addinterp: bad shape B=%d n=%d m=%d K=%d Nout=%d", B, n, m, K, Nout);
    MlpParams P = {};
    P.rows = (long)B * n; P.K = K; P.wpack = wpack; P.bias = bias; P.Nout = Nout; P.relu = relu;
    P.out = out; P.ld_out = ld_out; P.col_off = col_off; P.pool_ns = 0;
    P.in = in; P.ld_in = ld_in;
    P.vec_a = aligned16(in) && (ld_in % 4 == 0);
    P.addY = y_cl; P.ldY = ld_y; P.idx3 = idx3; P.w3 = w3; P.n = n; P.m = m;
    static const int addy_phase = getenv("PRCNN_ADDY_PHASE") ? atoi(getenv("PRCNN_ADDY_PHASE")) : 2;      // A/B switch: 0 = all in the epilogue
    P.addy_phase = addy_phase;
    return launch_mlp(MODE_PLAIN, P, (hipStream_t)stream);
}

PRCNN_API int prcnn_mlp_group(const float* xyz, const float* new_xyz, const int32_t* idx, const float* feat_cl,
                              int ld_feat, int B, int N, int M, int nsample, int C, const float* act_wx,
                              const float* act_bias, const float* wpack, const float* bias, int Nout, int relu,
                              float* out, int ld_out, int col_off, int pool_ns, const int32_t* groups_dev, prcnn_stream_t stream) {
    PRCNN_REQUIRE(xyz && idx, "prcnn_mlp_group: null pointer");
    PRCNN_REQUIRE(C == 0 || feat_cl, "prcnn_mlp_group: C=%d but feat_cl is null", C);
    PRCNN_REQUIRE(B >= 0 && N > 0 && M > 0 && nsample > 0 && C >= 0 && (C == 0 || ld_feat >= C),
                  "prcnn_mlp_group: bad shape B=%d N=%d M=%d ns=%d C=%d ld=%d", B, N, M, nsample, C, ld_feat);
    PRCNN_REQUIRE(ld_out >= col_off + Nout, "prcnn_mlp_group: ld_out=%d < col_off+Nout", ld_out);
    MlpParams P = {};
    P.rows = (long)B * M * nsample; P.K = C + 3; P.wpack = wpack; P.bias = bias; P.Nout = Nout; P.relu = relu;
    P.out = out; P.ld_out = ld_out; P.col_off = col_off; P.pool_ns = pool_ns;
    P.rows_dev = groups_dev; P.rows_unit = nsample;
    P.xyz = xyz; P.new_xyz = new_xyz; P.idx = idx; P.feat = feat_cl; P.ld_feat = ld_feat;
    P.N = N; P.M = M; P.ns = nsample; P.C = C;
    P.vec_a = C > 0 && aligned16(feat_cl) && (ld_feat % 4 == 0);
    int rc = set_group_act(P, act_wx, act_bias, C);
    if (rc) return rc;
    return launch_mlp(MODE_GROUP, P, (hipStream_t)stream);
}

// prcnn_mlp_group's hoisted form on the split-bf16 layer kernel (wsplit = prcnn_pack_weight_split image of the layer, terms 3 / 6; wpack:
// the fp32 image, read only by rows that hold inf / NaN).  C a multiple of 32, 16-byte aligned feature rows; otherwise, or with
// PRCNN_GROUP_SPLIT=0, the call runs the fp32 layer kernel exactly as prcnn_mlp_group does.
PRCNN_API int prcnn_mlp_group_split(const float* xyz, const float* new_xyz, const int32_t* idx, const float* feat_cl,
                                    int ld_feat, int B, int N, int M, int nsample, int C, const float* act_wx,
                                    const float* act_bias, const float* wpack, const void* wsplit, int terms, const float* bias, int Nout,
                                    int relu, float* out, int ld_out, int col_off, int pool_ns, const int32_t* groups_dev,
                                    prcnn_stream_t stream) {
    PRCNN_REQUIRE(xyz && idx && wsplit, "prcnn_mlp_group_split: null pointer");
    PRCNN_REQUIRE(C > 0 && feat_cl, "prcnn_mlp_group_split: the hoisted form needs features (C=%d)", C);
    PRCNN_REQUIRE(B >= 0 && N > 0 && M > 0 && nsample > 0 && ld_feat >= C,
                  "prcnn_mlp_group_split: bad shape B=%d N=%d M=%d ns=%d C=%d ld=%d", B, N, M, nsample, C, ld_feat);
    PRCNN_REQUIRE(terms == 3 || terms == 6, "prcnn_mlp_group_split: terms=%d (3 or 6)", terms);
    PRCNN_REQUIRE(ld_out >= col_off + Nout, "prcnn_mlp_group_split: ld_out=%d < col_off+Nout", ld_out);
    MlpParams P = {};
    P.rows = (long)B * M * nsample; P.K = C + 3; P.wpack = wpack; P.bias = bias; P.Nout = Nout; P.relu = relu;
    P.out = out; P.ld_out = ld_out; P.col_off = col_off; P.pool_ns = pool_ns;
    P.rows_dev = groups_dev; P.rows_unit = nsample;
    P.xyz = xyz; P.new_xyz = new_xyz; P.idx = idx; P.feat = feat_cl; P.ld_feat = ld_feat;
    P.N = N; P.M = M; P.ns = nsample; P.C = C;
    P.vec_a = C > 0 && aligned16(feat_cl) && (ld_feat % 4 == 0);
    int rc = set_group_act(P, act_wx, act_bias, C);
    if (rc) return rc;
    P.wsplit = wsplit; P.split_terms = terms;
    return launch_mlp(MODE_GROUP, P, (hipStream_t)stream);
}

// act_bias non-null (requires C1 == 0): hoisted first layer -- known_cl is Y = W . known per known point, the A row
// is relu(interp(Y) + act_bias).
static int set_interp_act(MlpParams& P, const float* act_bias, int C2, int C1) {
    if (!act_bias) return PRCNN_OK;
    PRCNN_REQUIRE(C1 == 0 && aligned16(act_bias) && C2 % 4 == 0,
                  "prcnn_mlp_interp: hoisted mode needs C1 == 0, aligned act_bias and C2 %% 4 == 0 (C2=%d C1=%d)", C2, C1);
    P.act = 2; P.act_bias = act_bias;
    return PRCNN_OK;
}

PRCNN_API int prcnn_mlp_interp(const float* known_cl, int ld_known, const int32_t* idx3, const float* w3,
                               const float* skip_cl, int ld_skip, int B, int n, int m, int C2, int C1,
                               const float* act_bias, const float* wpack, const float* bias, int Nout, int relu,
                               float* out, int ld_out, int col_off, prcnn_stream_t stream) {
    PRCNN_REQUIRE(known_cl && idx3 && w3, "prcnn_mlp_interp: null pointer");
    PRCNN_REQUIRE(C1 == 0 || skip_cl, "prcnn_mlp_interp: C1=%d but skip_cl is null", C1);
    PRCNN_REQUIRE(B >= 0 && n > 0 && m > 0 && C2 > 0 && C1 >= 0 && ld_known >= C2 && (C1 == 0 || ld_skip >= C1),
                  "prcnn_mlp_interp: bad shape B=%d n=%d m=%d C2=%d C1=%d", B, n, m, C2, C1);
    PRCNN_REQUIRE(ld_out >= col_off + Nout, "prcnn_mlp_interp: ld_out=%d < col_off+Nout", ld_out);
    MlpParams P = {};
    P.rows = (long)B * n; P.K = C2 + C1; P.wpack = wpack; P.bias = bias; P.Nout = Nout; P.relu = relu;
    P.out = out; P.ld_out = ld_out; P.col_off = col_off; P.pool_ns = 0;
    P.known = known_cl; P.idx3 = idx3; P.w3 = w3; P.skip = skip_cl; P.ld_known = ld_known; P.ld_skip = ld_skip;
    P.n = n; P.m = m; P.C2 = C2; P.C1 = C1;
    P.vec_a = aligned16(known_cl) && (ld_known % 4 == 0);
    P.vec_b = C1 > 0 && aligned16(skip_cl) && (ld_skip % 4 == 0) && (C2 % 4 == 0);
    int rc = set_interp_act(P, act_bias, C2, C1);
    if (rc) return rc;
    return launch_mlp(MODE_INTERP, P, (hipStream_t)stream);
}

PRCNN_API int prcnn_maxpool_rows(const float* in, int ld_in, int64_t rows_out, int ns, int C, float* out, int ld_out,
                                 int col_off, prcnn_stream_t stream) {
    PRCNN_REQUIRE(in && out, "prcnn_maxpool_rows: null pointer");
    PRCNN_REQUIRE(rows_out >= 0 && ns > 0 && C > 0 && ld_in >= C && ld_out >= col_off + C, "prcnn_maxpool_rows: bad shape");
    if (rows_out == 0) return PRCNN_OK;
    hipLaunchKernelGGL(maxpool_rows_kernel, dim3(prcnn_divup(rows_out * C, 256)), dim3(256), 0, (hipStream_t)stream, in,
                       ld_in, (long)rows_out, ns, C, out, ld_out, col_off);
    PRCNN_LAUNCH_CHECK("prcnn_maxpool_rows");
    return PRCNN_OK;
}

// ---- chain launcher -----------------------------------------------------------------------------
template <int MODE, int NB0, int NB1, int NB2>
static void launch_chain(const ChainParams& C, hipStream_t s) {
    dim3 grid(prcnn_divup(C.a.rows, 128));
    hipLaunchKernelGGL((mlp_chain_kernel<MODE, NB0, NB1, NB2>), grid, dim3(256), 0, s, C);
}
template <int MODE, int NB0, int NB1, int NB2>
static void launch_chain_fast(const ChainParams& C, hipStream_t s) {
    dim3 grid(prcnn_divup(C.a.rows, 128));
    hipLaunchKernelGGL((mlp_chain_fast_kernel<MODE, NB0, NB1, NB2>), grid, dim3(256), 0, s, C);
}

// the straight-line variant applies when nothing in the layer-0 loop needs a bounds check (see mlp_chain_fast_kernel)
static bool chain_fast_ok(int mode, const ChainParams& C, int n0, int n1, int n2) {
    const MlpParams& P = C.a;
    if (P.K % 8 != 0 || P.K > FAST_MAX_K || P.K < 16 || !P.vec_a || P.addY) return false;
    const int G0 = n0 == 3 ? 4 : CH_STAGE_TILES / n0;
    if ((P.K / 8) % G0 != 0) return false;
    if (n1 > 0 && P.Nout != n0 * 32) return false;                 // KB of layer 1 == 4 * NB0
    if (n2 > 0 && C.N1 != n1 * 32) return false;
    if (n1 > 0 && (n0 * 4) % (n1 == 3 ? 4 : CH_STAGE_TILES / n1) != 0) return false;
    if (n2 > 0 && (n1 * 4) % (n2 == 3 ? 4 : CH_STAGE_TILES / n2) != 0) return false;
    if (mode == MODE_GROUP) return P.act == 1 && P.K == P.C && P.act_wx && P.act_bias;
    if (mode == MODE_INTERP) return P.act == 2 && P.C1 == 0 && P.K == P.C2 && P.act_bias;
    return mode == MODE_PLAIN;
}

static int nb32(int n) { return (n + 31) / 32; }

// =====================================================================================================
// Two-layer STACK for short row lists (round 2): the wide SA levels (SA3 / SA4: 128 -> 196 -> 256, 256 -> 256|384 -> 512) run
// on padding-free flat lists of only a few thousand rows, where the layer kernel's 128-row tiles leave most CUs idle and every
// launch is a chain of exposed latencies (two launches of 20-45 us per scale for < 1.3 GFLOP).  Here a workgroup owns
// 32 rows and carries them through BOTH layers: the activated-gather rows are built once into LDS, the four waves split each
// layer's output columns (32-column blocks), layer A's result goes bias + ReLU into a second LDS tile and is layer B's A
// operand; the weights stream from L2 straight into the MFMA B registers (4-slot ring, as mlp_layer_b_kernel).  With few
// row tiles the layer-B column blocks are additionally split over gridDim.y workgroups (each recomputes the cheap layer A),
// so that ~256 workgroups exist whatever the list length.
// Arithmetic per output element is the layer kernels': k ascending in the same MFMA steps, bias then ReLU -- bit-identical.
// =====================================================================================================
#define ST_ROWS 32
#define ST_MAX_K0 256
#define ST_MAX_NB0 12
#define ST_RING 8                // k-blocks of weights in flight per wave (one wave per SIMD: registers are plentiful)
template <int NBW0>
__global__ __launch_bounds__(256, 1) void mlp_stack2_kernel(const ChainParams Cin) {
    MlpParams P = Cin.a;
    P.rows = effective_rows(Cin.a);
    // layer-B column groups (4 blocks, one per wave) are dealt to `ysplit` workgroups per row tile: 1 when the list alone
    // fills the chip, up to Cin.stack_split when it is short -- decided HERE from the device-side row count (the host only
    // knows the padded bound: 8192 rows for the 2048 that SA4 really has).  The live workgroups are the FIRST tiles * ysplit
    // of the 1-D grid, so that the dispatcher spreads them over XCDs and CUs like any dense launch; the rest leave.
    const int tiles_eff = (int)((P.rows + ST_ROWS - 1) / ST_ROWS);
    const int ysplit = min(Cin.stack_split, max(1, (256 + tiles_eff - 1) / max(tiles_eff, 1)));
    extern __shared__ __attribute__((aligned(16))) float st_lds[];
    // (the grid is capped: the padded bound can be 16 x the live list, and thousands of workgroups that only leave cost more than
    //  the loop does)
    for (long unit = blockIdx.x; unit < (long)tiles_eff * ysplit; unit += gridDim.x) {
    const int tile = (int)(unit / ysplit), ysl = (int)(unit - (long)tile * ysplit);
    const long row0 = (long)tile * ST_ROWS;
    const int lda = P.K + 4;                               // row strides (floats): +4 keeps the ds_read_b128 of 16 rows conflict-free
    const int n0pad = P.NB * 32, ldb = n0pad + 4;
    float* actA = st_lds;                                  // [32][K0 + 4]
    float* actB = st_lds + ST_ROWS * lda;                  // [32][NB0 * 32 + 4]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;

    // ---- weight rings: layer A's and the first layer-B column group's, requested first
    const int NB1 = (Cin.N1 + 31) / 32, NCG = (NB1 + 3) / 4;
    const int KBa = (PRCNN_ABL & 32) ? 1 : P.KB, KBb = (PRCNN_ABL & 64) ? 1 : Cin.KB1;
    int nbsA[4];
    const float* bpA[4];
    float4 bqA[ST_RING][4], bqB[ST_RING][4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        nbsA[q] = wave + 4 * q;
        bpA[q] = P.wpack + ((long)min(nbsA[q], P.NB - 1) * P.KB) * 256 + lane * 4;
    }
    const float* bpB[4];
    auto prime_b = [&](int cg) {
        bpB[0] = Cin.wpack1 + ((long)min(cg * 4 + wave, NB1 - 1) * Cin.KB1) * 256 + lane * 4;
#pragma unroll
        for (int g = 0; g < ST_RING; g++) bqB[g][0] = ld4(bpB[0] + (long)min(g, KBb - 1) * 256);
    };
#pragma unroll
    for (int g = 0; g < ST_RING; g++) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (q < NBW0) bqA[g][q] = ld4(bpA[q] + (long)min(g, KBa - 1) * 256);
    }
    bpB[1] = bpB[2] = bpB[3] = Cin.wpack1;
    prime_b(ysl);

    // ---- activated-gather rows -> actA: thread = (row tid / 8, 16-byte column piece tid % 8 of every 32-channel chunk)
    {
        const int r = tid >> 3, c4 = tid & 7;
        long grow = row0 + r;
        if (grow >= P.rows) grow = P.rows - 1;             // clamped: never stored
        RowMeta<MODE_GROUP> meta;
        make_meta<MODE_GROUP>(P, grow, meta);
        const float dx = meta.dx, dy = meta.dy, dz = meta.dz;
        // all of the row's gathered pieces first (independent L2 / HBM round trips in flight together), then the arithmetic
        float4 zr[ST_MAX_K0 / 32];
#pragma unroll
        for (int u = 0; u < ST_MAX_K0 / 32; u++) {
            const int k = c4 * 4 + 32 * u;
            zr[u] = (k < P.K && !(PRCNN_ABL & 16)) ? ld4(P.feat + meta.off + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < ST_MAX_K0 / 32; u++) {
            const int k = c4 * 4 + 32 * u;
            if (k >= P.K) break;
            const float4 z = zr[u];
            const float4 aw0 = ld4(P.act_wx + (long)k * 3), aw1 = ld4(P.act_wx + (long)k * 3 + 4), aw2 = ld4(P.act_wx + (long)k * 3 + 8);
            const float4 ab = ld4(P.act_bias + k);
            float4 v;                                       // == the layer kernels' activated gather, same expression
            v.x = fmaxf(z.x + (aw0.x * dx + aw0.y * dy + aw0.z * dz) + ab.x, 0.f);
            v.y = fmaxf(z.y + (aw0.w * dx + aw1.x * dy + aw1.y * dz) + ab.y, 0.f);
            v.z = fmaxf(z.z + (aw1.z * dx + aw1.w * dy + aw2.x * dz) + ab.z, 0.f);
            v.w = fmaxf(z.w + (aw2.y * dx + aw2.z * dy + aw2.w * dz) + ab.w, 0.f);
            *reinterpret_cast<float4*>(actA + r * lda + k) = v;
        }
    }

    // one layer: this wave's NBW column blocks over KB k-blocks of the A tile `act` (stride ld); the weight ring `bq` was
    // primed by ring_fill -- at kernel entry for BOTH layers, so that the cold-weight round trips (these launches run once per
    // step, between kernels that sweep the L2) overlap the row gather instead of heading each layer
    auto run_layer = [&](auto& acc, float4 (&bq)[ST_RING][4], const float* const (&bptr)[4], int nbw, int KB, const float* act, int ld) {
        auto load_b = [&](int g, int slot) {
            const long off = (long)min(g, KB - 1) * 256;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (q < nbw) bq[slot][q] = ld4(bptr[q] + off);
        };
        const float* a_base = act + j * ld + 4 * h;
        float4 a_next = *reinterpret_cast<const float4*>(a_base);
        for (int kb0 = 0; kb0 < KB; kb0 += ST_RING) {
#pragma unroll
            for (int u = 0; u < ST_RING; u++) {
                const int kb = kb0 + u;
                if (kb < KB) {
                    // one wave per SIMD here: the next k-block's A operand is requested before this one's MFMAs, or its LDS
                    // latency would show between every 4-16 MFMAs
                    const float4 a = a_next;
                    a_next = *reinterpret_cast<const float4*>(a_base + min(kb + 1, KB - 1) * 8);
#pragma unroll
                    for (int q = 0; q < 4; q++) if (q < nbw) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[u][q].x, acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; q++) if (q < nbw) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[u][q].y, acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; q++) if (q < nbw) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[u][q].z, acc[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 4; q++) if (q < nbw) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[u][q].w, acc[q], 0, 0, 0);
                }
                if (!(PRCNN_ABL & 128)) load_b(kb + ST_RING, u);
            }
        }
    };

    __syncthreads();
    // ---- layer A: column blocks wave, wave + 4, ... of N0
    {
        f32x16 acc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = (f32x16){0};
        run_layer(acc, bqA, bpA, NBW0, KBa, actA, lda);
#pragma unroll
        for (int q = 0; q < NBW0; q++) {
            const int nb = nbsA[q];
            if (nb >= P.NB) continue;                       // (wave-uniform)
            const int n = nb * 32 + j;
            const float bias = (P.bias && n < P.Nout) ? P.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[q][r] + bias;
                if (P.relu) v = fmaxf(v, 0.f);
                actB[rin * ldb + n] = n < P.Nout ? v : 0.f;  // columns past N0 are exact zeros (layer B's K padding)
            }
        }
    }
    __syncthreads();
    // ---- layer B: column groups ysl, ysl + ysplit, ...; one 32-column block per wave and pass
    // (two groups per pass -- a second accumulator chain -- was tried: the registers it takes cost the second resident
    //  workgroup, 37 -> 47 us on a 16384-row list)
    for (int cg = ysl; cg < NCG; cg += ysplit) {
        if (cg != ysl) prime_b(cg);
        f32x16 acc[4];
        acc[0] = (f32x16){0};
        run_layer(acc, bqB, bpB, 1, KBb, actB, ldb);
        const int nb = cg * 4 + wave;
        if (nb >= NB1) continue;                            // (wave-uniform)
        const int n = nb * 32 + j;
        const bool n_ok = n < Cin.N1;
        const float bias = (Cin.bias1 && n_ok) ? Cin.bias1[n] : 0.f;
        if (row0 + 32 <= P.rows && nb * 32 + 32 <= Cin.N1) {
            // whole block inside the output: unguarded stores, issued back to back (a bounds branch per element makes the compiler
            // open every store with s_waitcnt vmcnt(0) -- the counter includes stores on gfx9: one round trip per store)
            float* o = P.out + (row0 + 4 * h) * P.ld_out + P.col_off + n;
            const long ld = P.ld_out;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float v = acc[0][r] + bias;
                v = Cin.relu1 ? fmaxf(v, 0.f) : v;
                o[((r & 3) + 8 * (r >> 2)) * ld] = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
                const long g = row0 + rin;
                float v = acc[0][r] + bias;
                if (Cin.relu1) v = fmaxf(v, 0.f);
                if (n_ok && g < P.rows) P.out[g * P.ld_out + P.col_off + n] = v;
            }
        }
    }
    __syncthreads();                                        // the LDS tiles are rebuilt by the next unit
    }
}

// =====================================================================================================
// SHORT plain layers (round 2): a long-K layer on a few thousand rows (FP3's hoisted 2048 x 1024 -> 512 product) gives the
// 128-row layer kernel 128 workgroups of 32 chunks each -- half the chip, a chain of exposed latencies (53 us for 2.1 GFLOP).
// Same shape of solution as the stack kernel: a workgroup owns 32 rows, holds them whole in LDS (K <= 1024: 131 KB), its
// four waves take one 32-column block each per pass with the weights streamed from L2 into the register ring, and the column
// groups of a row tile are dealt to up to Nout/128 workgroups.  k ascends from a zero accumulator in the layer kernels' MFMA
// steps, bias then ReLU: the same bits.
// =====================================================================================================
#define R32_MAX_K 1024
__global__ __launch_bounds__(256, 1) void mlp_rows32_kernel(const MlpParams Pin, int split_max) {
    MlpParams P = Pin;
    P.rows = effective_rows(Pin);
    const int tiles_eff = (int)((P.rows + ST_ROWS - 1) / ST_ROWS);
    const int ysplit = min(split_max, max(1, (256 + tiles_eff - 1) / max(tiles_eff, 1)));
    extern __shared__ __attribute__((aligned(16))) float st_lds[];
    const int lda = P.K + 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    const int NCG = (P.NB + 3) / 4;
    for (long unit = blockIdx.x; unit < (long)tiles_eff * ysplit; unit += gridDim.x) {
        const int tile = (int)(unit / ysplit), ysl = (int)(unit - (long)tile * ysplit);
        const long row0 = (long)tile * ST_ROWS;
        const float* bp;
        float4 bq[ST_RING];
        auto prime = [&](int cg) {
            bp = P.wpack + ((long)min(cg * 4 + wave, P.NB - 1) * P.KB) * 256 + lane * 4;
#pragma unroll
            for (int g = 0; g < ST_RING; g++) bq[g] = ld4(bp + (long)min(g, P.KB - 1) * 256);
        };
        prime(ysl);
        // the 32 rows -> LDS: thread = (row tid / 8, 16-byte piece tid % 8 of every 32-float chunk), sixteen pieces in flight
        {
            const int r = tid >> 3, c4 = tid & 7;
            long grow = row0 + r;
            if (grow >= P.rows) grow = P.rows - 1;          // clamped: never stored
            const float* src = P.in + grow * (long)P.ld_in + c4 * 4;
            float* dst = st_lds + r * lda + c4 * 4;
            for (int k0 = 0; k0 < P.K; k0 += 512) {
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) v[u] = (k0 + 32 * u + c4 * 4 < P.K) ? ld4(src + k0 + 32 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 16; u++)
                    if (k0 + 32 * u + c4 * 4 < P.K) *reinterpret_cast<float4*>(dst + k0 + 32 * u) = v[u];
            }
        }
        __syncthreads();
        const float* a_base = st_lds + j * lda + 4 * h;
        for (int cg = ysl; cg < NCG; cg += ysplit) {
            if (cg != ysl) prime(cg);
            f32x16 acc = (f32x16){0};
            float4 a_next = *reinterpret_cast<const float4*>(a_base);
            for (int kb0 = 0; kb0 < P.KB; kb0 += ST_RING) {
#pragma unroll
                for (int u = 0; u < ST_RING; u++) {
                    const int kb = kb0 + u;
                    if (kb < P.KB) {
                        const float4 a = a_next;
                        a_next = *reinterpret_cast<const float4*>(a_base + min(kb + 1, P.KB - 1) * 8);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[u].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[u].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[u].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[u].w, acc, 0, 0, 0);
                    }
                    bq[u] = ld4(bp + (long)min(kb + ST_RING, P.KB - 1) * 256);
                }
            }
            const int nb = cg * 4 + wave;
            if (nb >= P.NB) continue;                       // (wave-uniform)
            const int n = nb * 32 + j;
            const bool n_ok = n < P.Nout;
            const float bias = (P.bias && n_ok) ? P.bias[n] : 0.f;
            if (row0 + 32 <= P.rows && nb * 32 + 32 <= P.Nout) {      // whole block inside the output: unguarded stores (see mlp_stack2_kernel)
                float* o = P.out + (row0 + 4 * h) * P.ld_out + P.col_off + n;
                const long ld = P.ld_out;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float v = acc[r] + bias;
                    v = P.relu ? fmaxf(v, 0.f) : v;
                    o[((r & 3) + 8 * (r >> 2)) * ld] = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const long g = row0 + rin;
                    float v = acc[r] + bias;
                    if (P.relu) v = fmaxf(v, 0.f);
                    if (n_ok && g < P.rows) P.out[g * P.ld_out + P.col_off + n] = v;
                }
            }
        }
        __syncthreads();
    }
}

static bool rows32_ok(int mode, const MlpParams& P) {
    static const bool off = getenv("PRCNN_NO_ROWS32") != nullptr;    // A/B switch (the layer kernel gives the same bits)
    return !off && mode == MODE_PLAIN && !P.addY && P.pool_ns == 0 && !P.seg_cnt && P.vec_a && P.K % 8 == 0 && P.K >= 256 &&
           P.K <= R32_MAX_K && P.Nout >= 128 && P.rows <= 4096;
}

static int launch_rows32(MlpParams& P, hipStream_t s) {
    static PrcnnLdsLimit attr;
    if (!attr.raise((const void*)mlp_rows32_kernel, 144 * 1024))
        return prcnn_fail(PRCNN_EHIP, "prcnn_mlp(rows32): cannot raise the dynamic LDS limit");
    const int split_max = prcnn_divup(P.NB, 4);
    const size_t lds = (size_t)ST_ROWS * (P.K + 4) * sizeof(float);
    hipLaunchKernelGGL(mlp_rows32_kernel, dim3((unsigned)min((long)prcnn_divup(P.rows, ST_ROWS) * split_max, 2048L)), dim3(256), lds, s,
                       P, split_max);
    PRCNN_LAUNCH_CHECK("prcnn_mlp(rows32)");
    return PRCNN_OK;
}

// shapes the stack kernel takes (hoisted grouped form on flat row lists, no pooling)
static bool stack2_ok(int mode, const ChainParams& C) {
    const MlpParams& P = C.a;
    return mode == MODE_GROUP && C.nlayers == 2 && P.act == 1 && P.pool_ns == 0 && P.ns == 1 && P.C == P.K && P.K % 8 == 0 &&
           P.K <= ST_MAX_K0 && P.vec_a && (P.Nout > 128 || C.N1 > 128) && nb32(P.Nout) <= ST_MAX_NB0 && nb32(C.N1) <= 16;
}

template <int NBW0>
static void launch_stack2(const ChainParams& C, hipStream_t s) {
    const size_t lds = (size_t)ST_ROWS * ((C.a.K + 4) + (nb32(C.a.Nout) * 32 + 4)) * sizeof(float);
    hipLaunchKernelGGL((mlp_stack2_kernel<NBW0>), dim3((unsigned)min((long)prcnn_divup(C.a.rows, ST_ROWS) * C.stack_split, 2048L)), dim3(256), lds, s, C);
}

static bool chain_instance_exists(int mode, int n0, int n1, int n2) {
    struct { int m, a, b, c; } T[] = {{MODE_GROUP, 1, 1, 1}, {MODE_GROUP, 1, 1, 2}, {MODE_GROUP, 2, 2, 4}, {MODE_GROUP, 2, 3, 4},
                                      {MODE_GROUP, 2, 4, 0}, {MODE_GROUP, 3, 4, 0},          // hoisted SA2 stacks
                                      {MODE_INTERP, 4, 4, 0}, {MODE_INTERP, 4, 0, 0},        // FP0 / hoisted FP0
                                      {MODE_PLAIN, 4, 4, 0}, {MODE_PLAIN, 4, 1, 0}, {MODE_PLAIN, 4, 3, 0}};
    for (auto& t : T)
        if (t.m == mode && t.a == n0 && t.b == n1 && t.c == n2) return true;
    return false;
}

PRCNN_API int prcnn_mlp_chain_supported(int mode, int nlayers, const int* nout, int pool_ns) {
    if (!nout || nlayers < 1 || nlayers > 3) return 0;
    if (!(pool_ns == 0 || pool_ns == 16 || pool_ns == 32)) return 0;
    // two wide layers on an un-pooled grouped list: the stack kernel (hoisted form, nsample 1 -- checked again at dispatch)
    if (mode == MODE_GROUP && nlayers == 2 && pool_ns == 0 && nout[0] > 0 && nout[1] > 0 && (nout[0] > 128 || nout[1] > 128) &&
        nb32(nout[0]) <= ST_MAX_NB0 && nb32(nout[1]) <= 16 && getenv("PRCNN_NO_STACK") == nullptr)
        return 1;
    for (int l = 0; l < nlayers; l++)
        if (nout[l] <= 0 || nout[l] > 128) return 0;
    return chain_instance_exists(mode, nb32(nout[0]), nlayers > 1 ? nb32(nout[1]) : 0, nlayers > 2 ? nb32(nout[2]) : 0) ? 1 : 0;
}

// returns PRCNN_EUNSUPPORTED when no instance matches
static int dispatch_chain(int mode, ChainParams& C, hipStream_t s) {
    MlpParams& P = C.a;
    P.KB = (P.K + 7) / 8;
    P.NB = nb32(P.Nout);
    const int n0 = nb32(P.Nout), n1 = C.nlayers > 1 ? nb32(C.N1) : 0, n2 = C.nlayers > 2 ? nb32(C.N2) : 0;
    if (C.nlayers > 1) C.KB1 = (P.Nout + 7) / 8;
    if (C.nlayers > 2) C.KB2 = (C.N1 + 7) / 8;
    if (P.rows == 0) return PRCNN_OK;
    if (stack2_ok(mode, C)) {
        // stack_split = the most workgroups a row tile's layer-B column groups may be dealt to; the kernel picks the split
        // from the device-side row count
        const int nbw0 = prcnn_divup(n0, 4);
        C.stack_split = prcnn_divup(n1, 4);
        static PrcnnLdsLimit attr[3];
#define STACK_CASE(A)                                                                                                    \
        if (nbw0 == A) {                                                                                                 \
            if (!attr[A - 1].raise((const void*)mlp_stack2_kernel<A>, 96 * 1024))                                        \
                return prcnn_fail(PRCNN_EHIP, "prcnn_mlp_chain(stack): cannot raise the dynamic LDS limit");             \
            launch_stack2<A>(C, s);                                                                                      \
            PRCNN_LAUNCH_CHECK("prcnn_mlp_chain(stack)");                                                                \
            return PRCNN_OK;                                                                                             \
        }
        STACK_CASE(1) STACK_CASE(2) STACK_CASE(3)
#undef STACK_CASE
    }
    // SA level 0: xyz-only rows, three narrow layers, pooled -- persistent register-weight kernel
    // (pooled groups, or -- nsample 1, no pooling -- the flat row list of the padding-free path: the same kernel writes rows)
    if (mode == MODE_GROUP && P.C == 0 && !P.act && P.K == 3 && C.nlayers == 3 && P.new_xyz &&
        (P.pool_ns == P.ns || (P.pool_ns == 0 && P.ns == 1)) && C.N2 % 4 == 0 &&
        getenv("PRCNN_NO_SA0") == nullptr) {              // (A/B switch; the generic chain kernel gives the same bits)
        // persistent: one resident workgroup per occupancy slot (256 CUs x 3 or 2 workgroups at 115 / 243 registers)
#define SA0_CASE(W0, W1, NBL, NSV)                                                                                        \
        if (P.Nout == W0 && C.N1 == W1 && n2 == NBL && P.ns == NSV) {                                                   \
            const int grid = (int)min((long)(NBL == 1 ? 768 : 512), (long)prcnn_divup(P.rows, 128));                     \
            hipLaunchKernelGGL((sa_xyz_chain_kernel<W0 / 8, W1 / 8, NBL, NSV>), dim3(grid), dim3(256), 0, s, C);          \
            PRCNN_LAUNCH_CHECK("prcnn_mlp_chain(sa0)");                                                                  \
            return PRCNN_OK;                                                                                             \
        }
        SA0_CASE(16, 16, 1, 16)
        SA0_CASE(32, 32, 2, 32)
        SA0_CASE(16, 16, 1, 1)
        SA0_CASE(32, 32, 2, 1)
#undef SA0_CASE
    }
    // Opt-in (PRCNN_PERSISTENT_CHAIN=1): 6-12 % faster per launch with ONE batch in flight, but a persistent workgroup
    // holds its CU's LDS for the whole kernel, which starves the other in-flight batches' kernels (FPS sort, layer tiles):
    // measured -6 % RPN throughput at 3 batches in flight, so the default keeps the per-tile workgroups.
    if (chain_fast_ok(mode, C, n0, n1, n2) && !P.seg_cnt && getenv("PRCNN_PERSISTENT_CHAIN") != nullptr) {
        // persistent form: weights of the whole stack resident in LDS, one 8-wave workgroup per CU
#define PERS_CASE(M, KB0V, A, B, CC)                                                                                         \
        if (mode == M && P.KB == KB0V && n0 == A && n1 == B && n2 == CC) {                                                    \
            constexpr size_t lds = pers_lds_bytes<M, KB0V, A, B, CC>();                                                       \
            static PrcnnLdsLimit attr;                                                                                       \
            if (!attr.raise((const void*)mlp_chain_pers_kernel<M, KB0V, A, B, CC>, (int)lds))                                \
                return prcnn_fail(PRCNN_EHIP, "prcnn_mlp_chain: cannot raise the dynamic LDS limit");                        \
            const int grid = (int)min((long)256, (long)prcnn_divup(P.rows, 32 * PERS_WAVES));                                \
            hipLaunchKernelGGL((mlp_chain_pers_kernel<M, KB0V, A, B, CC>), dim3(grid), dim3(PERS_WAVES * 64), lds, s, C);     \
            PRCNN_LAUNCH_CHECK("prcnn_mlp_chain(persistent)");                                                               \
            return PRCNN_OK;                                                                                                 \
        }
        PERS_CASE(MODE_GROUP, 8, 2, 4, 0)
        PERS_CASE(MODE_GROUP, 8, 3, 4, 0)
        PERS_CASE(MODE_INTERP, 16, 4, 0, 0)
        PERS_CASE(MODE_PLAIN, 16, 4, 1, 0)
        PERS_CASE(MODE_PLAIN, 16, 4, 3, 0)
#undef PERS_CASE
    }
    if (chain_fast_ok(mode, C, n0, n1, n2) && getenv("PRCNN_NO_FAST_CHAIN") == nullptr) {      // (A/B switch, same bits)
#define FAST_CASE(M, A, B, CC) if (mode == M && n0 == A && n1 == B && n2 == CC) { launch_chain_fast<M, A, B, CC>(C, s); PRCNN_LAUNCH_CHECK("prcnn_mlp_chain(fast)"); return PRCNN_OK; }
        FAST_CASE(MODE_GROUP, 2, 4, 0)
        FAST_CASE(MODE_GROUP, 3, 4, 0)
        FAST_CASE(MODE_INTERP, 4, 0, 0)
        FAST_CASE(MODE_PLAIN, 4, 1, 0)
        FAST_CASE(MODE_PLAIN, 4, 3, 0)
        FAST_CASE(MODE_PLAIN, 4, 4, 0)
#undef FAST_CASE
    }
#define CHAIN_CASE(M, A, B, CC) if (mode == M && n0 == A && n1 == B && n2 == CC) { launch_chain<M, A, B, CC>(C, s); PRCNN_LAUNCH_CHECK("prcnn_mlp_chain"); return PRCNN_OK; }
    CHAIN_CASE(MODE_GROUP, 1, 1, 1)
    CHAIN_CASE(MODE_GROUP, 1, 1, 2)
    CHAIN_CASE(MODE_GROUP, 2, 2, 4)
    CHAIN_CASE(MODE_GROUP, 2, 3, 4)
    CHAIN_CASE(MODE_GROUP, 2, 4, 0)
    CHAIN_CASE(MODE_GROUP, 3, 4, 0)
    CHAIN_CASE(MODE_INTERP, 4, 4, 0)
    CHAIN_CASE(MODE_INTERP, 4, 0, 0)
    CHAIN_CASE(MODE_PLAIN, 4, 4, 0)
    CHAIN_CASE(MODE_PLAIN, 4, 1, 0)
    CHAIN_CASE(MODE_PLAIN, 4, 3, 0)
#undef CHAIN_CASE
    return prcnn_fail(PRCNN_EUNSUPPORTED, "prcnn_mlp_chain: no register-chain instance for mode %d widths (%d,%d,%d)/32", mode, n0, n1, n2);
}

static int fill_chain(ChainParams& C, int nlayers, const float* const* wpack, const float* const* bias, const int* nout,
                      const int* relu, float* out, int ld_out, int col_off, int pool_ns) {
    PRCNN_REQUIRE(nlayers >= 1 && nlayers <= 3, "prcnn_mlp_chain: nlayers=%d (1..3)", nlayers);
    PRCNN_REQUIRE(wpack && bias && nout && relu && out, "prcnn_mlp_chain: null pointer");
    for (int l = 0; l < nlayers; l++) {
        PRCNN_REQUIRE(wpack[l] && aligned16(wpack[l]), "prcnn_mlp_chain: layer %d wpack null/unaligned", l);
        PRCNN_REQUIRE(bias[l] == nullptr || aligned16(bias[l]), "prcnn_mlp_chain: layer %d bias must be 16-byte aligned and padded to a multiple of 32", l);
        PRCNN_REQUIRE(nout[l] > 0 && nout[l] <= 512, "prcnn_mlp_chain: layer %d width %d (1..512)", l, nout[l]);
    }
    PRCNN_REQUIRE(pool_ns == 0 || pool_ns == 16 || pool_ns == 32, "prcnn_mlp_chain: pool_ns=%d (0/16/32)", pool_ns);
    PRCNN_REQUIRE(ld_out >= col_off + nout[nlayers - 1], "prcnn_mlp_chain: ld_out too small");
    C.nlayers = nlayers;
    C.a.wpack = wpack[0]; C.a.bias = bias[0]; C.a.Nout = nout[0]; C.a.relu = relu[0];
    if (nlayers > 1) { C.wpack1 = wpack[1]; C.bias1 = bias[1]; C.N1 = nout[1]; C.relu1 = relu[1]; }
    if (nlayers > 2) { C.wpack2 = wpack[2]; C.bias2 = bias[2]; C.N2 = nout[2]; C.relu2 = relu[2]; }
    C.a.out = out; C.a.ld_out = ld_out; C.a.col_off = col_off; C.a.pool_ns = pool_ns;
    return PRCNN_OK;
}

PRCNN_API int prcnn_mlp_chain_rows(const float* in, int ld_in, int64_t rows, int K, int nlayers,
                                   const float* const* wpack, const float* const* bias, const int* nout,
                                   const int* relu, float* out, int ld_out, int col_off, int pool_ns,
                                   const int32_t* seg_cnt, int seg_rows, prcnn_stream_t stream) {
    PRCNN_REQUIRE(in && ld_in >= K && K > 0 && rows >= 0, "prcnn_mlp_chain_rows: bad input");
    PRCNN_REQUIRE(!seg_cnt || (seg_rows > 0 && seg_rows % 128 == 0 && rows % seg_rows == 0 && pool_ns == 0),
                  "prcnn_mlp_chain_rows: seg_rows=%d must be a multiple of 128 dividing rows (and no pooling)", seg_rows);
    ChainParams C = {};
    int rc = fill_chain(C, nlayers, wpack, bias, nout, relu, out, ld_out, col_off, pool_ns);
    if (rc) return rc;
    PRCNN_REQUIRE(pool_ns == 0 || rows % pool_ns == 0, "prcnn_mlp_chain_rows: rows not a multiple of pool_ns");
    C.a.rows = rows; C.a.K = K; C.a.in = in; C.a.ld_in = ld_in;
    C.a.vec_a = aligned16(in) && (ld_in % 4 == 0);
    C.a.seg_cnt = seg_cnt; C.a.seg_rows = seg_rows;
    return dispatch_chain(MODE_PLAIN, C, (hipStream_t)stream);
}

PRCNN_API int prcnn_mlp_chain_group(const float* xyz, const float* new_xyz, const int32_t* idx, const float* feat_cl,
                                    int ld_feat, int B, int N, int M, int nsample, int C_, const float* act_wx,
                                    const float* act_bias, int nlayers,
                                    const float* const* wpack, const float* const* bias, const int* nout,
                                    const int* relu, float* out, int ld_out, int col_off, int pool_ns,
                                    const int32_t* groups_dev, prcnn_stream_t stream) {
    PRCNN_REQUIRE(xyz && idx && (C_ == 0 || feat_cl), "prcnn_mlp_chain_group: null pointer");
    PRCNN_REQUIRE(B >= 0 && N > 0 && M > 0 && nsample > 0 && C_ >= 0 && (C_ == 0 || ld_feat >= C_), "prcnn_mlp_chain_group: bad shape");
    ChainParams C = {};
    int rc = fill_chain(C, nlayers, wpack, bias, nout, relu, out, ld_out, col_off, pool_ns);
    if (rc) return rc;
    PRCNN_REQUIRE(pool_ns == 0 || pool_ns == nsample, "prcnn_mlp_chain_group: pool_ns must equal nsample");
    C.a.rows = (long)B * M * nsample; C.a.K = C_ + 3;
    C.a.rows_dev = groups_dev; C.a.rows_unit = nsample;
    C.a.xyz = xyz; C.a.new_xyz = new_xyz; C.a.idx = idx; C.a.feat = feat_cl; C.a.ld_feat = ld_feat;
    C.a.N = N; C.a.M = M; C.a.ns = nsample; C.a.C = C_;
    C.a.vec_a = C_ > 0 && aligned16(feat_cl) && (ld_feat % 4 == 0);
    rc = set_group_act(C.a, act_wx, act_bias, C_);
    if (rc) return rc;
    return dispatch_chain(MODE_GROUP, C, (hipStream_t)stream);
}

PRCNN_API int prcnn_mlp_chain_interp(const float* known_cl, int ld_known, const int32_t* idx3, const float* w3,
                                     const float* skip_cl, int ld_skip, int B, int n, int m, int C2, int C1,
                                     const float* act_bias, int nlayers, const float* const* wpack, const float* const* bias,
                                     const int* nout, const int* relu, float* out, int ld_out, int col_off,
                                     prcnn_stream_t stream) {
    PRCNN_REQUIRE(known_cl && idx3 && w3 && (C1 == 0 || skip_cl), "prcnn_mlp_chain_interp: null pointer");
    PRCNN_REQUIRE(B >= 0 && n > 0 && m > 0 && C2 > 0 && C1 >= 0 && ld_known >= C2 && (C1 == 0 || ld_skip >= C1), "prcnn_mlp_chain_interp: bad shape");
    ChainParams C = {};
    int rc = fill_chain(C, nlayers, wpack, bias, nout, relu, out, ld_out, col_off, 0);
    if (rc) return rc;
    C.a.rows = (long)B * n; C.a.K = C2 + C1;
    C.a.known = known_cl; C.a.idx3 = idx3; C.a.w3 = w3; C.a.skip = skip_cl; C.a.ld_known = ld_known; C.a.ld_skip = ld_skip;
    C.a.n = n; C.a.m = m; C.a.C2 = C2; C.a.C1 = C1;
    C.a.vec_a = aligned16(known_cl) && (ld_known % 4 == 0);
    C.a.vec_b = C1 > 0 && aligned16(skip_cl) && (ld_skip % 4 == 0) && (C2 % 4 == 0);
    rc = set_interp_act(C.a, act_bias, C2, C1);
    if (rc) return rc;
    if (B % 8 == 0 && n % 128 == 0 && getenv("PRCNN_NO_XCD_ORDER") == nullptr) C.a.xcd_tpf = n / 128;
    return dispatch_chain(MODE_INTERP, C, (hipStream_t)stream);
}

// Hoisted FP0 on the split chain kernel: rows relu(interp(known_cl) + act_bias) (C2 = 128, no skip features) through ONE
// 128 -> 128 layer.  wchain: prcnn_pack_weight_split(chain = 1).  PRCNN_EUNSUPPORTED for other shapes (issue prcnn_mlp_chain_interp).
PRCNN_API int prcnn_mlp_chain_interp_split(const float* known_cl, int ld_known, const int32_t* idx3, const float* w3, int B, int n,
                                           int m, int C2, const float* act_bias, const void* wchain, const float* wpack, const float* bias,
                                           int Nout, int relu, int terms, float* out, int ld_out, int col_off, prcnn_stream_t stream) {
    PRCNN_REQUIRE(known_cl && idx3 && w3 && act_bias && wchain && wpack && aligned16(wpack) && out, "prcnn_mlp_chain_interp_split: null / misaligned pointer");
    PRCNN_REQUIRE(terms == 3 || terms == 6, "prcnn_mlp_chain_interp_split: terms=%d (3 or 6)", terms);
    PRCNN_REQUIRE(B >= 0 && n > 0 && m > 0 && ld_known >= C2 && ld_out >= col_off + Nout, "prcnn_mlp_chain_interp_split: bad shape");
    if (!(C2 == 128 && Nout == 128 && aligned16(known_cl) && ld_known % 4 == 0 && aligned16(act_bias))) return PRCNN_EUNSUPPORTED;
    if (B == 0) return PRCNN_OK;
    ChainParams C = {};
    MlpParams& P = C.a;
    P.rows = (long)B * n; P.K = C2; P.bias = bias; P.Nout = Nout; P.relu = relu;
    P.out = out; P.ld_out = ld_out; P.col_off = col_off; P.rows_unit = 1;
    P.known = known_cl; P.idx3 = idx3; P.w3 = w3; P.ld_known = ld_known; P.n = n; P.m = m; P.C2 = C2; P.C1 = 0;
    P.vec_a = 1; P.act = 2; P.act_bias = act_bias;
    P.wsplit = wchain; P.split_terms = terms; P.wpack = wpack;
    C.nlayers = 1;
    if (B % 8 == 0 && n % 128 == 0 && getenv("PRCNN_NO_XCD_ORDER") == nullptr) P.xcd_tpf = n / 128;
    const dim3 grid(prcnn_divup(P.rows, 128));
    if (terms == 6 && chain_coop_on() && chain_persist_on() && (long)B * m * ld_known < (1L << 30)) {
        const int rc = launch_chain_p<MODE_INTERP, 0>(C, (hipStream_t)stream);
        if (rc) return rc;
    } else if (terms == 6 && chain_coop_on()) hipLaunchKernelGGL((mlp_chain_c_kernel<MODE_INTERP, 0, 6>), grid, dim3(256), 0, (hipStream_t)stream, C);
    else if (terms == 6) hipLaunchKernelGGL((mlp_chain_s_kernel<MODE_INTERP, 0, 6>), grid, dim3(256), 0, (hipStream_t)stream, C);
    else hipLaunchKernelGGL((mlp_chain_s_kernel<MODE_INTERP, 0, 3>), grid, dim3(256), 0, (hipStream_t)stream, C);
    PRCNN_LAUNCH_CHECK("prcnn_mlp_chain_interp_split");
    return PRCNN_OK;
}

// training-mode SharedMLP (forward with batch statistics, dgrad, wgrad): shares the row fetchers and weight image above
#include "mlp_train.h"
