// mlp_train.h -- TRAINING-mode SharedMLP for gfx950 (included at the end of mlp.hip: shares its row fetchers and the packed
// weight image).  BASELINE config 4 (`train_rcnn.py --train_mode rpn`): the reference runs every SharedMLP layer as
// nn.Conv2d(1x1) -> nn.BatchNorm2d (batch statistics) -> ReLU over a materialised (B, C, npoint, nsample) tensor, then
// F.max_pool2d over nsample (lib/net/pointnet2_msg.py:20-45 through the upstream modules; cuDNN / ATen kernels, three saved
// activations per layer).  Here a layer is
//
//   forward   train_fwd_kernel     y = A . W^T on the fp32 MFMA pipe.  A rows are built on the fly: plain rows, rows with the
//                                   PREVIOUS layer's BatchNorm + ReLU applied while they are staged (a = relu(y_prev * scale +
//                                   shift): the normalised activations never exist in HBM), grouped rows [feat[idx] | dxyz] or
//                                   3-NN-interpolated rows [interp(known) | skip].  Only the pre-normalisation output y is
//                                   stored -- ONE saved tensor per layer -- and the epilogue emits per-64-row (mean, M2)
//                                   column partials for the batch statistics (Welford-style: no E[y^2] - mean^2 cancellation).
//             bn_finalize_kernel   partials -> mean / biased variance in double, fixed order (deterministic), scale / shift,
//                                   running statistics (momentum, unbiased variance) exactly as nn.BatchNorm does.
//             train_pool_kernel    last layer: relu(bn(y)) and the max over nsample with the FIRST arg-max (torch.max's rule).
//   backward  bn_bwd_reduce_kernel sum dyhat, sum dyhat * xhat per column (dyhat = G * [yhat > 0]; pooled layers: G lives only
//                                   at the arg-max rows) -> dgamma, dbeta and the constants of
//                                       dy = gamma * invstd * (dyhat - mean(dyhat) - xhat * mean(dyhat * xhat)).
//             train_dgrad_kernel   G_prev = dy . W on the MFMA pipe, dy rebuilt from (G, y) while the A tile is staged.
//             train_wgrad_kernel   dW = dy^T . a: both operands go from global memory STRAIGHT into MFMA operand registers --
//                                   in channels-last rows the 32 lanes of an operand are 32 consecutive channels of one row,
//                                   which is exactly the v_mfma_f32_32x32x2_f32 A / B layout when the reduction runs over rows
//                                   (no LDS, no barrier); row ranges are split over workgroups, partial tiles reduced in fixed order.
// Nothing here folds BatchNorm or removes padded rows: the arithmetic is the reference's, row for row.
#pragma once

struct TrainFwd {
    MlpParams P;                  // rows, K, KB, NB, wpack, Nout, out (= y), ld_out, and the MODE_* source fields
    const float* pro_scale;       // MODE_PLAIN: A = relu(in * pro_scale[k] + pro_shift[k]); NULL = raw rows.  Padded with zeros
    const float* pro_shift;       //             to a multiple of 32 floats
    float* a_dump;                // MODE_GROUP / MODE_INTERP: the assembled A rows are also written here (rows x ld_dump) for wgrad
    int ld_dump;                  // multiple of 4, >= K
    float* part;                  // (nslab, 2, ld_part): column mean and M2 of every 64-row slab
    int ld_part;
    // padding-free rows (training twin of dedup.hip): the rows are the DISTINCT rows of the groups, row r standing for mult[r]
    // identical rows of the padded tensor (ball_query pads a group with copies of its first hit).  Statistics weigh a row by its
    // multiplicity; slab_w (nslab) receives every slab's weight sum.  The live row count is P.rows_dev (device side).
    const float* mult;
    float* slab_w;
};

// FASTP (plain rows, 16-byte aligned, K a multiple of 4): the chunk loads are unconditional 16-byte loads from clamped addresses (rows
// past the end are clamped -- never stored or counted --, k-pieces past K re-read the last piece and are zeroed by multiplication), so
// the four loads of a chunk go out back to back; with the bounds-checked fetch every piece is its own branch and the compiler waits for
// each load before the next (s_waitcnt vmcnt(0) at the joins).
template <int MODE, int WNB, bool FASTP = false>
__global__ __launch_bounds__(MLP_THREADS, ((WNB == 1 && MODE != MODE_INTERP) ? 3 : 2)) void train_fwd_kernel(const TrainFwd T) {
    static_assert(!FASTP || MODE == MODE_PLAIN, "FASTP is the plain-rows form");
    MlpParams P = T.P;
    P.rows = effective_rows(T.P);
    constexpr int QN = 2 * WNB;
    const long tile_id = blockIdx.x;
    const int nb0 = blockIdx.y * QN;
    if (tile_id * MLP_BM >= P.rows) return;              // (device-side row count: tiles past the live rows exit at once)
    __shared__ __attribute__((aligned(16))) float As[2][MLP_BM * MLP_ALD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, j = lane & 31;
    const long row0 = tile_id * MLP_BM;
    const int nchunks = (P.KB + 3) >> 2;
    const int c4 = tid & 7, r0 = tid >> 3;
    RowMeta<MODE> meta[4];
    bool live[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        long grow = row0 + r0 + 32 * u;
        live[u] = grow < P.rows;
        if (!live[u]) grow = P.rows - 1;                  // clamped: results of dead rows are never stored or counted
        make_meta<MODE>(P, grow, meta[u]);
    }
    Raw<MODE> ra[4];
    const bool pro = MODE == MODE_PLAIN && T.pro_scale != nullptr;
    float4 ps = make_float4(0.f, 0.f, 0.f, 0.f), pb = ps;
    auto load_chunk = [&](int c) {
        const int k = c * MLP_BK + c4 * 4;
        if (pro) { ps = ld4(T.pro_scale + k); pb = ld4(T.pro_shift + k); }
        if constexpr (FASTP) {
            const int kc = min(k, P.K - 4);
#pragma unroll
            for (int u = 0; u < 4; u++) ra[u].a = ld4(P.in + meta[u].off + kc);
        } else {
#pragma unroll
            for (int u = 0; u < 4; u++) fetch<MODE>(P, meta[u], k, ra[u]);
        }
    };
    auto store_chunk = [&](int c, int buf) {
        const int k = c * MLP_BK + c4 * 4;
        const float kz = (FASTP && k >= P.K) ? 0.f : 1.f;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float4 v = finish<MODE>(P, meta[u], k, ra[u]);
            if (FASTP) { v.x *= kz; v.y *= kz; v.z *= kz; v.w *= kz; }
            if (pro) {                                    // (scale / shift are zero beyond K: the padding stays zero)
                v.x = fmaxf(v.x * ps.x + pb.x, 0.f); v.y = fmaxf(v.y * ps.y + pb.y, 0.f);
                v.z = fmaxf(v.z * ps.z + pb.z, 0.f); v.w = fmaxf(v.w * ps.w + pb.w, 0.f);
            }
            *reinterpret_cast<float4*>(&As[buf][(r0 + 32 * u) * MLP_ALD + c4 * 4]) = v;
            if (MODE != MODE_PLAIN && T.a_dump && blockIdx.y == 0 && live[u] && k < T.ld_dump)
                *reinterpret_cast<float4*>(T.a_dump + (row0 + r0 + 32 * u) * (long)T.ld_dump + k) = v;
        }
    };
    f32x16 acc[2][WNB];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int n = 0; n < WNB; n++) acc[r][n] = (f32x16){0};
    const bool n_active = (nb0 + wn * WNB) < P.NB;
    constexpr int RING = 4;
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
    store_chunk(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        const int buf = c & 1;
        load_chunk(min(c + 1, nchunks - 1));
        const float* a_base = &As[buf][(wm * 64 + j) * MLP_ALD + 4 * h];
#pragma unroll
        for (int kbl = 0; kbl < 4; kbl++) {
            if (n_active) {
                float4 a[2];
                a[0] = *reinterpret_cast<const float4*>(a_base + kbl * 8);
                a[1] = *reinterpret_cast<const float4*>(a_base + 32 * MLP_ALD + kbl * 8);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].x, bq[kbl][n].x, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].y, bq[kbl][n].y, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].z, bq[kbl][n].z, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].w, bq[kbl][n].w, acc[r][n], 0, 0, 0);
            }
            load_b(c * 4 + kbl + RING, kbl);
        }
        store_chunk(min(c + 1, nchunks - 1), buf ^ 1);
        __syncthreads();
    }
    if (!n_active && !(T.slab_w && nb0 == 0 && wn == 0)) return;
    // ---- epilogue: raw store + column statistics of this wave's 64-row slab ---------------------------------------------
    const long wrow0 = row0 + wm * 64;
    const long left = P.rows - wrow0;
    const int cnt = left >= 64 ? 64 : (left > 0 ? (int)left : 0);          // live rows of the slab (wave-uniform)
    // row weights (multiplicities; 1 without padding-free rows) of this lane's 32 rows, and the slab's weight sum
    float w0[16], w1[16], wsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
        const long g0 = wrow0 + rin, g1 = g0 + 32;
        w0[r] = g0 < P.rows ? (T.mult ? T.mult[g0] : 1.f) : 0.f;
        w1[r] = g1 < P.rows ? (T.mult ? T.mult[g1] : 1.f) : 0.f;
        wsum += w0[r] + w1[r];
    }
    wsum += __shfl_xor(wsum, 32);                                           // (every lane of a half holds the same 32 rows)
    const float inv_w = wsum > 0.f ? 1.0f / wsum : 0.f;
    if (T.slab_w && lane == 0 && nb0 == 0 && wn == 0 && cnt > 0) T.slab_w[wrow0 >> 6] = wsum;
#pragma unroll
    for (int nn = 0; nn < WNB; nn++) {
        const int nb = nb0 + wn * WNB + nn;
        if (nb >= P.NB) continue;
        const int n = nb * 32 + j;
        const bool n_ok = n < P.Nout;
        const f32x16& a0 = acc[0][nn];
        const f32x16& a1 = acc[1][nn];
        float s = 0.f;
        if (cnt == 64 && nb * 32 + 32 <= P.Nout) {       // whole slab x block inside y: unguarded stores, back to back (see layer_epilogue)
            float* o = P.out + (wrow0 + 4 * h) * P.ld_out + P.col_off + n;
            const long ld = P.ld_out;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = (r & 3) + 8 * (r >> 2);
                s += w0[r] * a0[r] + w1[r] * a1[r];
                o[rr * ld] = a0[r];
                o[(32 + rr) * ld] = a1[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
                const long g0 = wrow0 + rin, g1 = g0 + 32;
                s += w0[r] * a0[r] + w1[r] * a1[r];
                if (n_ok && g0 < P.rows) P.out[g0 * P.ld_out + P.col_off + n] = a0[r];
                if (n_ok && g1 < P.rows) P.out[g1 * P.ld_out + P.col_off + n] = a1[r];
            }
        }
        if (T.part) {
            s += __shfl_xor(s, 32);
            const float mean = s * inv_w;
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float d0 = a0[r] - mean, d1 = a1[r] - mean;
                m2 += w0[r] * (d0 * d0) + w1[r] * (d1 * d1);
            }
            m2 += __shfl_xor(m2, 32);
            if (h == 0 && n_ok && cnt > 0) {
                const long slab = wrow0 >> 6;
                T.part[(slab * 2 + 0) * T.ld_part + n] = mean;
                T.part[(slab * 2 + 1) * T.ld_part + n] = m2;
            }
        }
    }
}

// Batch statistics from the slab partials, in two deterministic stages (everything in double, fixed summation order):
//   bn_chunk_kernel     grid (N / 64, chunks): block (64 columns x 4 lanes) sums its range of slabs -> (sum y, sum y^2-equivalent)
//   bn_finalize_kernel  grid (N / 64): combines the <= 64 chunk sums -> cst rows (ld_c floats each): 0 scale = gamma * invstd,
//                       1 shift = beta - mean * scale, 2 mean, 3 invstd; running statistics as nn.BatchNorm updates them.
// A single block walking all 32 k slabs of a 2 M-row layer took 150 us -- per layer, forty layers a step.
#define BN_CHUNKS 64
__global__ __launch_bounds__(256) void bn_chunk_kernel(const float* __restrict__ part, int ld_part, long rows_static, const int32_t* rows_dev,
                                                       const float* __restrict__ slab_w, int N, double* __restrict__ chunk) {
    __shared__ double s1[4][64], s2[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    long rows = rows_static;
    if (rows_dev && (long)*rows_dev < rows) rows = *rows_dev;
    const long nslab = (rows + 63) >> 6;
    const long per = (nslab + BN_CHUNKS - 1) / BN_CHUNKS;
    const long sl0 = (long)blockIdx.y * per;
    const long sl1 = sl0 + per < nslab ? sl0 + per : nslab;
    double a = 0.0, b = 0.0;
    if (n < N) {
        for (long sl = sl0 + q; sl < sl1; sl += 4) {
            const double cn = slab_w ? (double)slab_w[sl] : (double)(sl == nslab - 1 ? rows - sl * 64 : 64);
            const double m = (double)part[(sl * 2 + 0) * ld_part + n], m2 = (double)part[(sl * 2 + 1) * ld_part + n];
            a += cn * m;
            b += m2 + cn * m * m;
        }
    }
    s1[q][c] = a; s2[q][c] = b;
    __syncthreads();
    if (q == 0 && n < N) {
        chunk[((long)blockIdx.y * 2 + 0) * N + n] = (s1[0][c] + s1[1][c]) + (s1[2][c] + s1[3][c]);
        chunk[((long)blockIdx.y * 2 + 1) * N + n] = (s2[0][c] + s2[1][c]) + (s2[2][c] + s2[3][c]);
    }
}
__global__ __launch_bounds__(64) void bn_finalize_kernel(const double* __restrict__ chunk, long rows, int N,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         float momentum, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, float* __restrict__ cst, int ld_c) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    double A = 0.0, Bq = 0.0;
    for (int t = 0; t < BN_CHUNKS; t++) { A += chunk[((long)t * 2 + 0) * N + n]; Bq += chunk[((long)t * 2 + 1) * N + n]; }
    const double mean = A / (double)rows;
    double var = Bq / (double)rows - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[n] : 1.f, be = beta ? beta[n] : 0.f;
    const float scale = g * invstd;
    cst[0 * ld_c + n] = scale;
    cst[1 * ld_c + n] = be - (float)mean * scale;
    cst[2 * ld_c + n] = (float)mean;
    cst[3 * ld_c + n] = invstd;
    if (running_mean) running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * (float)mean;
    if (running_var) {
        const double unbiased = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        running_var[n] = (1.f - momentum) * running_var[n] + momentum * (float)unbiased;
    }
}

// out[g, col_off + n] = max_s relu(y[g * ns + s, n] * scale[n] + shift[n]); arg[g, n] = the FIRST s that attains it (torch.max /
// max_pool2d backward route the gradient there).  ns == 1: plain normalise + ReLU, no arg.  Thread = (group, 4 channels).
__global__ __launch_bounds__(256) void train_pool_kernel(const float* __restrict__ y, int ld_y, long groups, int ns, int N,
                                                         const float* __restrict__ cst, int ld_c, float* __restrict__ out, int ld_out,
                                                         int col_off, uint8_t* __restrict__ arg, const int32_t* __restrict__ seg_off,
                                                         const int32_t* __restrict__ seg_cnt) {
    const int nq = N >> 2;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= groups * nq) return;
    const long g = e / nq;
    const int n = (int)(e - g * nq) * 4;
    const float4 sc = ld4(cst + n), sh = ld4(cst + ld_c + n);
    // padding-free rows: group g owns rows seg_off[g] .. + seg_cnt[g] (its distinct rows; the max over copies of a row is the row)
    const long first = seg_off ? (long)seg_off[g] : g * ns;
    if (seg_cnt) ns = seg_cnt[g];
    const float* p = y + first * (long)ld_y + n;
    float4 best = make_float4(-1.f, -1.f, -1.f, -1.f);           // relu output is >= 0: the first row always wins the first test
    uchar4 ba = make_uchar4(0, 0, 0, 0);
    for (int s = 0; s < ns; s++) {
        const float4 v = ld4(p + (long)s * ld_y);
        const float a0 = fmaxf(v.x * sc.x + sh.x, 0.f), a1 = fmaxf(v.y * sc.y + sh.y, 0.f);
        const float a2 = fmaxf(v.z * sc.z + sh.z, 0.f), a3 = fmaxf(v.w * sc.w + sh.w, 0.f);
        if (a0 > best.x) { best.x = a0; ba.x = (uint8_t)s; }
        if (a1 > best.y) { best.y = a1; ba.y = (uint8_t)s; }
        if (a2 > best.z) { best.z = a2; ba.z = (uint8_t)s; }
        if (a3 > best.w) { best.w = a3; ba.w = (uint8_t)s; }
    }
    *reinterpret_cast<float4*>(out + g * (long)ld_out + col_off + n) = best;
    if (arg) *reinterpret_cast<uchar4*>(arg + g * (long)N + n) = ba;
}

// ---- backward ------------------------------------------------------------------------------------------------------------
// G: gradient w.r.t. the layer's ACTIVATED output.  pool_ns == 0: G is (rows, N) with row stride ldG.  pool_ns > 0 (the layer was
// max-pooled): G is (rows / pool_ns, N) and row r receives G[r / ns, n] iff arg[r / ns, n] == r % ns.
struct TrainBwd {
    long rows;
    int N;                        // channels of this layer's output (multiple of 4)
    const float* G; int ldG;
    const uint8_t* arg; int pool_ns;
    const float* y; int ld_y;     // pre-normalisation output saved by the forward pass
    const float* cst; int ld_c;   // rows 0..3 from bn_finalize_kernel; rows 4, 5 = mean(dyhat), mean(dyhat * xhat) (bn_bwd_finalize)
    // padding-free rows: live row count on the device, row multiplicities (the mean terms of dy count a row mult[r] times), and for
    // the pooled last layer the group of every row + the first row of every group (pool_ns = -1: variable-length groups)
    const int32_t* rows_dev;
    const float* mult;
    const int32_t* row_grp;
    const int32_t* seg_off;
};
__device__ __forceinline__ long bwd_live_rows(const TrainBwd& T) {
    if (!T.rows_dev) return T.rows;
    const long r = *T.rows_dev;
    return r < T.rows ? r : T.rows;
}

__device__ __forceinline__ float4 bwd_G4(const TrainBwd& T, long r, int n) {
    if (T.pool_ns == 0) return ld4(T.G + r * (long)T.ldG + n);
    long g;
    int s;
    if (T.pool_ns < 0) { g = T.row_grp[r]; s = (int)(r - T.seg_off[g]); }
    else { g = r / T.pool_ns; s = (int)(r - g * T.pool_ns); }
    const float4 v = ld4(T.G + g * (long)T.ldG + n);
    const uchar4 a = *reinterpret_cast<const uchar4*>(T.arg + g * (long)T.N + n);
    return make_float4(a.x == s ? v.x : 0.f, a.y == s ? v.y : 0.f, a.z == s ? v.z : 0.f, a.w == s ? v.w : 0.f);
}

// Per 128-row tile: column sums of dyhat and dyhat * xhat.  Block 256 = 16 column quads x 16 row lanes; grid (tiles, N / 64).
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const TrainBwd T, float* __restrict__ part, int ld_part) {
    __shared__ float4 sa[16][16], sb[16][16];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int n = blockIdx.y * 64 + cq * 4;
    const long row0 = (long)blockIdx.x * 128;
    const long live = bwd_live_rows(T);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (n < T.N && row0 < live) {
        const float4 sc = ld4(T.cst + n), sh = ld4(T.cst + T.ld_c + n), mu = ld4(T.cst + 2 * T.ld_c + n), is = ld4(T.cst + 3 * T.ld_c + n);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const long r = row0 + rl + 16 * u;
            if (r >= live) break;
            const float4 g = bwd_G4(T, r, n);
            const float4 yv = ld4(T.y + r * (long)T.ld_y + n);
            const float d0 = (yv.x * sc.x + sh.x > 0.f) ? g.x : 0.f, d1 = (yv.y * sc.y + sh.y > 0.f) ? g.y : 0.f;
            const float d2 = (yv.z * sc.z + sh.z > 0.f) ? g.z : 0.f, d3 = (yv.w * sc.w + sh.w > 0.f) ? g.w : 0.f;
            a.x += d0; a.y += d1; a.z += d2; a.w += d3;
            b.x += d0 * ((yv.x - mu.x) * is.x); b.y += d1 * ((yv.y - mu.y) * is.y);
            b.z += d2 * ((yv.z - mu.z) * is.z); b.w += d3 * ((yv.w - mu.w) * is.w);
        }
    }
    sa[rl][cq] = a; sb[rl][cq] = b;
    __syncthreads();
    if (rl == 0 && n < T.N) {
        float4 A = sa[0][cq], Bq = sb[0][cq];
        for (int t = 1; t < 16; t++) {
            const float4 x = sa[t][cq], z = sb[t][cq];
            A.x += x.x; A.y += x.y; A.z += x.z; A.w += x.w;
            Bq.x += z.x; Bq.y += z.y; Bq.z += z.z; Bq.w += z.w;
        }
        *reinterpret_cast<float4*>(part + ((long)blockIdx.x * 2 + 0) * ld_part + n) = A;
        *reinterpret_cast<float4*>(part + ((long)blockIdx.x * 2 + 1) * ld_part + n) = Bq;
    }
}

// tile partials -> dbeta = sum dyhat, dgamma = sum dyhat * xhat (double, fixed order, the same two stages as the forward
// statistics); cst rows 4, 5 = their means over the rows
__global__ __launch_bounds__(256) void bn_bwd_chunk_kernel(const float* __restrict__ part, int ld_part, long tiles, const int32_t* rows_dev, int N,
                                                           double* __restrict__ chunk) {
    __shared__ double s1[4][64], s2[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    if (rows_dev) { const long lt = ((long)*rows_dev + 127) >> 7; if (lt < tiles) tiles = lt; }
    const long per = (tiles + BN_CHUNKS - 1) / BN_CHUNKS;
    const long t0 = (long)blockIdx.y * per;
    const long t1 = t0 + per < tiles ? t0 + per : tiles;
    double a = 0.0, b = 0.0;
    if (n < N)
        for (long t = t0 + q; t < t1; t += 4) { a += (double)part[(t * 2 + 0) * ld_part + n]; b += (double)part[(t * 2 + 1) * ld_part + n]; }
    s1[q][c] = a; s2[q][c] = b;
    __syncthreads();
    if (q == 0 && n < N) {
        chunk[((long)blockIdx.y * 2 + 0) * N + n] = (s1[0][c] + s1[1][c]) + (s1[2][c] + s1[3][c]);
        chunk[((long)blockIdx.y * 2 + 1) * N + n] = (s2[0][c] + s2[1][c]) + (s2[2][c] + s2[3][c]);
    }
}
// no_bn: a layer WITHOUT normalisation (Conv with bias -> ReLU, the RCNN stage: cfg.RCNN.USE_BN = False): the column sum of dyhat is
// the bias gradient and dy = dyhat (constants 4, 5 stay zero)
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const double* __restrict__ chunk, long rows, int N, float* __restrict__ cst,
                                                             int ld_c, float* __restrict__ dgamma, float* __restrict__ dbeta, int no_bn) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    double A = 0.0, Bq = 0.0;
    for (int t = 0; t < BN_CHUNKS; t++) { A += chunk[((long)t * 2 + 0) * N + n]; Bq += chunk[((long)t * 2 + 1) * N + n]; }
    if (dbeta) dbeta[n] = (float)A;
    if (dgamma) dgamma[n] = (float)Bq;
    cst[4 * ld_c + n] = no_bn ? 0.f : (float)(A / (double)rows);
    cst[5 * ld_c + n] = no_bn ? 0.f : (float)(Bq / (double)rows);
}
// constants of a layer without normalisation: scale 1, shift = bias (or 0), mean 0, invstd 1
__global__ __launch_bounds__(64) void nobn_cst_kernel(const float* __restrict__ bias, int N, float* __restrict__ cst, int ld_c) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    cst[n] = 1.f; cst[ld_c + n] = bias ? bias[n] : 0.f; cst[2 * ld_c + n] = 0.f; cst[3 * ld_c + n] = 1.f;
    cst[4 * ld_c + n] = 0.f; cst[5 * ld_c + n] = 0.f;
}

// dy for 4 consecutive channels: scale * (dyhat - c1 - xhat * c2)
// m: the row's multiplicity (1 without padding-free rows): the row stands for m identical rows of the padded tensor, G is already
// the SUM of their gradients (only the arg-max copy had one), the two mean terms apply to each of the m copies
__device__ __forceinline__ float4 bwd_dy4(const float4 g, const float4 yv, const float4 sc, const float4 sh, const float4 mu,
                                          const float4 is, const float4 c1, const float4 c2, const float m = 1.f) {
    float4 o;
    o.x = sc.x * (((yv.x * sc.x + sh.x > 0.f) ? g.x : 0.f) - m * (c1.x + ((yv.x - mu.x) * is.x) * c2.x));
    o.y = sc.y * (((yv.y * sc.y + sh.y > 0.f) ? g.y : 0.f) - m * (c1.y + ((yv.y - mu.y) * is.y) * c2.y));
    o.z = sc.z * (((yv.z * sc.z + sh.z > 0.f) ? g.z : 0.f) - m * (c1.z + ((yv.z - mu.z) * is.z) * c2.z));
    o.w = sc.w * (((yv.w * sc.w + sh.w > 0.f) ? g.w : 0.f) - m * (c1.w + ((yv.w - mu.w) * is.w) * c2.w));
    return o;
}

// G_prev[rows x Kin] = dy[rows x N] . W[N x Kin]: the layer kernel's tiling with the reduction over this layer's OUTPUT channels;
// wpack = prcnn_pack_weight(W^T (Kin x N)).  The A tile (dy) is rebuilt from (G, y) while it is staged.
struct TrainDgrad {
    TrainBwd B;
    const float* wpack;
    int KB, NB, Kin;              // k-blocks over N, n-blocks over Kin
    float* out; int ld_out;
};
// POOL as in train_wgrad_lds_kernel (0: G per row; 1: pooled over fixed groups; 2: over the groups of padding-free rows).  The rows a
// thread stages are the same for every chunk, so their group / slot are looked up once; the chunk loads themselves are unconditional
// 16-byte loads from clamped addresses (pieces past N zeroed by multiplication).
template <int WNB, int POOL>
__global__ __launch_bounds__(MLP_THREADS, (WNB == 1 ? 3 : 2)) void train_dgrad_kernel(const TrainDgrad D) {
    const TrainBwd& T = D.B;
    constexpr int QN = 2 * WNB;
    const long tile_id = blockIdx.x;
    const int nb0 = blockIdx.y * QN;
    __shared__ __attribute__((aligned(16))) float As[2][MLP_BM * MLP_ALD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, j = lane & 31;
    const long row0 = tile_id * MLP_BM;
    const int nchunks = (D.KB + 3) >> 2;
    const int c4 = tid & 7, r0 = tid >> 3;
    const long live = bwd_live_rows(T);
    if (row0 >= live) return;                          // (device-side row count)
    long grow[4], ggrp[4];
    int gslot[4];
    float rm[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        grow[u] = row0 + r0 + 32 * u;
        if (grow[u] >= live) grow[u] = live - 1;
        rm[u] = T.mult ? T.mult[grow[u]] : 1.f;
        if (POOL == 2) { ggrp[u] = T.row_grp[grow[u]]; gslot[u] = (int)(grow[u] - T.seg_off[ggrp[u]]); }
        else if (POOL == 1) { ggrp[u] = grow[u] / T.pool_ns; gslot[u] = (int)(grow[u] - ggrp[u] * T.pool_ns); }
        else { ggrp[u] = grow[u]; gslot[u] = 0; }
    }
    float4 rg[4], ry[4], sc, sh, mu, is, c1, c2;
    uchar4 rarg[4];
    auto load_chunk = [&](int c) {
        const int k = min(c * MLP_BK + c4 * 4, T.N - 4);          // (N is a multiple of 4)
        sc = ld4(T.cst + k); sh = ld4(T.cst + T.ld_c + k); mu = ld4(T.cst + 2 * T.ld_c + k); is = ld4(T.cst + 3 * T.ld_c + k);
        c1 = ld4(T.cst + 4 * T.ld_c + k); c2 = ld4(T.cst + 5 * T.ld_c + k);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            rg[u] = ld4(T.G + ggrp[u] * (long)T.ldG + k);
            if (POOL != 0) rarg[u] = *reinterpret_cast<const uchar4*>(T.arg + ggrp[u] * (long)T.N + k);
            ry[u] = ld4(T.y + grow[u] * (long)T.ld_y + k);
        }
    };
    auto store_chunk = [&](int c, int buf) {
        const float kz = (c * MLP_BK + c4 * 4 < T.N) ? 1.f : 0.f;   // pieces past N: zero operand (the weight image re-reads its last k-block there)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float4 g = rg[u];
            if (POOL != 0) {
                const int sl = gslot[u];
                g.x = rarg[u].x == sl ? g.x : 0.f; g.y = rarg[u].y == sl ? g.y : 0.f;
                g.z = rarg[u].z == sl ? g.z : 0.f; g.w = rarg[u].w == sl ? g.w : 0.f;
            }
            float4 d = bwd_dy4(g, ry[u], sc, sh, mu, is, c1, c2, rm[u]);
            d.x *= kz; d.y *= kz; d.z *= kz; d.w *= kz;
            *reinterpret_cast<float4*>(&As[buf][(r0 + 32 * u) * MLP_ALD + c4 * 4]) = d;
        }
    };
    f32x16 acc[2][WNB];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int n = 0; n < WNB; n++) acc[r][n] = (f32x16){0};
    const bool n_active = (nb0 + wn * WNB) < D.NB;
    constexpr int RING = 4;
    float4 bq[RING][WNB];
    const float* bptr[WNB];
#pragma unroll
    for (int n = 0; n < WNB; n++) bptr[n] = D.wpack + ((long)min(nb0 + wn * WNB + n, D.NB - 1) * D.KB) * 256 + lane * 4;
    const int kb_last = D.KB - 1;
    auto load_b = [&](int g, int slot) {
        const long off = (long)min(g, kb_last) * 256;
#pragma unroll
        for (int n = 0; n < WNB; n++) bq[slot][n] = ld4(bptr[n] + off);
    };
#pragma unroll
    for (int g = 0; g < RING; g++) load_b(g, g);
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunks; c++) {
        const int buf = c & 1;
        load_chunk(min(c + 1, nchunks - 1));
        const float* a_base = &As[buf][(wm * 64 + j) * MLP_ALD + 4 * h];
#pragma unroll
        for (int kbl = 0; kbl < 4; kbl++) {
            if (n_active) {
                float4 a[2];
                a[0] = *reinterpret_cast<const float4*>(a_base + kbl * 8);
                a[1] = *reinterpret_cast<const float4*>(a_base + 32 * MLP_ALD + kbl * 8);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].x, bq[kbl][n].x, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].y, bq[kbl][n].y, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].z, bq[kbl][n].z, acc[r][n], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int n = 0; n < WNB; n++) acc[r][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r].w, bq[kbl][n].w, acc[r][n], 0, 0, 0);
            }
            load_b(c * 4 + kbl + RING, kbl);
        }
        store_chunk(min(c + 1, nchunks - 1), buf ^ 1);
        __syncthreads();
    }
    if (!n_active) return;
    const long wrow0 = row0 + wm * 64;
#pragma unroll
    for (int nn = 0; nn < WNB; nn++) {
        const int nb = nb0 + wn * WNB + nn;
        if (nb >= D.NB) continue;
        const int n = nb * 32 + j;
        if (wrow0 + 64 <= live && nb * 32 + 32 <= D.Kin) {     // whole block inside the output: unguarded stores (see layer_epilogue)
            float* o = D.out + (wrow0 + 4 * h) * D.ld_out + n;
            const long ld = D.ld_out;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = (r & 3) + 8 * (r >> 2);
                o[rr * ld] = acc[0][nn][r];
                o[(32 + rr) * ld] = acc[1][nn][r];
            }
            continue;
        }
        if (n >= D.Kin) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
            const long g0 = wrow0 + rin, g1 = g0 + 32;
            if (g0 < live) D.out[g0 * D.ld_out + n] = acc[0][nn][r];
            if (g1 < live) D.out[g1 * D.ld_out + n] = acc[1][nn][r];
        }
    }
}

// dW[n, k] = sum_r dy[r, n] * a[r, k].  Both operands go from global memory straight into MFMA operand registers: per step a wave
// consumes TWO rows; lane (h, j) loads KQ consecutive input channels (KQ * j ...) of row r + h and NQ consecutive output channels of
// dy -- the 32 lanes of a half read one contiguous piece of the row -- and element x of its vector feeds the blocks (x, .), whose C
// row i therefore stands for channel KQ * i + x.  A wave owns a (32 KQ) x (32 NQ) tile; the four waves of a workgroup are laid out
// WK x WN over the channels and WR = 4 / (WK WN) over the ROWS: narrow layers (SA1: 3..64 channels, millions of rows) put all four
// waves on different rows of ONE tile instead of computing padding -- they are HBM-bound then (one pass over a, G and y).
// Rows are dealt to grid.x in contiguous ranges; every (range, row group) writes its partial tile, train_wgrad_reduce_kernel sums
// the partials in a fixed order (deterministic).
struct TrainWgrad {
    TrainBwd B;
    const float* a; int lda; int K;          // a rows (rows x lda), K valid channels; lda a multiple of 2
    const float* pro_scale;                  // optional relu(a * scale + shift) (the forward's prologue), zero-padded to 128 floats
    const float* pro_shift;
    long rows_per_split;                     // multiple of 2 * WR
    float* part;                             // (splits * WR, N, K) partial dW
};
#define WG_UNROLL 4
template <int KQ, int NQ, int WK, int WN>
__global__ __launch_bounds__(256) void train_wgrad_kernel(const TrainWgrad W) {
    constexpr int WR = 4 / (WK * WN);
    const TrainBwd& T = W.B;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / (WK * WN), wk = (wave / WN) % WK, wn = wave % WN;
    const int h = lane >> 5, j = lane & 31;
    const int kbase = (blockIdx.y * WK + wk) * 32 * KQ, nbase = (blockIdx.z * WN + wn) * 32 * NQ;
    const int k = kbase + KQ * j, n = nbase + NQ * j;            // this lane's first input / output channel
    const bool k_ok = k + KQ <= W.lda, n_ok = n + NQ <= T.N;     // (vector wholly inside the row; lda, N multiples of KQ, NQ)
    const bool pro = W.pro_scale != nullptr;
    float ps[KQ], pb[KQ], km[KQ];
#pragma unroll
    for (int x = 0; x < KQ; x++) {
        km[x] = k + x < W.K ? 1.f : 0.f;
        ps[x] = (pro && k_ok) ? W.pro_scale[k + x] : 1.f;
        pb[x] = (pro && k_ok) ? W.pro_shift[k + x] : 0.f;
    }
    float sc[NQ], sh[NQ], mu[NQ], is[NQ], c1[NQ], c2[NQ];
#pragma unroll
    for (int z = 0; z < NQ; z++) {
        sc[z] = n_ok ? T.cst[n + z] : 0.f; sh[z] = n_ok ? T.cst[T.ld_c + n + z] : 0.f;
        mu[z] = n_ok ? T.cst[2 * T.ld_c + n + z] : 0.f; is[z] = n_ok ? T.cst[3 * T.ld_c + n + z] : 0.f;
        c1[z] = n_ok ? T.cst[4 * T.ld_c + n + z] : 0.f; c2[z] = n_ok ? T.cst[5 * T.ld_c + n + z] : 0.f;
    }
    const long per_group = W.rows_per_split / WR;
    const long r_begin = (long)blockIdx.x * W.rows_per_split + wr * per_group;
    long r_end = r_begin + per_group;
    const long live = bwd_live_rows(T);
    if (r_end > live) r_end = live;
    f32x16 acc[KQ][NQ];
#pragma unroll
    for (int x = 0; x < KQ; x++)
#pragma unroll
        for (int z = 0; z < NQ; z++) acc[x][z] = (f32x16){0};
    const int kk = k_ok ? k : 0, nn = n_ok ? n : 0;
    typedef float vk_t __attribute__((ext_vector_type(KQ)));
    typedef float vn_t __attribute__((ext_vector_type(NQ)));
    for (long r = r_begin; r < r_end; r += 2 * WG_UNROLL) {
        vk_t av[WG_UNROLL];
        vn_t gv[WG_UNROLL], yv[WG_UNROLL];
        float valid[WG_UNROLL], rmu[WG_UNROLL];
#pragma unroll
        for (int u = 0; u < WG_UNROLL; u++) {
            long rr = r + 2 * u + h;
            valid[u] = rr < r_end ? 1.f : 0.f;
            if (rr >= r_end) rr = r_end - 1;
            rmu[u] = T.mult ? T.mult[rr] : 1.f;
            av[u] = *reinterpret_cast<const vk_t*>(W.a + rr * (long)W.lda + kk);
            yv[u] = *reinterpret_cast<const vn_t*>(T.y + rr * (long)T.ld_y + nn);
            if (T.pool_ns == 0) {
                gv[u] = *reinterpret_cast<const vn_t*>(T.G + rr * (long)T.ldG + nn);
            } else {
                long g;
                int s;
                if (T.pool_ns < 0) { g = T.row_grp[rr]; s = (int)(rr - T.seg_off[g]); }
                else { g = rr / T.pool_ns; s = (int)(rr - g * T.pool_ns); }
                const vn_t v = *reinterpret_cast<const vn_t*>(T.G + g * (long)T.ldG + nn);
                const uint8_t* ap = T.arg + g * (long)T.N + nn;
#pragma unroll
                for (int z = 0; z < NQ; z++) gv[u][z] = ap[z] == s ? v[z] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < WG_UNROLL; u++) {
            float a[KQ], d[NQ];
#pragma unroll
            for (int x = 0; x < KQ; x++) {
                float t = av[u][x];
                if (pro) t = fmaxf(t * ps[x] + pb[x], 0.f);
                a[x] = t * (km[x] * valid[u]);                   // channels past K and rows past the range contribute zero
            }
#pragma unroll
            for (int z = 0; z < NQ; z++) {
                const float yy = yv[u][z];
                d[z] = sc[z] * (((yy * sc[z] + sh[z] > 0.f) ? gv[u][z] : 0.f) - rmu[u] * (c1[z] + ((yy - mu[z]) * is[z]) * c2[z]));
            }
#pragma unroll
            for (int x = 0; x < KQ; x++)
#pragma unroll
                for (int z = 0; z < NQ; z++) acc[x][z] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], d[z], acc[x][z], 0, 0, 0);
        }
    }
    // C[i][jj] of block (x, z): input channel kbase + KQ i + x, output channel nbase + NQ jj + z
    float* P = W.part + ((long)blockIdx.x * WR + wr) * T.N * W.K;
    if (kbase + 32 * KQ <= W.K && nbase + 32 * NQ <= T.N) {
        // the wave's block lies inside dW: unguarded stores, back to back (with a bounds branch per element the compiler opens every
        // store with s_waitcnt vmcnt(0), and that counter includes the stores themselves on gfx9)
#pragma unroll
        for (int x = 0; x < KQ; x++)
#pragma unroll
            for (int z = 0; z < NQ; z++) {
                float* q = P + (long)(nbase + NQ * j + z) * W.K + kbase + 4 * h * KQ + x;
#pragma unroll
                for (int e = 0; e < 16; e++) q[((e & 3) + 8 * (e >> 2)) * KQ] = acc[x][z][e];
            }
        return;
    }
#pragma unroll
    for (int x = 0; x < KQ; x++)
#pragma unroll
        for (int z = 0; z < NQ; z++) {
            const int no = nbase + NQ * j + z;
            if (no >= T.N) continue;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int i = (e & 3) + 8 * (e >> 2) + 4 * h;
                const int ko = kbase + KQ * i + x;
                if (ko < W.K) P[(long)no * W.K + ko] = acc[x][z][e];
            }
        }
}

// Wide layers (K > 64 and N > 64): the 128 x 128 tile through LDS.  The direct kernel above has every wave fetch its own operands,
// so the two waves that share a k-half (or an n-half) read the same rows twice and rebuild dy twice -- at 6 bytes per cycle and
// wave it ran into the L2 bandwidth (203 us for 262 144 x 128 x 128, 42 TFLOP/s).  Here a workgroup stages 32 rows at a time:
// thread (row lane t / 32, channel quad t % 32) loads a, G, y as 16-byte pieces, applies the forward's prologue to a and rebuilds
// dy ONCE, and writes both tiles to LDS; a wave then reads its operands as one ds_read_b64 each per two rows (lane j: channels
// 2j, 2j+1 of its half -- 256 contiguous bytes per half-wave, conflict-free).  Register-staged double buffer, one barrier per chunk.
#define WL_ROWS 32
#define WL_LD 128
// POOL: how the layer's upstream gradient reaches a row (0: G is (rows, N); 1: max-pooled over fixed groups of pool_ns rows; 2: over
// the variable-length groups of padding-free rows), MULT: padding-free rows (row multiplicities), PRO: the A operand is the previous
// layer's saved y (its BatchNorm + ReLU applied while staging).  Compile-time, and every load of a chunk is issued back to back from
// clamped addresses (rows past the range and channels past K / N are zeroed by multiplication afterwards, never by a branch around
// the load): with run-time branches the compiler put an s_waitcnt vmcnt(0) at every join, i.e. a dozen exposed global round trips
// per 32-row chunk against 1.7 us of MFMA work in it.  The one dependent load of the flat form (row -> group) is requested a chunk
// ahead of the data that needs it.
template <int POOL, bool MULT, bool PRO>
__global__ __launch_bounds__(256, 2) void train_wgrad_lds_kernel(const TrainWgrad W) {
    const TrainBwd& T = W.B;
    __shared__ __attribute__((aligned(16))) float As[2][WL_ROWS * WL_LD], Ds[2][WL_ROWS * WL_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, j = lane & 31;
    const int kt0 = blockIdx.y * 128, nt0 = blockIdx.z * 128;
    // staging role: rows rl + 8 u (u = 0..3), channels 4 cq .. 4 cq + 3 of the tile
    const int cq = tid & 31, rl = tid >> 5;
    const int ka = kt0 + 4 * cq, na = nt0 + 4 * cq;
    const int kc = (ka + 4 <= W.lda) ? ka : 0, nc = (na + 4 <= T.N) ? na : 0;      // (lda and N are multiples of 4 here)
    const float dead_n = (na + 4 <= T.N) ? 1.f : 0.f;
    float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f), km;
    const bool ka_ok = ka + 4 <= W.lda;
    km.x = (ka_ok && ka < W.K) ? 1.f : 0.f; km.y = (ka_ok && ka + 1 < W.K) ? 1.f : 0.f;
    km.z = (ka_ok && ka + 2 < W.K) ? 1.f : 0.f; km.w = (ka_ok && ka + 3 < W.K) ? 1.f : 0.f;
    if (PRO) { ps = ld4(W.pro_scale + kc); pb = ld4(W.pro_shift + kc); }
    const float4 sc = ld4(T.cst + nc), sh = ld4(T.cst + T.ld_c + nc), mu = ld4(T.cst + 2 * T.ld_c + nc), is = ld4(T.cst + 3 * T.ld_c + nc);
    const float4 c1 = ld4(T.cst + 4 * T.ld_c + nc), c2 = ld4(T.cst + 5 * T.ld_c + nc);
    const long r_begin = (long)blockIdx.x * W.rows_per_split;
    long r_end = r_begin + W.rows_per_split;
    const long live = bwd_live_rows(T);
    if (r_end > live) r_end = live;
    const int nchunks = r_end > r_begin ? (int)((r_end - r_begin + WL_ROWS - 1) / WL_ROWS) : 0;
    float4 ra[4], rg[4], ry[4];
    uchar4 rarg[4];
    int rso[4], gcur[4], gnext[4];
    float mcur[4], mnext[4];
    auto row_of = [&](int c, int u, bool& ok) {
        long rr = r_begin + (long)c * WL_ROWS + rl + 8 * u;
        ok = rr < r_end;
        return ok ? rr : r_end - 1;
    };
    auto load_meta = [&](int c, int (&g)[4], float (&m)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            bool ok;
            const long rr = row_of(c, u, ok);
            g[u] = POOL == 2 ? T.row_grp[rr] : 0;
            m[u] = MULT ? T.mult[rr] : 1.f;
        }
    };
    auto load_data = [&](int c, const int (&g)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            bool ok;
            const long rr = row_of(c, u, ok);
            ra[u] = ld4(W.a + rr * (long)W.lda + kc);
            ry[u] = ld4(T.y + rr * (long)T.ld_y + nc);
            if (POOL == 0) {
                rg[u] = ld4(T.G + rr * (long)T.ldG + nc);
            } else {
                const long grp = POOL == 2 ? (long)g[u] : rr / T.pool_ns;
                rg[u] = ld4(T.G + grp * (long)T.ldG + nc);
                rarg[u] = *reinterpret_cast<const uchar4*>(T.arg + grp * (long)T.N + nc);
                rso[u] = POOL == 2 ? (int)rr - T.seg_off[grp] : (int)(rr - grp * T.pool_ns);
            }
        }
    };
    auto pin4 = [](float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); };
    auto store_chunk = [&](int c, int buf, const float (&m)[4]) {
        // (the optimiser otherwise hoists this arithmetic -- and with it the wait for the chunk's loads -- above the MFMA loop)
#pragma unroll
        for (int u = 0; u < 4; u++) { pin4(ra[u]); pin4(rg[u]); pin4(ry[u]); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            bool ok;
            (void)row_of(c, u, ok);
            float4 a = ra[u];
            if (PRO) { a.x = fmaxf(a.x * ps.x + pb.x, 0.f); a.y = fmaxf(a.y * ps.y + pb.y, 0.f); a.z = fmaxf(a.z * ps.z + pb.z, 0.f); a.w = fmaxf(a.w * ps.w + pb.w, 0.f); }
            const float v = ok ? 1.f : 0.f;                       // rows past the range and channels past K contribute zero
            a.x *= km.x * v; a.y *= km.y * v; a.z *= km.z * v; a.w *= km.w * v;
            float4 g = rg[u];
            if (POOL != 0) {
                const int sl = rso[u];
                g.x = rarg[u].x == sl ? g.x : 0.f; g.y = rarg[u].y == sl ? g.y : 0.f;
                g.z = rarg[u].z == sl ? g.z : 0.f; g.w = rarg[u].w == sl ? g.w : 0.f;
            }
            float4 d = bwd_dy4(g, ry[u], sc, sh, mu, is, c1, c2, m[u]);
            d.x *= dead_n; d.y *= dead_n; d.z *= dead_n; d.w *= dead_n;
            *reinterpret_cast<float4*>(&As[buf][(rl + 8 * u) * WL_LD + 4 * cq]) = a;
            *reinterpret_cast<float4*>(&Ds[buf][(rl + 8 * u) * WL_LD + 4 * cq]) = d;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int z = 0; z < 2; z++) acc[x][z] = (f32x16){0};
    if (nchunks > 0) {
        load_meta(0, gcur, mcur);
        load_data(0, gcur);
        load_meta(min(1, nchunks - 1), gnext, mnext);
        store_chunk(0, 0, mcur);
        __syncthreads();
        for (int c = 0; c < nchunks; c++) {
            const int buf = c & 1;
            const int c1n = min(c + 1, nchunks - 1);
#pragma unroll
            for (int u = 0; u < 4; u++) { gcur[u] = gnext[u]; mcur[u] = mnext[u]; }
            load_data(c1n, gcur);
            load_meta(min(c + 2, nchunks - 1), gnext, mnext);
            __builtin_amdgcn_sched_barrier(0);     // the requests go out BEFORE the chunk's 64 MFMAs (the scheduler sinks them to mid-loop otherwise)
            const float* ap = &As[buf][h * WL_LD + wk * 64 + 2 * j];
            const float* dp = &Ds[buf][h * WL_LD + wn * 64 + 2 * j];
#pragma unroll
            for (int t = 0; t < WL_ROWS / 2; t++) {
                const float2 a = *reinterpret_cast<const float2*>(ap + 2 * t * WL_LD);
                const float2 d = *reinterpret_cast<const float2*>(dp + 2 * t * WL_LD);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, d.x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, d.y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, d.x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, d.y, acc[1][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            store_chunk(c1n, buf ^ 1, mcur);      // (the last iteration re-stages the last chunk into the idle buffer: no branch around the loads' consumer)
            __syncthreads();
        }
    }
    float* P = W.part + (long)blockIdx.x * T.N * W.K;
    const int kbase = kt0 + wk * 64, nbase = nt0 + wn * 64;
    if (kbase + 64 <= W.K && nbase + 64 <= T.N) {        // block inside dW: unguarded stores (see train_wgrad_kernel)
#pragma unroll
        for (int x = 0; x < 2; x++)
#pragma unroll
            for (int z = 0; z < 2; z++) {
                float* q = P + (long)(nbase + 2 * j + z) * W.K + kbase + 8 * h + x;
#pragma unroll
                for (int e = 0; e < 16; e++) q[((e & 3) + 8 * (e >> 2)) * 2] = acc[x][z][e];
            }
        return;
    }
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int z = 0; z < 2; z++) {
            const int no = nbase + 2 * j + z;
            if (no >= T.N) continue;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int i = (e & 3) + 8 * (e >> 2) + 4 * h;
                const int ko = kbase + 2 * i + x;
                if (ko < W.K) P[(long)no * W.K + ko] = acc[x][z][e];
            }
        }
}

template <int POOL, bool MULT>
static void launch_wgrad_lds(const TrainWgrad& Wg, dim3 grid, hipStream_t s) {
    if (Wg.pro_scale) hipLaunchKernelGGL((train_wgrad_lds_kernel<POOL, MULT, true>), grid, dim3(256), 0, s, Wg);
    else hipLaunchKernelGGL((train_wgrad_lds_kernel<POOL, MULT, false>), grid, dim3(256), 0, s, Wg);
}

// out[n, (k + k_unrot) % K] = sum over the partials, in a fixed order (deterministic).  Block = 16 elements x 16 lanes over the partial
// index, eight loads in flight per lane (round 5: narrow layers have up to 4096 partials -- 1024 row ranges x 4 row groups -- and the
// round-3 form walked them 4 lanes x 4 loads at a time: 256 dependent L2 round trips, 24 us per call, 0.8 ms per training step);
// k_unrot = 3 turns the kernel's [feat | dxyz] column order of a grouped first layer back into torch's [dxyz | feat].
#define WRED_E 16
#define WRED_Q 16
__global__ __launch_bounds__(256) void train_wgrad_reduce_kernel(const float* __restrict__ part, int nparts, long count, int K, int k_unrot,
                                                                 float* __restrict__ out) {
    __shared__ float sm[WRED_Q][WRED_E];
    const int c = threadIdx.x & (WRED_E - 1), q = threadIdx.x / WRED_E;
    const long e = (long)blockIdx.x * WRED_E + c;
    float s = 0.f;
    if (e < count) {
        const int per = (nparts + WRED_Q - 1) / WRED_Q;
        const int t0 = q * per, t1 = min(nparts, t0 + per);
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int t = t0;
        for (; t + 8 <= t1; t += 8)
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] += part[(long)(t + u) * count + e];
        for (; t < t1; t++) a[0] += part[(long)t * count + e];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    sm[q][c] = s;
    __syncthreads();
    if (q == 0 && e < count) {
        float tot = 0.f;
#pragma unroll
        for (int u = 0; u < WRED_Q; u++) tot += sm[u][c];
        const long n = e / K;
        const int kk = (int)(e - n * K);
        out[n * K + (kk + k_unrot) % K] = tot;
    }
}

// Backward of the grouped first layer's gather, channels-last: dfeat[b, idx[b, m, s], 0:C] += G[(b, m, s), 0:C].  ball_query pads a
// group with copies of its first hit: the rows that map to the group's first index are summed in registers (one atomic per
// channel for all of them), every other row gets its own.  Thread = (group, 4 channels); correct for any index tensor.
__global__ __launch_bounds__(256) void group_rows_grad_kernel(const float* __restrict__ G, int ldG, const int32_t* __restrict__ idx,
                                                              long groups, int per_frame_groups, int ns, int C, int N,
                                                              float* __restrict__ dfeat, int ld_d) {
    const int cq = (C + 3) >> 2;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= groups * cq) return;
    const long g = e / cq;
    const int c = (int)(e - g * cq) * 4;
    const long b = g / per_frame_groups;
    const int32_t* ip = idx + g * ns;
    const int first = ip[0];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float* base = dfeat + b * (long)N * ld_d + c;
    const bool full = c + 4 <= C;
    for (int s = 0; s < ns; s++) {
        const float* gp = G + (g * ns + s) * (long)ldG + c;
        float4 v;
        if (full) v = ld4(gp);
        else { v.x = gp[0]; v.y = c + 1 < C ? gp[1] : 0.f; v.z = c + 2 < C ? gp[2] : 0.f; v.w = 0.f; }
        const int id = ip[s];
        if (id == first) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        else {
            float* d = base + (long)id * ld_d;
            atomicAdd(d, v.x);
            if (c + 1 < C) atomicAdd(d + 1, v.y);
            if (c + 2 < C) atomicAdd(d + 2, v.z);
            if (c + 3 < C) atomicAdd(d + 3, v.w);
        }
    }
    float* d = base + (long)first * ld_d;
    atomicAdd(d, acc.x);
    if (c + 1 < C) atomicAdd(d + 1, acc.y);
    if (c + 2 < C) atomicAdd(d + 2, acc.z);
    if (c + 3 < C) atomicAdd(d + 3, acc.w);
}

// Backward of three_interpolate on channels-last rows: dknown[b, idx3[r, j], 0:C] += w3[r, j] * G[r, 0:C].  One wave per row piece:
// lanes = consecutive channels (one or two cache lines per atomic instruction).
__global__ __launch_bounds__(256) void interp_rows_grad_kernel(const float* __restrict__ G, int ldG, const int32_t* __restrict__ idx3,
                                                               const float* __restrict__ w3, long rows, int n, int m, int C,
                                                               float* __restrict__ dknown, int ld_d) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * C) return;
    const long r = e / C;
    const int c = (int)(e - r * C);
    const long b = r / n;
    const float g = G[r * (long)ldG + c];
    float* base = dknown + b * (long)m * ld_d + c;
#pragma unroll
    for (int t = 0; t < 3; t++) atomicAdd(base + (long)idx3[r * 3 + t] * ld_d, g * w3[r * 3 + t]);
}


// The same backward as a GATHER, no atomics: the (row, weight) pairs are first bucketed by the known point they refer to (a
// counting sort per frame, counters in LDS), then one wave per known point adds up its references -- in ascending row order, so the
// result does not depend on scheduling -- and writes the row once (dknown need not be zeroed).  181 M float atomics per RPN training
// step (1.05 ms) become 400 MB of coalesced row reads.
#define ICSR_THREADS 1024
#define ICSR_MAX_M 8192
#define ICSR_SORT_CAP 512                    // references to one known point sorted in LDS (more: taken in bucket order, not repeatable)
// Frames of the reference list: frame b owns rows [b n, (b + 1) n) (row_off == NULL: the interpolation's (B, n, refs) triples), or --
// padding-free rows -- the rows of its groups, [row_off[b gpf], row_off[(b + 1) gpf]) with the list's live length *rows_dev as
// the last bound.  A row holds `refs` references; a reference is (idx[e] - b idx_sub) in [0, m), weight w[e] (NULL: 1).
struct CsrFrames { const int32_t* row_off; int gpf; const int32_t* rows_dev; int n; int B; };
__global__ __launch_bounds__(ICSR_THREADS) void interp_csr_build_kernel(const int32_t* __restrict__ idx, const float* __restrict__ w, int refs,
                                                                        int m, int idx_sub, int32_t* __restrict__ off_all,
                                                                        int2* __restrict__ ent, CsrFrames F) {
    __shared__ int cnt[ICSR_MAX_M];
    __shared__ int wsum[ICSR_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long r0, r1;
    if (F.row_off) {
        r0 = F.row_off[(long)b * F.gpf];
        r1 = b + 1 < F.B ? (long)F.row_off[(long)(b + 1) * F.gpf] : (long)*F.rows_dev;
    } else {
        r0 = (long)b * F.n; r1 = r0 + F.n;
    }
    const int e0 = (int)(r0 * refs), E = (int)((r1 - r0) * refs);
    const int32_t* __restrict__ ib = idx + e0;
    const int base = b * idx_sub;
    int32_t* __restrict__ off = off_all + (size_t)b * (m + 1);
    for (int i = tid; i < m; i += ICSR_THREADS) cnt[i] = 0;
    __syncthreads();
    for (int q0 = tid; q0 < E; q0 += 8 * ICSR_THREADS) {
        int jv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) jv[u] = ib[min(q0 + u * ICSR_THREADS, E - 1)] - base;
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (q0 + u * ICSR_THREADS < E) atomicAdd(&cnt[min(max(jv[u], 0), m - 1)], 1);
    }
    __syncthreads();
    // exclusive scan over the m counters: ceil(m / 1024) consecutive counters per thread
    const int per = (m + ICSR_THREADS - 1) / ICSR_THREADS;
    const int j0 = tid * per, j1 = min(m, j0 + per);
    int tsum = 0;
    for (int jj = j0; jj < j1; jj++) tsum += cnt[jj];
    int incl = tsum;
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = e0 + incl - tsum;                           // absolute positions in the entry array
    for (int wv = 0; wv < wave; wv++) run += wsum[wv];
    for (int jj = j0; jj < j1; jj++) { const int c = cnt[jj]; off[jj] = run; cnt[jj] = run; run += c; }
    if (tid == ICSR_THREADS - 1) off[m] = e0 + E;
    __syncthreads();
    for (int q0 = tid; q0 < E; q0 += 8 * ICSR_THREADS) {
        int jv[8]; float wv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int q = min(q0 + u * ICSR_THREADS, E - 1); jv[u] = ib[q] - base; wv[u] = w ? w[e0 + q] : 1.f; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = q0 + u * ICSR_THREADS;
            if (q < E) {
                const int pos = atomicAdd(&cnt[min(max(jv[u], 0), m - 1)], 1);
                ent[pos] = make_int2((e0 + q) / refs, __float_as_int(wv[u]));      // (row of the whole list, weight); rows ascend with q
            }
        }
    }
}

// one wave per known point; lanes = channel quads (C <= 1024)
__global__ __launch_bounds__(256) void interp_csr_gather_kernel(const float* __restrict__ G, int ldG, const int32_t* __restrict__ off_all,
                                                                const int2* __restrict__ ent, long total, int m, int C,
                                                                float* __restrict__ dknown, int ld_d) {
    __shared__ int2 sl[4][2][ICSR_SORT_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gj = (long)blockIdx.x * 4 + wave;               // b * m + j
    if (gj >= total) return;
    const int b = (int)(gj / m), j = (int)(gj - (long)b * m);
    const int32_t* off = off_all + (size_t)b * (m + 1);
    const int s0 = off[j], len = off[j + 1] - s0;
    int2* U = sl[wave][0];
    int2* L = sl[wave][1];
    const bool sorted = len <= ICSR_SORT_CAP;
    if (sorted) {
        // rank by row (a row refers to a known point at most three times; those tie-break on their position in the bucket), then
        // place by rank.  One wave per list: its LDS operations execute in order, no barrier.
        for (int k = lane; k < len; k += 64) U[k] = ent[s0 + k];
        for (int k = lane; k < len; k += 64) {
            const int2 mine = U[k];
            int rank = 0;
            for (int t = 0; t < len; t++) { const int rt = U[t].x; rank += (rt < mine.x || (rt == mine.x && t < k)) ? 1 : 0; }
            L[rank] = mine;
        }
    }
    const float* Gb = G;                                       // entries carry rows of the whole list
    float* d = dknown + ((size_t)b * m + j) * ld_d;
    for (int c = lane * 4; c < C; c += 256) {                  // (ld and C multiples of 4: host-checked)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = 0; k0 < len; k0 += 4) {                  // four rows requested before the first is added (same order of additions)
            int2 e[4]; float4 g[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = min(k0 + u, len - 1);
                e[u] = sorted ? L[k] : ent[s0 + k];
                g[u] = ld4(Gb + (size_t)e[u].x * ldG + c);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float w = k0 + u < len ? __int_as_float(e[u].y) : 0.f;
                if (k0 + u < len) {
                    acc.x = __fadd_rn(acc.x, __fmul_rn(g[u].x, w)); acc.y = __fadd_rn(acc.y, __fmul_rn(g[u].y, w));
                    acc.z = __fadd_rn(acc.z, __fmul_rn(g[u].z, w)); acc.w = __fadd_rn(acc.w, __fmul_rn(g[u].w, w));
                }
            }
        }
        *reinterpret_cast<float4*>(d + c) = acc;
    }
}

// ---- padding-free rows for training (the training twin of dedup.hip) -------------------------------------------------------
// ball_query pads a group that has fewer than nsample neighbours by repeating its first hit; the reference pushes every copy
// through all layers.  Copies are identical rows, so a group is represented by its DISTINCT rows, the first one carrying the
// multiplicity 1 + (number of copies): batch statistics weigh rows by multiplicity, the max over copies is the row, and in
// backward the copies' gradients add up on the one row (see bwd_dy4).  Exact algebra, deterministic layout: counts -> exclusive
// scan -> fill (no atomics, the row order is the group order).  Correct for any index tensor: an entry is a copy iff it equals
// the group's first entry and is not the first.
__global__ void train_group_count_kernel(const int32_t* __restrict__ idx, long groups, int ns, int32_t* __restrict__ cnt) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const int32_t* ip = idx + g * ns;
    const int first = ip[0];
    int c = 1;
    for (int s = 1; s < ns; s++) c += ip[s] != first ? 1 : 0;
    cnt[g] = c;
}
// exclusive scan of cnt[0 .. groups) by ONE workgroup (groups <= a few hundred thousand); total -> *rows_dev
__global__ __launch_bounds__(1024) void train_group_scan_kernel(const int32_t* __restrict__ cnt, long groups, int32_t* __restrict__ off,
                                                                int32_t* __restrict__ total) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const long per = (groups + 1023) / 1024;
    const long g0 = (long)tid * per;
    const long g1 = g0 + per < groups ? g0 + per : groups;
    int s = 0;
    for (long g = g0; g < g1; g++) s += cnt[g];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                     // Hillis-Steele inclusive scan of the 1024 partial sums
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid ? part[tid - 1] : 0;
    for (long g = g0; g < g1; g++) { off[g] = run; run += cnt[g]; }
    if (tid == 1023) *total = part[1023];
}
__global__ void train_group_fill_kernel(const int32_t* __restrict__ idx, const float* __restrict__ new_xyz, long groups, int per_frame_groups,
                                        int ns, int N, const int32_t* __restrict__ off, int32_t* __restrict__ ridx, float* __restrict__ rnx,
                                        float* __restrict__ mult, int32_t* __restrict__ row_grp) {
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const int32_t* ip = idx + g * ns;
    const int first = ip[0];
    const long base = (g / per_frame_groups) * (long)N;
    const float cx = new_xyz[g * 3], cy = new_xyz[g * 3 + 1], cz = new_xyz[g * 3 + 2];
    int r = off[g], copies = 0;
    const int r0 = r;
    for (int s = 0; s < ns; s++) {
        const int id = ip[s];
        if (s > 0 && id == first) { copies++; continue; }
        ridx[r] = (int32_t)(base + id);
        rnx[(long)r * 3] = cx; rnx[(long)r * 3 + 1] = cy; rnx[(long)r * 3 + 2] = cz;
        mult[r] = 1.f;
        row_grp[r] = (int32_t)g;
        r++;
    }
    mult[r0] = (float)(1 + copies);
}
// dfeat[ridx[r], 0:C] += G[r, 0:C] for the live rows (distinct rows: one atomic per element)
__global__ __launch_bounds__(256) void flat_rows_grad_kernel(const float* __restrict__ G, int ldG, const int32_t* __restrict__ ridx,
                                                             const int32_t* __restrict__ rows_dev, long max_rows, int C,
                                                             float* __restrict__ dfeat, int ld_d) {
    const int cq = (C + 3) >> 2;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long live = rows_dev ? ((long)*rows_dev < max_rows ? (long)*rows_dev : max_rows) : max_rows;
    if (e >= live * cq) return;
    const long r = e / cq;
    const int c = (int)(e - r * cq) * 4;
    const float* gp = G + r * (long)ldG + c;
    float* d = dfeat + (long)ridx[r] * ld_d + c;
    atomicAdd(d, gp[0]);
    if (c + 1 < C) atomicAdd(d + 1, gp[1]);
    if (c + 2 < C) atomicAdd(d + 2, gp[2]);
    if (c + 3 < C) atomicAdd(d + 3, gp[3]);
}

// Both weight images of a layer in one launch: wpack = pack(W (Nout x K), k_rot) for the forward GEMM and, when wpack_t is given,
// pack(T (kin_t x Nout)) with T[k][n] = W[n][t_col0 + k] for dgrad (a grouped first layer only needs its feature columns: t_col0 = 3).
__global__ void train_pack_kernel(const float* __restrict__ w, int Nout, int K, int k_rot, float* __restrict__ wpack,
                                  float* __restrict__ wpack_t, int kin_t, int t_col0) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int KB = (K + 7) / 8, NB = (Nout + 31) / 32;
    const long total = (long)NB * KB * 256;
    if (e < total) {
        const int s = e & 3, jj = (e >> 2) & 31, hh = (e >> 7) & 1;
        const long blk = e >> 8;
        const int kb = (int)(blk % KB), nb = (int)(blk / KB);
        const int n = nb * 32 + jj, kp = kb * 8 + 4 * hh + s;
        float val = 0.f;
        if (n < Nout && kp < K) {
            const int ko = kp < K - k_rot ? kp + k_rot : kp - (K - k_rot);
            val = w[(long)n * K + ko];
        }
        wpack[e] = val;
        return;
    }
    if (!wpack_t) return;
    const long e2 = e - total;
    const int KBt = (Nout + 7) / 8, NBt = (kin_t + 31) / 32;
    if (e2 >= (long)NBt * KBt * 256) return;
    const int s = e2 & 3, jj = (e2 >> 2) & 31, hh = (e2 >> 7) & 1;
    const long blk = e2 >> 8;
    const int kb = (int)(blk % KBt), nb = (int)(blk / KBt);
    const int kin = nb * 32 + jj, np = kb * 8 + 4 * hh + s;        // "output" index = input channel, "k" index = output channel
    wpack_t[e2] = (kin < kin_t && np < Nout) ? w[(long)np * K + t_col0 + kin] : 0.f;
}

// ---- host side: one SharedMLP stack per call (include/prcnn_pointops.h, "training-mode SharedMLP") ---------------------------
static int train_fill_src(const prcnn_train_src_t* S, MlpParams& P) {
    PRCNN_REQUIRE(S, "prcnn_train: null source descriptor");
    PRCNN_REQUIRE(S->rows > 0 && S->K > 0, "prcnn_train: bad shape rows=%ld K=%d", (long)S->rows, S->K);
    P.rows = S->rows; P.K = S->K;
    if (S->mode == MODE_PLAIN) {
        PRCNN_REQUIRE(S->in && S->ld_in >= S->K, "prcnn_train: plain source needs `in` with ld_in >= K");
        P.in = S->in; P.ld_in = S->ld_in;
        P.vec_a = aligned16(S->in) && (S->ld_in % 4 == 0);
    } else if (S->mode == MODE_GROUP) {
        PRCNN_REQUIRE(S->xyz && S->idx && (S->C == 0 || S->feat), "prcnn_train: grouped source: null pointer");
        PRCNN_REQUIRE(S->K == S->C + 3 && S->rows == (int64_t)S->B * S->M * S->ns && S->N > 0 && S->ns > 0,
                      "prcnn_train: grouped source: K must be C + 3 and rows B * M * ns");
        P.xyz = S->xyz; P.new_xyz = S->new_xyz; P.idx = S->idx; P.feat = S->feat; P.ld_feat = S->ld_feat;
        P.N = S->N; P.M = S->M; P.ns = S->ns; P.C = S->C;
        P.vec_a = S->C > 0 && aligned16(S->feat) && (S->ld_feat % 4 == 0);
    } else if (S->mode == MODE_INTERP) {
        PRCNN_REQUIRE(S->known && S->idx3 && S->w3 && (S->C1 == 0 || S->skip), "prcnn_train: interpolated source: null pointer");
        PRCNN_REQUIRE(S->K == S->C2 + S->C1 && S->rows == (int64_t)S->B * S->n && S->m > 0 && S->C2 > 0,
                      "prcnn_train: interpolated source: K must be C2 + C1 and rows B * n");
        P.known = S->known; P.idx3 = S->idx3; P.w3 = S->w3; P.skip = S->skip; P.ld_known = S->ld_known; P.ld_skip = S->ld_skip;
        P.n = S->n; P.m = S->m; P.C2 = S->C2; P.C1 = S->C1;
        P.vec_a = aligned16(S->known) && (S->ld_known % 4 == 0);
        P.vec_b = S->C1 > 0 && aligned16(S->skip) && (S->ld_skip % 4 == 0) && (S->C2 % 4 == 0);
    } else {
        return prcnn_fail(PRCNN_EINVAL, "prcnn_train: unknown source mode %d", S->mode);
    }
    return PRCNN_OK;
}

static inline size_t up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

// wgrad tile choice: per-wave vector widths (KQ, NQ) and the wave layout (WK x WN over channels, the rest over rows)
struct WgradPlan { int KQ, NQ, WK, WN, WR, tiles_k, tiles_n, splits; long rows_per_split; };
static WgradPlan wgrad_plan(long rows, int N, int K) {
    WgradPlan p;
    p.KQ = K <= 32 ? 1 : 2; p.WK = K <= 64 ? 1 : 2;
    p.NQ = N <= 32 ? 1 : 2; p.WN = N <= 64 ? 1 : 2;
    p.WR = 4 / (p.WK * p.WN);
    p.tiles_k = prcnn_divup(K, 32 * p.KQ * p.WK); p.tiles_n = prcnn_divup(N, 32 * p.NQ * p.WN);
    const long tiles = (long)p.tiles_k * p.tiles_n;
    long s = (1024 + tiles - 1) / tiles;                       // enough workgroups to fill the chip four deep
    const long by_rows = (rows + 511) / 512;                   // at least 512 rows per workgroup
    if (s > by_rows) s = by_rows;
    const long by_mem = (48L << 20) / ((long)N * K * 4 * p.WR);  // partial buffer <= 48 MB
    if (s > by_mem) s = by_mem;
    if (s < 1) s = 1;
    long per = (rows + s - 1) / s;
    per = (long)up_sz((size_t)per, (size_t)((p.WK == 2 && p.WN == 2) ? 32 : 2 * p.WR));
    p.splits = (int)((rows + per - 1) / per);
    p.rows_per_split = per;
    return p;
}

struct TrainWork { float* part; double* chunk; float* slab_w; float* wpart; float* G[2]; };
static size_t train_work_layout(int64_t rows, int nl, const prcnn_train_layer_t* L, int K0, bool backward, char* base, TrainWork* w) {
    int nmax = 0;
    size_t wg = 0;
    for (int l = 0; l < nl; l++) {
        nmax = L[l].Nout > nmax ? L[l].Nout : nmax;
        const int K = l == 0 ? K0 : L[l - 1].Nout;
        const WgradPlan p = wgrad_plan(rows, L[l].Nout, K);
        const size_t f = (size_t)p.splits * p.WR * L[l].Nout * K;
        wg = f > wg ? f : wg;
    }
    const int ldp = (int)up_sz((size_t)nmax, 4);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += up_sz(bytes, 256); return base ? base + o : (char*)nullptr; };
    char* part = take((size_t)((rows + 63) / 64) * 2 * ldp * sizeof(float));
    char* chunk = take((size_t)BN_CHUNKS * 2 * nmax * sizeof(double));
    char* slabw = take((size_t)((rows + 63) / 64) * sizeof(float));
    char *wpart = nullptr, *g0 = nullptr, *g1 = nullptr;
    if (backward) {
        wpart = take(wg * sizeof(float));
        g0 = take((size_t)rows * ldp * sizeof(float));
        g1 = take((size_t)rows * ldp * sizeof(float));
    }
    if (w) { w->part = (float*)part; w->chunk = (double*)chunk; w->slab_w = (float*)slabw; w->wpart = (float*)wpart; w->G[0] = (float*)g0; w->G[1] = (float*)g1; }
    return off;
}

static int train_check_layers(const prcnn_train_layer_t* L, int nl) {
    PRCNN_REQUIRE(L && nl > 0 && nl <= 8, "prcnn_train_stack: 1..8 layers");
    for (int l = 0; l < nl; l++) {
        PRCNN_REQUIRE(L[l].Nout > 0 && L[l].Nout % 4 == 0, "prcnn_train_stack: layer %d: Nout=%d must be a positive multiple of 4", l, L[l].Nout);
        PRCNN_REQUIRE(L[l].W && L[l].y && L[l].cst && L[l].wpack, "prcnn_train_stack: layer %d: null W / y / cst / wpack", l);
        PRCNN_REQUIRE(L[l].ld_c % 128 == 0 && L[l].ld_c >= L[l].Nout && aligned16(L[l].cst) && aligned16(L[l].y) && aligned16(L[l].wpack),
                      "prcnn_train_stack: layer %d: ld_c must be a multiple of 128 >= Nout; cst, y, wpack 16-byte aligned", l);
    }
    return PRCNN_OK;
}

PRCNN_API size_t prcnn_train_stack_work_bytes(int64_t rows, const prcnn_train_layer_t* layers, int nl, int K0, int backward) {
    if (!layers || nl <= 0 || rows <= 0) return 0;
    return train_work_layout(rows, nl, layers, K0, backward != 0, nullptr, nullptr);
}

PRCNN_API int prcnn_train_stack_fwd(const prcnn_train_src_t* src, const prcnn_train_layer_t* L, int nl, int pool_ns, float* a_dump,
                                    int ld_dump, float* out, int ld_out, int col_off, uint8_t* arg, void* work, size_t work_bytes,
                                    prcnn_stream_t stream) {
    TrainFwd T0 = {};
    int rc = train_fill_src(src, T0.P);
    if (rc) return rc;
    rc = train_check_layers(L, nl);
    if (rc) return rc;
    const long rows = src->rows;
    const bool flat = src->mult != nullptr;            // padding-free rows: variable-length groups (seg_off / seg_cnt), weighted statistics
    PRCNN_REQUIRE(!flat || (src->mode == MODE_GROUP && src->rows_dev && src->seg_off && src->seg_cnt && src->row_grp && src->groups > 0 &&
                            src->norm_rows > 0 && arg),
                  "prcnn_train_stack_fwd: padding-free rows need a grouped source with rows_dev, seg_off, seg_cnt, row_grp, groups, norm_rows, arg");
    const int ns = flat ? 1 : (pool_ns > 0 ? pool_ns : 1);
    const long norm_rows = flat ? src->norm_rows : rows;
    PRCNN_REQUIRE(out && ns <= 255 && rows % ns == 0 && (ns == 1 || arg), "prcnn_train_stack_fwd: bad pooling arguments (ns=%d)", ns);
    PRCNN_REQUIRE(ld_out % 4 == 0 && col_off % 4 == 0 && aligned16(out) && ld_out >= col_off + L[nl - 1].Nout,
                  "prcnn_train_stack_fwd: out rows must be 16-byte aligned (ld_out, col_off multiples of 4)");
    PRCNN_REQUIRE(!a_dump || (src->mode != MODE_PLAIN && ld_dump % 4 == 0 && ld_dump >= src->K && aligned16(a_dump)),
                  "prcnn_train_stack_fwd: a_dump needs a gathered source, 16-byte alignment and ld_dump %% 4 == 0, >= K");
    PRCNN_REQUIRE((src->pro_scale == nullptr) == (src->pro_shift == nullptr), "prcnn_train_stack_fwd: pro_scale and pro_shift go together");
    TrainWork W;
    const size_t need = train_work_layout(rows, nl, L, src->K, false, (char*)work, &W);
    PRCNN_REQUIRE(work && work_bytes >= need && ((uintptr_t)work & 255) == 0, "prcnn_train_stack_fwd: workspace too small or misaligned (%zu < %zu)", work_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    for (int l = 0; l < nl; l++) {
        const int K = l == 0 ? src->K : L[l - 1].Nout, N = L[l].Nout;
        const bool group0 = l == 0 && src->mode == MODE_GROUP;
        const int kin_t = group0 ? K - 3 : K;
        float* wt = (L[l].wpack_t && kin_t > 0) ? L[l].wpack_t : nullptr;
        const long tot = (long)prcnn_divup(N, 32) * prcnn_divup(K, 8) * 256 + (wt ? (long)prcnn_divup(kin_t, 32) * prcnn_divup(N, 8) * 256 : 0);
        hipLaunchKernelGGL(train_pack_kernel, dim3(prcnn_divup(tot, 256)), dim3(256), 0, s, L[l].W, N, K, group0 ? 3 : 0, L[l].wpack, wt, kin_t,
                           group0 ? 3 : 0);
        TrainFwd T = {};
        if (l == 0) {
            T = T0;
            T.pro_scale = src->mode == MODE_PLAIN ? src->pro_scale : nullptr;
            T.pro_shift = src->mode == MODE_PLAIN ? src->pro_shift : nullptr;
            T.a_dump = a_dump; T.ld_dump = ld_dump;
            if (flat) { T.P.rows_dev = src->rows_dev; T.P.rows_unit = 1; }
        } else {
            T.P.rows = rows; T.P.K = K; T.P.in = L[l - 1].y; T.P.ld_in = L[l - 1].Nout; T.P.vec_a = 1;
            T.pro_scale = L[l - 1].cst; T.pro_shift = L[l - 1].cst + L[l - 1].ld_c;
            if (flat) { T.P.rows_dev = src->rows_dev; T.P.rows_unit = 1; }
        }
        if (flat) { T.mult = src->mult; T.slab_w = W.slab_w; }
        MlpParams& P = T.P;
        P.wpack = L[l].wpack; P.Nout = N; P.out = L[l].y; P.ld_out = N; P.col_off = 0;
        P.KB = (K + 7) / 8; P.NB = (N + 31) / 32;
        const bool no_bn = L[l].gamma == nullptr;              // Conv (+ bias in `beta`) -> ReLU, no normalisation
        T.part = no_bn ? nullptr : W.part; T.ld_part = N;
        const long tiles = prcnn_divup(rows, MLP_BM);
        const bool wide = P.NB >= 4 && tiles * prcnn_divup(P.NB, 4) >= 192;
        const dim3 grid((unsigned)tiles, prcnn_divup(P.NB, wide ? 4 : 2));
        const int mode = l == 0 ? src->mode : MODE_PLAIN;
#define TRAIN_FWD(M)                                                                                            \
    do {                                                                                                        \
        if (wide) hipLaunchKernelGGL((train_fwd_kernel<M, 2>), grid, dim3(MLP_THREADS), 0, s, T);               \
        else hipLaunchKernelGGL((train_fwd_kernel<M, 1>), grid, dim3(MLP_THREADS), 0, s, T);                    \
    } while (0)
        const bool fastp = mode == MODE_PLAIN && P.vec_a && K % 4 == 0 && K >= 4 && P.ld_in % 4 == 0 && aligned16(P.in) &&
                           (!T.pro_scale || l > 0) && !getenv("PRCNN_TRAIN_FWD_GENERIC");
        if (fastp) {
            if (wide) hipLaunchKernelGGL((train_fwd_kernel<MODE_PLAIN, 2, true>), grid, dim3(MLP_THREADS), 0, s, T);
            else hipLaunchKernelGGL((train_fwd_kernel<MODE_PLAIN, 1, true>), grid, dim3(MLP_THREADS), 0, s, T);
        } else if (mode == MODE_PLAIN) TRAIN_FWD(MODE_PLAIN);
        else if (mode == MODE_GROUP) TRAIN_FWD(MODE_GROUP);
        else TRAIN_FWD(MODE_INTERP);
#undef TRAIN_FWD
        if (no_bn) {
            hipLaunchKernelGGL(nobn_cst_kernel, dim3(prcnn_divup(N, 64)), dim3(64), 0, s, L[l].beta, N, L[l].cst, L[l].ld_c);
        } else {
            hipLaunchKernelGGL(bn_chunk_kernel, dim3(prcnn_divup(N, 64), BN_CHUNKS), dim3(256), 0, s, W.part, N, rows, flat ? src->rows_dev : nullptr,
                               flat ? W.slab_w : nullptr, N, W.chunk);
            hipLaunchKernelGGL(bn_finalize_kernel, dim3(prcnn_divup(N, 64)), dim3(64), 0, s, W.chunk, norm_rows, N, L[l].gamma, L[l].beta, L[l].eps,
                               L[l].momentum, L[l].running_mean, L[l].running_var, L[l].cst, L[l].ld_c);
        }
    }
    const int N = L[nl - 1].Nout;
    const long groups = flat ? src->groups : rows / ns;
    hipLaunchKernelGGL(train_pool_kernel, dim3(prcnn_divup(groups * (N / 4), 256)), dim3(256), 0, s, L[nl - 1].y, N, groups, ns, N,
                       L[nl - 1].cst, L[nl - 1].ld_c, out, ld_out, col_off, (flat || ns > 1) ? arg : nullptr, flat ? src->seg_off : nullptr,
                       flat ? src->seg_cnt : nullptr);
    PRCNN_LAUNCH_CHECK("prcnn_train_stack_fwd");
    return PRCNN_OK;
}

template <int KQ, int NQ, int WK, int WN>
static void launch_wgrad(const TrainWgrad& W, const WgradPlan& p, hipStream_t s) {
    hipLaunchKernelGGL((train_wgrad_kernel<KQ, NQ, WK, WN>), dim3(p.splits, p.tiles_k, p.tiles_n), dim3(256), 0, s, W);
}

PRCNN_API int prcnn_train_stack_bwd(const prcnn_train_src_t* src, const prcnn_train_layer_t* L, int nl, int pool_ns, const float* a_dump,
                                    int ld_dump, const float* gout, int ld_gout, const uint8_t* arg, float* gin, int ld_gin, void* work,
                                    size_t work_bytes, prcnn_stream_t stream) {
    PRCNN_REQUIRE(src && gout, "prcnn_train_stack_bwd: null pointer");
    int rc = train_check_layers(L, nl);
    if (rc) return rc;
    const long rows = src->rows;
    const bool flat = src->mult != nullptr;
    PRCNN_REQUIRE(!flat || (src->rows_dev && src->seg_off && src->row_grp && src->norm_rows > 0 && arg),
                  "prcnn_train_stack_bwd: padding-free rows need rows_dev, seg_off, row_grp, norm_rows, arg");
    const int ns = flat ? -1 : (pool_ns > 1 ? pool_ns : 0);
    const long norm_rows = flat ? src->norm_rows : rows;
    PRCNN_REQUIRE(rows > 0 && ld_gout % 4 == 0 && aligned16(gout) && (ns <= 0 || (arg && rows % ns == 0)), "prcnn_train_stack_bwd: bad gradient / pooling arguments");
    const float* a0 = src->mode == MODE_PLAIN ? src->in : a_dump;
    const int lda0 = src->mode == MODE_PLAIN ? src->ld_in : ld_dump;
    PRCNN_REQUIRE(a0 && lda0 % 2 == 0 && ((uintptr_t)a0 & 7) == 0 && lda0 >= src->K,
                  "prcnn_train_stack_bwd: the first layer's input rows (plain `in` or a_dump) need an even row stride and 8-byte alignment");
    const int kin0 = src->mode == MODE_GROUP ? src->K - 3 : src->K;
    PRCNN_REQUIRE(!gin || (kin0 > 0 && ld_gin >= kin0 && L[0].wpack_t), "prcnn_train_stack_bwd: gin needs ld_gin >= %d and layer 0's wpack_t", kin0);
    TrainWork W;
    const size_t need = train_work_layout(rows, nl, L, src->K, true, (char*)work, &W);
    PRCNN_REQUIRE(work && work_bytes >= need && ((uintptr_t)work & 255) == 0, "prcnn_train_stack_bwd: workspace too small or misaligned (%zu < %zu)", work_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    const float* G = gout;
    int ldG = ld_gout;
    for (int l = nl - 1; l >= 0; l--) {
        const int N = L[l].Nout, K = l == 0 ? src->K : L[l - 1].Nout;
        PRCNN_REQUIRE(L[l].dW && (L[l].gamma == nullptr || L[l].dgamma) && (L[l].beta == nullptr || L[l].dbeta),
                      "prcnn_train_stack_bwd: layer %d: null gradient outputs", l);
        TrainBwd T;
        T.rows = rows; T.N = N; T.G = G; T.ldG = ldG;
        T.arg = (l == nl - 1 && ns) ? arg : nullptr; T.pool_ns = (l == nl - 1) ? ns : 0;
        T.y = L[l].y; T.ld_y = N; T.cst = L[l].cst; T.ld_c = L[l].ld_c;
        T.rows_dev = flat ? src->rows_dev : nullptr; T.mult = flat ? src->mult : nullptr;
        T.row_grp = flat ? src->row_grp : nullptr; T.seg_off = flat ? src->seg_off : nullptr;
        // BatchNorm backward reductions -> dgamma, dbeta, cst rows 4, 5
        const int ldp = (int)up_sz((size_t)N, 4);
        const long tiles = prcnn_divup(rows, 128);
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)tiles, prcnn_divup(N, 64)), dim3(256), 0, s, T, W.part, ldp);
        hipLaunchKernelGGL(bn_bwd_chunk_kernel, dim3(prcnn_divup(N, 64), BN_CHUNKS), dim3(256), 0, s, W.part, ldp, tiles, T.rows_dev, N, W.chunk);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(prcnn_divup(N, 64)), dim3(64), 0, s, W.chunk, norm_rows, N, L[l].cst, L[l].ld_c, L[l].dgamma,
                           L[l].dbeta, L[l].gamma == nullptr ? 1 : 0);
        // wgrad
        TrainWgrad Wg = {};
        Wg.B = T; Wg.K = K; Wg.part = W.wpart;
        if (l > 0) { Wg.a = L[l - 1].y; Wg.lda = L[l - 1].Nout; Wg.pro_scale = L[l - 1].cst; Wg.pro_shift = L[l - 1].cst + L[l - 1].ld_c; }
        else { Wg.a = a0; Wg.lda = lda0; }
        const WgradPlan p = wgrad_plan(rows, N, K);
        Wg.rows_per_split = p.rows_per_split;
        if (p.KQ == 1 && p.NQ == 1) launch_wgrad<1, 1, 1, 1>(Wg, p, s);
        else if (p.KQ == 1 && p.NQ == 2 && p.WN == 1) launch_wgrad<1, 2, 1, 1>(Wg, p, s);
        else if (p.KQ == 1 && p.NQ == 2) launch_wgrad<1, 2, 1, 2>(Wg, p, s);
        else if (p.KQ == 2 && p.WK == 1 && p.NQ == 1) launch_wgrad<2, 1, 1, 1>(Wg, p, s);
        else if (p.KQ == 2 && p.WK == 1 && p.WN == 1) launch_wgrad<2, 2, 1, 1>(Wg, p, s);
        else if (p.KQ == 2 && p.WK == 1) launch_wgrad<2, 2, 1, 2>(Wg, p, s);
        else if (p.NQ == 1) launch_wgrad<2, 1, 2, 1>(Wg, p, s);
        else if (p.WN == 1) launch_wgrad<2, 2, 2, 1>(Wg, p, s);
        else if (Wg.lda % 4 == 0 && aligned16(Wg.a) && !getenv("PRCNN_WGRAD_DIRECT")) {
            const dim3 wg(p.splits, p.tiles_k, p.tiles_n);
            if (Wg.B.pool_ns == 0) { if (Wg.B.mult) launch_wgrad_lds<0, true>(Wg, wg, s); else launch_wgrad_lds<0, false>(Wg, wg, s); }
            else if (Wg.B.pool_ns > 0) launch_wgrad_lds<1, false>(Wg, wg, s);
            else launch_wgrad_lds<2, true>(Wg, wg, s);
        } else launch_wgrad<2, 2, 2, 2>(Wg, p, s);
        const long count = (long)N * K;
        hipLaunchKernelGGL(train_wgrad_reduce_kernel, dim3(prcnn_divup(count, WRED_E)), dim3(256), 0, s, W.wpart, p.splits * p.WR, count, K,
                           (l == 0 && src->mode == MODE_GROUP && K > 3) ? 3 : 0, L[l].dW);
        // dgrad
        if (l == 0 && !gin) break;
        PRCNN_REQUIRE(L[l].wpack_t, "prcnn_train_stack_bwd: layer %d needs wpack_t (its input takes a gradient)", l);
        TrainDgrad D = {};
        D.B = T; D.wpack = L[l].wpack_t;
        D.Kin = l == 0 ? kin0 : K;
        D.KB = (N + 7) / 8; D.NB = (D.Kin + 31) / 32;
        D.out = l == 0 ? gin : W.G[l & 1]; D.ld_out = l == 0 ? ld_gin : K;
        const long dt = prcnn_divup(rows, MLP_BM);
        const bool wide = D.NB >= 4 && dt * prcnn_divup(D.NB, 4) >= 192;
        const dim3 grid((unsigned)dt, prcnn_divup(D.NB, wide ? 4 : 2));
#define TRAIN_DGRAD(POOL)                                                                                        \
    do {                                                                                                         \
        if (wide) hipLaunchKernelGGL((train_dgrad_kernel<2, POOL>), grid, dim3(MLP_THREADS), 0, s, D);           \
        else hipLaunchKernelGGL((train_dgrad_kernel<1, POOL>), grid, dim3(MLP_THREADS), 0, s, D);                \
    } while (0)
        if (T.pool_ns == 0) TRAIN_DGRAD(0);
        else if (T.pool_ns > 0) TRAIN_DGRAD(1);
        else TRAIN_DGRAD(2);
#undef TRAIN_DGRAD
        G = D.out; ldG = D.ld_out;
    }
    PRCNN_LAUNCH_CHECK("prcnn_train_stack_bwd");
    return PRCNN_OK;
}

PRCNN_API int prcnn_group_rows_grad(const float* G, int ldG, const int32_t* idx, int B, int M, int ns, int C, int N, float* dfeat,
                                    int ld_d, prcnn_stream_t stream) {
    PRCNN_REQUIRE(G && idx && dfeat && B >= 0 && M > 0 && ns > 0 && C > 0 && N > 0 && ldG >= C && ld_d >= C, "prcnn_group_rows_grad: bad arguments");
    PRCNN_REQUIRE(C % 4 != 0 || (ldG % 4 == 0 && aligned16(G)), "prcnn_group_rows_grad: 16-byte rows needed when C %% 4 == 0");
    const long groups = (long)B * M;
    if (groups == 0) return PRCNN_OK;
    hipLaunchKernelGGL(group_rows_grad_kernel, dim3(prcnn_divup(groups * ((C + 3) / 4), 256)), dim3(256), 0, (hipStream_t)stream, G, ldG,
                       idx, groups, M, ns, C, N, dfeat, ld_d);
    PRCNN_LAUNCH_CHECK("prcnn_group_rows_grad");
    return PRCNN_OK;
}

PRCNN_API size_t prcnn_interp_rows_grad_work_bytes(int B, int n, int m) {
    if (B <= 0 || n <= 0 || m <= 0 || m > ICSR_MAX_M || (long)n * 3 >= 2147483647L) return 0;
    return (size_t)B * (((size_t)(m + 1) * 4 + 15) / 16 * 16 + (size_t)n * 3 * 8);
}

// dknown is WRITTEN (not accumulated into) when the scratch buffer is given; without it: prcnn_interp_rows_grad (atomics into a
// zeroed dknown)
PRCNN_API int prcnn_interp_rows_grad_ws(const float* G, int ldG, const int32_t* idx3, const float* w3, int B, int n, int m, int C,
                                        float* dknown, int ld_d, void* work, size_t work_bytes, prcnn_stream_t stream) {
    PRCNN_REQUIRE(G && idx3 && w3 && dknown && B >= 0 && n > 0 && m > 0 && C > 0 && ldG >= C && ld_d >= C, "prcnn_interp_rows_grad: bad arguments");
    if (B == 0) return PRCNN_OK;
    const size_t need = prcnn_interp_rows_grad_work_bytes(B, n, m);
    const bool vec = C % 4 == 0 && ldG % 4 == 0 && ld_d % 4 == 0 && aligned16(G) && aligned16(dknown);
    if (!work || need == 0 || work_bytes < need || C > 1024 || !vec) {
        hipStream_t s = (hipStream_t)stream;
        if (hipMemsetAsync(dknown, 0, (size_t)B * m * ld_d * sizeof(float), s) != hipSuccess) return prcnn_fail(PRCNN_EHIP, "prcnn_interp_rows_grad: memset failed");
        return prcnn_interp_rows_grad(G, ldG, idx3, w3, B, n, m, C, dknown, ld_d, stream);
    }
    const size_t off_bytes = ((size_t)(m + 1) * 4 + 15) / 16 * 16;
    int32_t* off = (int32_t*)work;                                        // B x (m + 1), frames back to back ...
    int2* ent = (int2*)((char*)work + (size_t)B * off_bytes);             // ... then B x 3n entries
    // (the offsets of frame b live at off + b * (m + 1): the kernels index them that way; off_bytes only pads the total)
    hipStream_t s = (hipStream_t)stream;
    CsrFrames F = {nullptr, 0, nullptr, n, B};
    hipLaunchKernelGGL(interp_csr_build_kernel, dim3(B), dim3(ICSR_THREADS), 0, s, idx3, w3, 3, m, 0, off, ent, F);
    hipLaunchKernelGGL(interp_csr_gather_kernel, dim3(prcnn_divup((long)B * m, 4)), dim3(256), 0, s, G, ldG, (const int32_t*)off, (const int2*)ent,
                       (long)B * m, m, C, dknown, ld_d);
    PRCNN_LAUNCH_CHECK("prcnn_interp_rows_grad_ws");
    return PRCNN_OK;
}

PRCNN_API int prcnn_interp_rows_grad(const float* G, int ldG, const int32_t* idx3, const float* w3, int B, int n, int m, int C,
                                     float* dknown, int ld_d, prcnn_stream_t stream) {
    PRCNN_REQUIRE(G && idx3 && w3 && dknown && B >= 0 && n > 0 && m > 0 && C > 0 && ldG >= C && ld_d >= C, "prcnn_interp_rows_grad: bad arguments");
    const long rows = (long)B * n;
    if (rows == 0) return PRCNN_OK;
    hipLaunchKernelGGL(interp_rows_grad_kernel, dim3(prcnn_divup(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, G, ldG, idx3, w3, rows,
                       n, m, C, dknown, ld_d);
    PRCNN_LAUNCH_CHECK("prcnn_interp_rows_grad");
    return PRCNN_OK;
}


// distinct rows of every group (see train_group_count_kernel): cnt / off (B*M) i32 scratch + outputs, total -> *rows_dev;
// ridx (B*M*ns) global point index b*N + p, rnx (B*M*ns, 3) the group's centroid, mult (B*M*ns), row_grp (B*M*ns)
PRCNN_API int prcnn_train_group_rows(const int32_t* idx, const float* new_xyz, int B, int N, int M, int ns, int32_t* cnt, int32_t* off,
                                     int32_t* rows_dev, int32_t* ridx, float* rnx, float* mult, int32_t* row_grp, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B > 0 && N > 0 && M > 0 && ns > 0, "prcnn_train_group_rows: bad shape B=%d N=%d M=%d ns=%d", B, N, M, ns);
    PRCNN_REQUIRE(idx && new_xyz && cnt && off && rows_dev && ridx && rnx && mult && row_grp, "prcnn_train_group_rows: null pointer");
    PRCNN_REQUIRE((long)B * N < 2147483647L, "prcnn_train_group_rows: B * N must fit in 31 bits");
    const long groups = (long)B * M;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(train_group_count_kernel, dim3(prcnn_divup(groups, 256)), dim3(256), 0, s, idx, groups, ns, cnt);
    hipLaunchKernelGGL(train_group_scan_kernel, dim3(1), dim3(1024), 0, s, cnt, groups, off, rows_dev);
    hipLaunchKernelGGL(train_group_fill_kernel, dim3(prcnn_divup(groups, 256)), dim3(256), 0, s, idx, new_xyz, groups, M, ns, N, off, ridx, rnx,
                       mult, row_grp);
    PRCNN_LAUNCH_CHECK("prcnn_train_group_rows");
    return PRCNN_OK;
}

PRCNN_API size_t prcnn_flat_rows_grad_work_bytes(int B, int N, int64_t max_rows) {
    if (B <= 0 || N <= 0 || N > ICSR_MAX_M || max_rows <= 0 || max_rows >= 2147483647L) return 0;
    return ((size_t)B * (N + 1) * 4 + 15) / 16 * 16 + (size_t)max_rows * 8;
}

// The scatter of padding-free rows' gradients as a gather (no atomics, repeatable): the rows of frame b are those of its M groups
// (seg_off from prcnn_train_group_rows); dfeat (B * N rows) is WRITTEN, not accumulated into.  work NULL / too small / N beyond the
// LDS counters / rows not 16-byte aligned: dfeat is cleared and prcnn_flat_rows_grad runs.
PRCNN_API int prcnn_flat_rows_grad_ws(const float* G, int ldG, const int32_t* ridx, const int32_t* rows_dev, int64_t max_rows, int C, float* dfeat,
                                      int ld_d, const int32_t* seg_off, int B, int N, int M, void* work, size_t work_bytes, prcnn_stream_t stream) {
    PRCNN_REQUIRE(G && ridx && rows_dev && dfeat && seg_off && max_rows >= 0 && C > 0 && ldG >= C && ld_d >= C && B > 0 && N > 0 && M > 0,
                  "prcnn_flat_rows_grad_ws: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const size_t need = prcnn_flat_rows_grad_work_bytes(B, N, max_rows);
    const bool vec = C % 4 == 0 && ldG % 4 == 0 && ld_d % 4 == 0 && aligned16(G) && aligned16(dfeat);
    if (!work || need == 0 || work_bytes < need || C > 1024 || !vec) {
        if (hipMemsetAsync(dfeat, 0, (size_t)B * N * ld_d * sizeof(float), s) != hipSuccess) return prcnn_fail(PRCNN_EHIP, "prcnn_flat_rows_grad: memset failed");
        return prcnn_flat_rows_grad(G, ldG, ridx, rows_dev, max_rows, C, dfeat, ld_d, stream);
    }
    int32_t* off = (int32_t*)work;
    int2* ent = (int2*)((char*)work + ((size_t)B * (N + 1) * 4 + 15) / 16 * 16);
    CsrFrames F = {seg_off, M, rows_dev, 0, B};
    hipLaunchKernelGGL(interp_csr_build_kernel, dim3(B), dim3(ICSR_THREADS), 0, s, ridx, (const float*)nullptr, 1, N, N, off, ent, F);
    hipLaunchKernelGGL(interp_csr_gather_kernel, dim3(prcnn_divup((long)B * N, 4)), dim3(256), 0, s, G, ldG, (const int32_t*)off, (const int2*)ent,
                       (long)B * N, N, C, dfeat, ld_d);
    PRCNN_LAUNCH_CHECK("prcnn_flat_rows_grad_ws");
    return PRCNN_OK;
}

PRCNN_API int prcnn_flat_rows_grad(const float* G, int ldG, const int32_t* ridx, const int32_t* rows_dev, int64_t max_rows, int C, float* dfeat,
                                   int ld_d, prcnn_stream_t stream) {
    PRCNN_REQUIRE(G && ridx && dfeat && max_rows >= 0 && C > 0 && ldG >= C && ld_d >= C, "prcnn_flat_rows_grad: bad arguments");
    if (max_rows == 0) return PRCNN_OK;
    hipLaunchKernelGGL(flat_rows_grad_kernel, dim3(prcnn_divup(max_rows * ((C + 3) / 4), 256)), dim3(256), 0, (hipStream_t)stream, G, ldG, ridx,
                       rows_dev, (long)max_rows, C, dfeat, ld_d);
    PRCNN_LAUNCH_CHECK("prcnn_flat_rows_grad");
    return PRCNN_OK;
}
