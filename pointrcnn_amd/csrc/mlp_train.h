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
};

template <int MODE, int WNB>
__global__ __launch_bounds__(MLP_THREADS, ((WNB == 1 && MODE != MODE_INTERP) ? 3 : 2)) void train_fwd_kernel(const TrainFwd T) {
    const MlpParams& P = T.P;
    constexpr int QN = 2 * WNB;
    const long tile_id = blockIdx.x;
    const int nb0 = blockIdx.y * QN;
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
#pragma unroll
        for (int u = 0; u < 4; u++) fetch<MODE>(P, meta[u], k, ra[u]);
    };
    auto store_chunk = [&](int c, int buf) {
        const int k = c * MLP_BK + c4 * 4;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float4 v = finish<MODE>(P, meta[u], k, ra[u]);
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
    if (!n_active) return;
    // ---- epilogue: raw store + column statistics of this wave's 64-row slab ---------------------------------------------
    const long wrow0 = row0 + wm * 64;
    const long left = P.rows - wrow0;
    const int cnt = left >= 64 ? 64 : (left > 0 ? (int)left : 0);          // live rows of the slab (wave-uniform)
    const float inv_cnt = cnt > 0 ? 1.0f / (float)cnt : 0.f;
#pragma unroll
    for (int nn = 0; nn < WNB; nn++) {
        const int nb = nb0 + wn * WNB + nn;
        if (nb >= P.NB) continue;
        const int n = nb * 32 + j;
        const bool n_ok = n < P.Nout;
        const f32x16& a0 = acc[0][nn];
        const f32x16& a1 = acc[1][nn];
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
            const long g0 = wrow0 + rin, g1 = g0 + 32;
            if (g0 < P.rows) { s += a0[r]; if (n_ok) P.out[g0 * P.ld_out + P.col_off + n] = a0[r]; }
            if (g1 < P.rows) { s += a1[r]; if (n_ok) P.out[g1 * P.ld_out + P.col_off + n] = a1[r]; }
        }
        if (T.part) {
            s += __shfl_xor(s, 32);
            const float mean = s * inv_cnt;
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
                const long g0 = wrow0 + rin, g1 = g0 + 32;
                if (g0 < P.rows) { const float d = a0[r] - mean; m2 += d * d; }
                if (g1 < P.rows) { const float d = a1[r] - mean; m2 += d * d; }
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

// Batch statistics from the slab partials.  Block = 64 columns x 16 slab lanes (1024 threads); everything in double, combined in a
// fixed order.  cst rows (ld_c floats each): 0 scale = gamma * invstd, 1 shift = beta - mean * scale, 2 mean, 3 invstd.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int ld_part, long rows, int N,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float momentum, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ cst, int ld_c) {
    __shared__ double s1[16][64], s2[16][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const long nslab = (rows + 63) >> 6;
    double a = 0.0, b = 0.0;
    if (n < N) {
        for (long sl = q; sl < nslab; sl += 16) {
            const double cn = (double)(sl == nslab - 1 ? rows - sl * 64 : 64);
            const double m = (double)part[(sl * 2 + 0) * ld_part + n], m2 = (double)part[(sl * 2 + 1) * ld_part + n];
            a += cn * m;
            b += m2 + cn * m * m;
        }
    }
    s1[q][c] = a; s2[q][c] = b;
    __syncthreads();
    if (q == 0 && n < N) {
        double A = 0.0, Bq = 0.0;
        for (int t = 0; t < 16; t++) { A += s1[t][c]; Bq += s2[t][c]; }
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
}

// out[g, col_off + n] = max_s relu(y[g * ns + s, n] * scale[n] + shift[n]); arg[g, n] = the FIRST s that attains it (torch.max /
// max_pool2d backward route the gradient there).  ns == 1: plain normalise + ReLU, no arg.  Thread = (group, 4 channels).
__global__ __launch_bounds__(256) void train_pool_kernel(const float* __restrict__ y, int ld_y, long groups, int ns, int N,
                                                         const float* __restrict__ cst, int ld_c, float* __restrict__ out, int ld_out,
                                                         int col_off, uint8_t* __restrict__ arg) {
    const int nq = N >> 2;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= groups * nq) return;
    const long g = e / nq;
    const int n = (int)(e - g * nq) * 4;
    const float4 sc = ld4(cst + n), sh = ld4(cst + ld_c + n);
    const float* p = y + g * ns * (long)ld_y + n;
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
};

__device__ __forceinline__ float4 bwd_G4(const TrainBwd& T, long r, int n) {
    if (T.pool_ns == 0) return ld4(T.G + r * (long)T.ldG + n);
    const long g = r / T.pool_ns;
    const int s = (int)(r - g * T.pool_ns);
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
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (n < T.N) {
        const float4 sc = ld4(T.cst + n), sh = ld4(T.cst + T.ld_c + n), mu = ld4(T.cst + 2 * T.ld_c + n), is = ld4(T.cst + 3 * T.ld_c + n);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const long r = row0 + rl + 16 * u;
            if (r >= T.rows) break;
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

// tile partials -> dbeta = sum dyhat, dgamma = sum dyhat * xhat (double, fixed order); cst rows 4, 5 = their means over the rows
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ part, int ld_part, long tiles, long rows, int N,
                                                               float* __restrict__ cst, int ld_c, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
    __shared__ double s1[16][64], s2[16][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    double a = 0.0, b = 0.0;
    if (n < N)
        for (long t = q; t < tiles; t += 16) { a += (double)part[(t * 2 + 0) * ld_part + n]; b += (double)part[(t * 2 + 1) * ld_part + n]; }
    s1[q][c] = a; s2[q][c] = b;
    __syncthreads();
    if (q == 0 && n < N) {
        double A = 0.0, Bq = 0.0;
        for (int t = 0; t < 16; t++) { A += s1[t][c]; Bq += s2[t][c]; }
        if (dbeta) dbeta[n] = (float)A;
        if (dgamma) dgamma[n] = (float)Bq;
        cst[4 * ld_c + n] = (float)(A / (double)rows);
        cst[5 * ld_c + n] = (float)(Bq / (double)rows);
    }
}

// dy for 4 consecutive channels: scale * (dyhat - c1 - xhat * c2)
__device__ __forceinline__ float4 bwd_dy4(const float4 g, const float4 yv, const float4 sc, const float4 sh, const float4 mu,
                                          const float4 is, const float4 c1, const float4 c2) {
    float4 o;
    o.x = sc.x * (((yv.x * sc.x + sh.x > 0.f) ? g.x : 0.f) - c1.x - ((yv.x - mu.x) * is.x) * c2.x);
    o.y = sc.y * (((yv.y * sc.y + sh.y > 0.f) ? g.y : 0.f) - c1.y - ((yv.y - mu.y) * is.y) * c2.y);
    o.z = sc.z * (((yv.z * sc.z + sh.z > 0.f) ? g.z : 0.f) - c1.z - ((yv.z - mu.z) * is.z) * c2.z);
    o.w = sc.w * (((yv.w * sc.w + sh.w > 0.f) ? g.w : 0.f) - c1.w - ((yv.w - mu.w) * is.w) * c2.w);
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
template <int WNB>
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
    long grow[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { grow[u] = row0 + r0 + 32 * u; if (grow[u] >= T.rows) grow[u] = T.rows - 1; }
    float4 rg[4], ry[4], sc, sh, mu, is, c1, c2;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_chunk = [&](int c) {
        const int k = c * MLP_BK + c4 * 4;
        if (k < T.N) {
            sc = ld4(T.cst + k); sh = ld4(T.cst + T.ld_c + k); mu = ld4(T.cst + 2 * T.ld_c + k); is = ld4(T.cst + 3 * T.ld_c + k);
            c1 = ld4(T.cst + 4 * T.ld_c + k); c2 = ld4(T.cst + 5 * T.ld_c + k);
#pragma unroll
            for (int u = 0; u < 4; u++) { rg[u] = bwd_G4(T, grow[u], k); ry[u] = ld4(T.y + grow[u] * (long)T.ld_y + k); }
        } else {
            sc = sh = mu = is = c1 = c2 = zero4;
#pragma unroll
            for (int u = 0; u < 4; u++) { rg[u] = zero4; ry[u] = zero4; }
        }
    };
    auto store_chunk = [&](int c, int buf) {
#pragma unroll
        for (int u = 0; u < 4; u++)
            *reinterpret_cast<float4*>(&As[buf][(r0 + 32 * u) * MLP_ALD + c4 * 4]) = bwd_dy4(rg[u], ry[u], sc, sh, mu, is, c1, c2);
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
        if (n >= D.Kin) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rin = (r & 3) + 8 * (r >> 2) + 4 * h;
            const long g0 = wrow0 + rin, g1 = g0 + 32;
            if (g0 < T.rows) D.out[g0 * D.ld_out + n] = acc[0][nn][r];
            if (g1 < T.rows) D.out[g1 * D.ld_out + n] = acc[1][nn][r];
        }
    }
}

// dW[n, k] = sum_r dy[r, n] * a[r, k].  Workgroup tile: 128 input channels (k) x 128 output channels (n); wave (kh, nh) owns the
// 64 x 64 quadrant as 2 x 2 MFMA blocks.  Per step a wave consumes TWO rows: lane (h, j) loads channels 2j, 2j+1 of its k-half from
// row r + h (8 bytes; the 32 lanes of a half read 256 contiguous bytes) and the same for dy -- element x of the pair feeds block x,
// whose C row / column i therefore stands for channel 2i + x.  Rows are dealt to the grid's x dimension in contiguous ranges; each
// workgroup writes its partial tile, train_wgrad_reduce_kernel sums the partials in a fixed order.
struct TrainWgrad {
    TrainBwd B;
    const float* a; int lda; int K;          // a rows (rows x lda), K valid channels; lda even
    const float* pro_scale;                  // optional relu(a * scale + shift) (the forward's prologue), padded to a multiple of 128
    const float* pro_shift;
    long rows_per_split;                     // even
    float* part;                             // (splits, N, K) partial dW
};
#define WG_UNROLL 4
__global__ __launch_bounds__(256, 2) void train_wgrad_kernel(const TrainWgrad W) {
    const TrainBwd& T = W.B;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 1, nh = wave & 1;
    const int h = lane >> 5, j = lane & 31;
    const int k = blockIdx.y * 128 + kh * 64 + 2 * j;              // this lane's first input channel
    const int n = blockIdx.z * 128 + nh * 64 + 2 * j;              // ... and first output channel
    const bool k_ok = k < W.lda, n_ok = n < T.N;                   // (lda and N are even: a pair is in or out as a whole)
    const float kmx = k < W.K ? 1.f : 0.f, kmy = k + 1 < W.K ? 1.f : 0.f;
    float2 ps = make_float2(1.f, 1.f), pb = make_float2(0.f, 0.f);
    const bool pro = W.pro_scale != nullptr;
    if (pro) { ps = *reinterpret_cast<const float2*>(W.pro_scale + k); pb = *reinterpret_cast<const float2*>(W.pro_shift + k); }
    float2 sc = make_float2(0.f, 0.f), sh = sc, mu = sc, is = sc, c1 = sc, c2 = sc;
    if (n_ok) {
        sc = *reinterpret_cast<const float2*>(T.cst + n); sh = *reinterpret_cast<const float2*>(T.cst + T.ld_c + n);
        mu = *reinterpret_cast<const float2*>(T.cst + 2 * T.ld_c + n); is = *reinterpret_cast<const float2*>(T.cst + 3 * T.ld_c + n);
        c1 = *reinterpret_cast<const float2*>(T.cst + 4 * T.ld_c + n); c2 = *reinterpret_cast<const float2*>(T.cst + 5 * T.ld_c + n);
    }
    const long r_begin = (long)blockIdx.x * W.rows_per_split;
    long r_end = r_begin + W.rows_per_split;
    if (r_end > T.rows) r_end = T.rows;
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int z = 0; z < 2; z++) acc[x][z] = (f32x16){0};
    const int kk = k_ok ? k : 0, nn = n_ok ? n : 0;
    for (long r = r_begin; r < r_end; r += 2 * WG_UNROLL) {
        float2 av[WG_UNROLL], gv[WG_UNROLL], yv[WG_UNROLL];
        float valid[WG_UNROLL];
#pragma unroll
        for (int u = 0; u < WG_UNROLL; u++) {
            long rr = r + 2 * u + h;
            valid[u] = rr < r_end ? 1.f : 0.f;
            if (rr >= r_end) rr = r_end - 1;
            av[u] = *reinterpret_cast<const float2*>(W.a + rr * (long)W.lda + kk);
            yv[u] = *reinterpret_cast<const float2*>(T.y + rr * (long)T.ld_y + nn);
            if (T.pool_ns == 0) {
                gv[u] = *reinterpret_cast<const float2*>(T.G + rr * (long)T.ldG + nn);
            } else {
                const long g = rr / T.pool_ns;
                const int s = (int)(rr - g * T.pool_ns);
                const float2 v = *reinterpret_cast<const float2*>(T.G + g * (long)T.ldG + nn);
                const uchar2 ag = *reinterpret_cast<const uchar2*>(T.arg + g * (long)T.N + nn);
                gv[u] = make_float2(ag.x == s ? v.x : 0.f, ag.y == s ? v.y : 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < WG_UNROLL; u++) {
            float2 a = av[u];
            if (pro) { a.x = fmaxf(a.x * ps.x + pb.x, 0.f); a.y = fmaxf(a.y * ps.y + pb.y, 0.f); }
            a.x *= kmx * valid[u]; a.y *= kmy * valid[u];           // channels past K and rows past the range contribute zero
            float2 d;
            d.x = sc.x * (((yv[u].x * sc.x + sh.x > 0.f) ? gv[u].x : 0.f) - c1.x - ((yv[u].x - mu.x) * is.x) * c2.x);
            d.y = sc.y * (((yv[u].y * sc.y + sh.y > 0.f) ? gv[u].y : 0.f) - c1.y - ((yv[u].y - mu.y) * is.y) * c2.y);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, d.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, d.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, d.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, d.y, acc[1][1], 0, 0, 0);
        }
    }
    // C[i][jj] of block (x, z): input channel kbase + 2 i + x, output channel nbase + 2 jj + z
    float* P = W.part + (long)blockIdx.x * T.N * W.K;
    const int kbase = blockIdx.y * 128 + kh * 64, nbase = blockIdx.z * 128 + nh * 64;
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

__global__ void train_wgrad_reduce_kernel(const float* __restrict__ part, int splits, long count, float* __restrict__ out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    float s = 0.f;
    for (int t = 0; t < splits; t++) s += part[(long)t * count + e];
    out[e] = s;
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


// ---- C ABI (include/prcnn_pointops.h, "training-mode SharedMLP") -------------------------------------------------------------
static int train_fill_src(const prcnn_train_src_t* S, MlpParams& P) {
    PRCNN_REQUIRE(S, "prcnn_train: null source descriptor");
    PRCNN_REQUIRE(S->rows >= 0 && S->K > 0, "prcnn_train: bad shape rows=%ld K=%d", (long)S->rows, S->K);
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

PRCNN_API size_t prcnn_train_part_floats(int64_t rows, int ld_part) { return (size_t)((rows + 63) / 64) * 2 * (size_t)ld_part; }

PRCNN_API int prcnn_train_fwd(const prcnn_train_src_t* src, const float* wpack, int Nout, float* y, int ld_y, float* a_dump,
                              int ld_dump, float* part, int ld_part, prcnn_stream_t stream) {
    TrainFwd T = {};
    int rc = train_fill_src(src, T.P);
    if (rc) return rc;
    MlpParams& P = T.P;
    PRCNN_REQUIRE(wpack && y && Nout > 0 && ld_y >= Nout && aligned16(wpack), "prcnn_train_fwd: bad output / weight arguments");
    PRCNN_REQUIRE(!part || ld_part >= Nout, "prcnn_train_fwd: ld_part %d < Nout %d", ld_part, Nout);
    PRCNN_REQUIRE(!a_dump || (src->mode != MODE_PLAIN && ld_dump % 4 == 0 && ld_dump >= src->K && aligned16(a_dump)),
                  "prcnn_train_fwd: a_dump needs a gathered source, 16-byte alignment and ld_dump %% 4 == 0, >= K");
    PRCNN_REQUIRE((src->pro_scale == nullptr) == (src->pro_shift == nullptr), "prcnn_train_fwd: pro_scale and pro_shift go together");
    if (P.rows == 0) return PRCNN_OK;
    P.wpack = wpack; P.Nout = Nout; P.out = y; P.ld_out = ld_y; P.col_off = 0;
    P.KB = (P.K + 7) / 8; P.NB = (Nout + 31) / 32;
    T.pro_scale = src->mode == MODE_PLAIN ? src->pro_scale : nullptr; T.pro_shift = src->mode == MODE_PLAIN ? src->pro_shift : nullptr;
    T.a_dump = a_dump; T.ld_dump = ld_dump; T.part = part; T.ld_part = ld_part;
    const long tiles = prcnn_divup(P.rows, MLP_BM);
    const bool wide = P.NB >= 4 && tiles * prcnn_divup(P.NB, 4) >= 192;
    const dim3 grid((unsigned)tiles, prcnn_divup(P.NB, wide ? 4 : 2));
    hipStream_t s = (hipStream_t)stream;
#define TRAIN_FWD(M)                                                                                            \
    do {                                                                                                        \
        if (wide) hipLaunchKernelGGL((train_fwd_kernel<M, 2>), grid, dim3(MLP_THREADS), 0, s, T);               \
        else hipLaunchKernelGGL((train_fwd_kernel<M, 1>), grid, dim3(MLP_THREADS), 0, s, T);                    \
    } while (0)
    if (src->mode == MODE_PLAIN) TRAIN_FWD(MODE_PLAIN);
    else if (src->mode == MODE_GROUP) TRAIN_FWD(MODE_GROUP);
    else TRAIN_FWD(MODE_INTERP);
#undef TRAIN_FWD
    PRCNN_LAUNCH_CHECK("prcnn_train_fwd");
    return PRCNN_OK;
}

PRCNN_API int prcnn_train_bn_finalize(const float* part, int ld_part, int64_t rows, int N, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var, float* cst, int ld_c,
                                      prcnn_stream_t stream) {
    PRCNN_REQUIRE(part && cst && rows > 0 && N > 0 && ld_part >= N && ld_c >= N, "prcnn_train_bn_finalize: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(prcnn_divup(N, 64)), dim3(1024), 0, (hipStream_t)stream, part, ld_part, (long)rows, N,
                       gamma, beta, eps, momentum, running_mean, running_var, cst, ld_c);
    PRCNN_LAUNCH_CHECK("prcnn_train_bn_finalize");
    return PRCNN_OK;
}

PRCNN_API int prcnn_train_pool(const float* y, int ld_y, int64_t groups, int ns, int N, const float* cst, int ld_c, float* out,
                               int ld_out, int col_off, uint8_t* arg, prcnn_stream_t stream) {
    PRCNN_REQUIRE(y && cst && out && groups >= 0 && ns > 0 && ns <= 255 && N > 0, "prcnn_train_pool: bad arguments");
    PRCNN_REQUIRE(N % 4 == 0 && ld_y % 4 == 0 && ld_out % 4 == 0 && col_off % 4 == 0 && ld_c % 4 == 0 && aligned16(y) && aligned16(out) &&
                      aligned16(cst) && ld_out >= col_off + N,
                  "prcnn_train_pool: channels, strides and offsets must be multiples of 4 floats (16-byte rows)");
    if (groups == 0) return PRCNN_OK;
    hipLaunchKernelGGL(train_pool_kernel, dim3(prcnn_divup(groups * (N / 4), 256)), dim3(256), 0, (hipStream_t)stream, y, ld_y,
                       (long)groups, ns, N, cst, ld_c, out, ld_out, col_off, arg);
    PRCNN_LAUNCH_CHECK("prcnn_train_pool");
    return PRCNN_OK;
}

static int train_fill_grad(const prcnn_train_grad_t* g, TrainBwd& T) {
    PRCNN_REQUIRE(g && g->G && g->y && g->cst, "prcnn_train backward: null pointer");
    PRCNN_REQUIRE(g->rows > 0 && g->N > 0 && g->N % 4 == 0 && g->ldG % 4 == 0 && g->ld_y % 4 == 0 && g->ld_c % 4 == 0 && g->ld_c >= g->N,
                  "prcnn_train backward: N, ldG, ld_y, ld_c must be multiples of 4 (N=%d)", g->N);
    PRCNN_REQUIRE(aligned16(g->G) && aligned16(g->y) && aligned16(g->cst), "prcnn_train backward: 16-byte aligned G / y / cst");
    PRCNN_REQUIRE(g->pool_ns == 0 || (g->arg && g->rows % g->pool_ns == 0), "prcnn_train backward: pooled layer needs arg and rows %% ns == 0");
    T.rows = g->rows; T.N = g->N; T.G = g->G; T.ldG = g->ldG; T.arg = g->arg; T.pool_ns = g->pool_ns; T.y = g->y; T.ld_y = g->ld_y;
    T.cst = g->cst; T.ld_c = g->ld_c;
    return PRCNN_OK;
}

PRCNN_API size_t prcnn_train_bwd_part_floats(int64_t rows, int ld_part) { return (size_t)((rows + 127) / 128) * 2 * (size_t)ld_part; }

PRCNN_API int prcnn_train_bn_backward(const prcnn_train_grad_t* g, float* part, int ld_part, float* dgamma, float* dbeta,
                                      prcnn_stream_t stream) {
    TrainBwd T;
    int rc = train_fill_grad(g, T);
    if (rc) return rc;
    PRCNN_REQUIRE(part && ld_part >= T.N && ld_part % 4 == 0 && aligned16(part), "prcnn_train_bn_backward: bad partial buffer");
    const long tiles = prcnn_divup(T.rows, 128);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)tiles, prcnn_divup(T.N, 64)), dim3(256), 0, s, T, part, ld_part);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(prcnn_divup(T.N, 64)), dim3(1024), 0, s, part, ld_part, tiles, T.rows, T.N,
                       g->cst, T.ld_c, dgamma, dbeta);
    PRCNN_LAUNCH_CHECK("prcnn_train_bn_backward");
    return PRCNN_OK;
}

PRCNN_API int prcnn_train_dgrad(const prcnn_train_grad_t* g, const float* wpack_t, int Kin, float* out, int ld_out, prcnn_stream_t stream) {
    TrainDgrad D = {};
    int rc = train_fill_grad(g, D.B);
    if (rc) return rc;
    PRCNN_REQUIRE(wpack_t && out && Kin > 0 && ld_out >= Kin && aligned16(wpack_t), "prcnn_train_dgrad: bad arguments");
    D.wpack = wpack_t; D.Kin = Kin; D.KB = (D.B.N + 7) / 8; D.NB = (Kin + 31) / 32; D.out = out; D.ld_out = ld_out;
    const long tiles = prcnn_divup(D.B.rows, MLP_BM);
    const bool wide = D.NB >= 4 && tiles * prcnn_divup(D.NB, 4) >= 192;
    const dim3 grid((unsigned)tiles, prcnn_divup(D.NB, wide ? 4 : 2));
    if (wide) hipLaunchKernelGGL(train_dgrad_kernel<2>, grid, dim3(MLP_THREADS), 0, (hipStream_t)stream, D);
    else hipLaunchKernelGGL(train_dgrad_kernel<1>, grid, dim3(MLP_THREADS), 0, (hipStream_t)stream, D);
    PRCNN_LAUNCH_CHECK("prcnn_train_dgrad");
    return PRCNN_OK;
}

PRCNN_API int prcnn_train_wgrad_splits(int64_t rows, int N, int K) {
    // enough workgroups to fill the chip (>= ~768 over all tiles), at least 256 rows each, partial buffer <= 64 MB
    const long tiles = (long)prcnn_divup(K, 128) * prcnn_divup(N, 128);
    long s = (768 + tiles - 1) / tiles;
    const long by_rows = (rows + 255) / 256;
    if (s > by_rows) s = by_rows;
    const long by_mem = (64L << 20) / ((long)N * K * 4);
    if (s > by_mem) s = by_mem;
    return (int)(s < 1 ? 1 : s);
}

PRCNN_API int prcnn_train_wgrad(const prcnn_train_grad_t* g, const float* a, int lda, int K, const float* pro_scale,
                                const float* pro_shift, float* part, int splits, float* dW, prcnn_stream_t stream) {
    TrainWgrad W = {};
    int rc = train_fill_grad(g, W.B);
    if (rc) return rc;
    PRCNN_REQUIRE(a && part && dW && K > 0 && lda >= K && lda % 2 == 0 && splits > 0 && ((uintptr_t)a & 7) == 0,
                  "prcnn_train_wgrad: bad arguments (lda must be even, a 8-byte aligned)");
    PRCNN_REQUIRE((pro_scale == nullptr) == (pro_shift == nullptr), "prcnn_train_wgrad: pro_scale and pro_shift go together");
    W.a = a; W.lda = lda; W.K = K; W.pro_scale = pro_scale; W.pro_shift = pro_shift; W.part = part;
    long per = (W.B.rows + splits - 1) / splits;
    per = (per + 1) & ~1L;
    W.rows_per_split = per;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(train_wgrad_kernel, dim3(splits, prcnn_divup(K, 128), prcnn_divup(W.B.N, 128)), dim3(256), 0, s, W);
    const long count = (long)W.B.N * K;
    hipLaunchKernelGGL(train_wgrad_reduce_kernel, dim3(prcnn_divup(count, 256)), dim3(256), 0, s, part, splits, count, dW);
    PRCNN_LAUNCH_CHECK("prcnn_train_wgrad");
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
