// gather.hip -- gather_operation / grouping_operation / three_interpolate and their backward passes,
// in the reference op surface's channel-first layout (B,C,N).
//
// Replaces pointnet2_cuda.{gather_points,group_points,three_interpolate}{,_grad}_wrapper [UPSTREAM, not in
// tree]; semantics per SURVEY Appendix A.2 / A.4 / A.6.  These are the drop-in ops for callers that use
// the op surface directly (and for training); the inference fast path never materialises grouped tensors
// (mlp.hip gathers straight into the MFMA A-tile).
//
// HBM-bound copies: one thread per output (m[,s]) position, looping over channels with the index held
// in a register, so index traffic is read once (upstream re-reads idx for every channel) and the writes
// are fully coalesced; the gathered reads are scattered by nature of the (B,C,N) layout.
#include "common.h"
#include <stdlib.h>

#define G_THREADS 256

__global__ __launch_bounds__(G_THREADS) void gather_kernel(const float* __restrict__ feat,
                                                           const int32_t* __restrict__ idx, int C, int N, int J,
                                                           float* __restrict__ out) {
    // J = outputs per (b,c) plane: M for gather, M*ns for group
    const int b = blockIdx.y;
    const int j = blockIdx.x * G_THREADS + threadIdx.x;
    if (j >= J) return;
    const int id = idx[(size_t)b * J + j];
    const float* f = feat + (size_t)b * C * N + id;
    float* o = out + (size_t)b * C * J + j;
#pragma unroll 4
    for (int c = 0; c < C; c++) o[(size_t)c * J] = f[(size_t)c * N];
}

// grouping_operation / gather_operation with the source rows staged in LDS (round 6; the three_interpolate kernel's plan, one source point
// per output instead of three): in the op surface's (B, C, N) layout the plain kernel above reads 64 unrelated addresses of one 4 N-byte
// row per load instruction -- 64 separate cache accesses for 256 bytes.  A workgroup here owns CGT consecutive channels of one frame,
// staged POINT-major (CGT floats per source point, contiguous: one or two 16-byte LDS reads fetch all of a point's channels), and a slice
// of the J outputs: four consecutive outputs per thread (one 16-byte index load, one 16-byte store per channel).  Pure copies: exact.
#define GP_THREADS 1024
template <int CGT>
__global__ __launch_bounds__(GP_THREADS) void gather_pm_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx, int C, int N,
                                                               int J, int jslices, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float gpm[];     // N rows of CGT floats
    const int b = blockIdx.y, cgi = blockIdx.x / jslices, js = blockIdx.x - cgi * jslices, c0 = cgi * CGT;
    const float* f = feat + ((size_t)b * C + c0) * N;
    for (int i0 = threadIdx.x; i0 < N; i0 += 4 * GP_THREADS) {
        float v[4][CGT];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int i = min(i0 + p * GP_THREADS, N - 1);
#pragma unroll
            for (int c = 0; c < CGT; c++) v[p][c] = f[(size_t)c * N + i];
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int i = i0 + p * GP_THREADS;
            if (i < N)
#pragma unroll
                for (int q = 0; q < CGT / 4; q++)
                    *reinterpret_cast<float4*>(gpm + (size_t)i * CGT + 4 * q) = make_float4(v[p][4 * q], v[p][4 * q + 1], v[p][4 * q + 2], v[p][4 * q + 3]);
        }
    }
    __syncthreads();
    float* o = out + ((size_t)b * C + c0) * J;
    const int32_t* __restrict__ ib = idx + (size_t)b * J;
    // this workgroup's slice of the outputs, in units of 4: [q0, q1)
    const int quads = J / 4, per = (quads + jslices - 1) / jslices;
    const int q0 = js * per, q1 = min(quads, q0 + per);
    int qd = q0 + threadIdx.x;
    if (qd >= q1) return;
    int4 id = *reinterpret_cast<const int4*>(ib + 4 * (size_t)qd);
    while (true) {
        const int nq = qd + GP_THREADS;
        const int4 nid = *reinterpret_cast<const int4*>(ib + 4 * (size_t)(nq < q1 ? nq : qd));
        float r[CGT][4];
        const int ii[4] = {id.x, id.y, id.z, id.w};
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
            for (int q = 0; q < CGT / 4; q++) {
                const float4 a = *reinterpret_cast<const float4*>(gpm + (size_t)ii[p] * CGT + 4 * q);
                r[4 * q][p] = a.x; r[4 * q + 1][p] = a.y; r[4 * q + 2][p] = a.z; r[4 * q + 3][p] = a.w;
            }
#pragma unroll
        for (int c = 0; c < CGT; c++) *reinterpret_cast<float4*>(o + (size_t)c * J + 4 * (size_t)qd) = make_float4(r[c][0], r[c][1], r[c][2], r[c][3]);
        if (nq >= q1) break;
        qd = nq; id = nid;
    }
}
// -> true when the staged kernel took the call
static bool launch_gather_pm(const float* feat, const int32_t* idx, int B, int C, int N, long J, float* out, hipStream_t s) {
    if (getenv("PRCNN_GATHER_DIRECT") != nullptr) return false;                          // A/B switch (same values)
    const int CGT = 8;
    if (C % CGT || J % 4 || (long)N * CGT * 4 > 128 * 1024 || J < 4L * N || J > 0x7fffffffL) return false;    // (staging must pay: >= 4 outputs per source point)
    if ((((uintptr_t)idx | (uintptr_t)out) & 15) != 0) return false;
    static PrcnnLdsLimit lim;
    if (!lim.raise((const void*)gather_pm_kernel<8>, 128 * 1024)) return false;
    // enough workgroups for the chip: slices of the outputs when B x C / 8 alone is short of ~4 per CU
    int jslices = 1;
    while ((long)B * (C / CGT) * jslices < 1024 && (J / 4) / (jslices * 2) >= 2 * GP_THREADS) jslices *= 2;
    hipLaunchKernelGGL(gather_pm_kernel<8>, dim3((C / CGT) * jslices, B), dim3(GP_THREADS), (size_t)CGT * N * sizeof(float), s, feat, idx, C, N,
                       (int)J, jslices, out);
    return true;
}

__global__ __launch_bounds__(G_THREADS) void gather_grad_kernel(const float* __restrict__ grad_out,
                                                                const int32_t* __restrict__ idx, int C, int N, int J,
                                                                float* __restrict__ grad_feat) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * G_THREADS + threadIdx.x;
    if (j >= J) return;
    const int id = idx[(size_t)b * J + j];
    const float* g = grad_out + (size_t)b * C * J + j;
    float* gf = grad_feat + (size_t)b * C * N + id;
    for (int c = 0; c < C; c++) atomicAdd(gf + (size_t)c * N, g[(size_t)c * J]);
}

__global__ __launch_bounds__(G_THREADS) void three_interp_kernel(const float* __restrict__ feat,
                                                                 const int32_t* __restrict__ idx,
                                                                 const float* __restrict__ w, int C, int m, int n,
                                                                 float* __restrict__ out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * G_THREADS + threadIdx.x;
    if (i >= n) return;
    const size_t o3 = ((size_t)b * n + i) * 3;
    const int i0 = idx[o3], i1 = idx[o3 + 1], i2 = idx[o3 + 2];
    const float w0 = w[o3], w1 = w[o3 + 1], w2 = w[o3 + 2];
    const float* f = feat + (size_t)b * C * m;
    float* o = out + (size_t)b * C * n + i;
#pragma unroll 4
    for (int c = 0; c < C; c++) {
        const float* fc = f + (size_t)c * m;
        float v = __fadd_rn(__fadd_rn(__fmul_rn(w0, fc[i0]), __fmul_rn(w1, fc[i1])), __fmul_rn(w2, fc[i2]));
        o[(size_t)c * n] = v;
    }
}

// three_interpolate with the source rows staged in LDS.  In the op surface's (B, C, m) layout a wave's three gathers per channel
// touch 64 unrelated addresses of one 4 m-byte row: every load instruction is 64 separate cache accesses (906 us for
// 32 x 256 x 4096 -> 16384, 0.3 of the HBM roofline on its algorithmic bytes).  A workgroup here owns CG consecutive channels of
// one frame -- CG x m floats, up to 128 KB of LDS, loaded once with 16-byte coalesced reads -- and walks ALL n output points:
// index / weight triples are read once per point (coalesced, L2-resident across the C / CG workgroups of the frame), the 3 CG
// gathers per point are LDS reads, every output row is written coalesced.  Same arithmetic (individually rounded) as above.
#define TI_THREADS 1024
__global__ __launch_bounds__(TI_THREADS) void three_interp_lds_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                                      const float* __restrict__ w, int C, int m, int n, int CG,
                                                                      float* __restrict__ out) {
    extern __shared__ float srow[];                 // CG rows of m floats
    const int b = blockIdx.y, c0 = blockIdx.x * CG;
    const int cg = min(CG, C - c0);
    const float* f = feat + ((size_t)b * C + c0) * m;
    const int total = cg * m;                        // contiguous in memory: rows c0 .. c0 + cg - 1
    if ((m & 3) == 0 && (((uintptr_t)f) & 15) == 0) {
        for (int e = threadIdx.x * 4; e < total; e += TI_THREADS * 4) *reinterpret_cast<float4*>(srow + e) = *reinterpret_cast<const float4*>(f + e);
    } else {
        for (int e = threadIdx.x; e < total; e += TI_THREADS) srow[e] = f[e];
    }
    __syncthreads();
    float* o = out + ((size_t)b * C + c0) * n;
    for (int i = threadIdx.x; i < n; i += TI_THREADS) {
        const size_t o3 = ((size_t)b * n + i) * 3;
        const int i0 = idx[o3], i1 = idx[o3 + 1], i2 = idx[o3 + 2];
        const float w0 = w[o3], w1 = w[o3 + 1], w2 = w[o3 + 2];
#pragma unroll 4
        for (int c = 0; c < cg; c++) {
            const float* fc = srow + c * m;
            o[(size_t)c * n + i] = __fadd_rn(__fadd_rn(__fmul_rn(w0, fc[i0]), __fmul_rn(w1, fc[i1])), __fmul_rn(w2, fc[i2]));
        }
    }
}

// The same with the staged rows stored POINT-major in LDS (CGT floats per source point, contiguous): a neighbour's CGT channel
// values are one or two 16-byte LDS reads instead of CGT scalar reads at a 4 m-byte stride (24 -> 6 LDS instructions per output
// point at CGT = 8, on random addresses either way), and the next point's index / weight triple is requested before the current
// point's arithmetic.  Same individually rounded arithmetic, same bits.
template <int CGT, bool VEC>
__global__ __launch_bounds__(TI_THREADS) void three_interp_pm_kernel(const float* __restrict__ feat, const int32_t* __restrict__ idx,
                                                                     const float* __restrict__ w, int C, int m, int n,
                                                                     float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float spm[];     // m rows of CGT floats
    const int b = blockIdx.y, c0 = blockIdx.x * CGT;
    const float* f = feat + ((size_t)b * C + c0) * m;
    // staging: four source points per thread and trip, all 4 x CGT loads requested before the first LDS write (round 6: with one point
    // per trip the 128 KB came in as eight dependent round trips)
    for (int i0 = threadIdx.x; i0 < m; i0 += 4 * TI_THREADS) {
        float v[4][CGT];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int i = min(i0 + p * TI_THREADS, m - 1);
#pragma unroll
            for (int c = 0; c < CGT; c++) v[p][c] = f[(size_t)c * m + i];
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int i = i0 + p * TI_THREADS;
            if (i < m)
#pragma unroll
                for (int q = 0; q < CGT / 4; q++)
                    *reinterpret_cast<float4*>(spm + (size_t)i * CGT + 4 * q) = make_float4(v[p][4 * q], v[p][4 * q + 1], v[p][4 * q + 2], v[p][4 * q + 3]);
        }
    }
    __syncthreads();
    float* o = out + ((size_t)b * C + c0) * n;
    const int32_t* __restrict__ ib = idx + (size_t)b * n * 3;
    const float* __restrict__ wb = w + (size_t)b * n * 3;
    // PP output points per thread and trip: the workgroup owns 128 KB of LDS, so it is alone on its CU and nothing else hides the
    // latency of the index / weight triples -- with one point per trip (rounds 2-5) every trip waited a memory round trip for the
    // triple requested one trip earlier (~1.5 us x 32 trips per thread).  Now 4 triples are requested together, a trip ahead.
    // VEC (n a multiple of 4, 16-byte aligned bases): the four points are CONSECUTIVE, their 12 indices / 12 weights are three 16-byte
    // loads each and every channel's four results one 16-byte store (the output is 4/5 of the op's traffic: 32 dword stores -> 8).
    constexpr int PP = 4;
    const int stride = VEC ? 1 : TI_THREADS, step = PP * TI_THREADS;
    int id[PP][3];
    float wt[PP][3];
    auto request = [&](int base, int (&di)[PP][3], float (&dw)[PP][3]) {
        if (VEC) {
            const int4* ip = reinterpret_cast<const int4*>(ib + (size_t)base * 3);
            const float4* wp = reinterpret_cast<const float4*>(wb + (size_t)base * 3);
            const int4 a0 = ip[0], a1 = ip[1], a2 = ip[2];
            const float4 b0 = wp[0], b1 = wp[1], b2 = wp[2];
            di[0][0] = a0.x; di[0][1] = a0.y; di[0][2] = a0.z; di[1][0] = a0.w; di[1][1] = a1.x; di[1][2] = a1.y;
            di[2][0] = a1.z; di[2][1] = a1.w; di[2][2] = a2.x; di[3][0] = a2.y; di[3][1] = a2.z; di[3][2] = a2.w;
            dw[0][0] = b0.x; dw[0][1] = b0.y; dw[0][2] = b0.z; dw[1][0] = b0.w; dw[1][1] = b1.x; dw[1][2] = b1.y;
            dw[2][0] = b1.z; dw[2][1] = b1.w; dw[2][2] = b2.x; dw[3][0] = b2.y; dw[3][1] = b2.z; dw[3][2] = b2.w;
        } else {
#pragma unroll
            for (int p = 0; p < PP; p++) {
                const int i = min(base + p * stride, n - 1);
                di[p][0] = ib[i * 3]; di[p][1] = ib[i * 3 + 1]; di[p][2] = ib[i * 3 + 2];
                dw[p][0] = wb[i * 3]; dw[p][1] = wb[i * 3 + 1]; dw[p][2] = wb[i * 3 + 2];
            }
        }
    };
    int base = VEC ? PP * threadIdx.x : threadIdx.x;
    if (base >= n) return;
    request(base, id, wt);
    while (true) {
        const int nbase = base + step;
        int nid[PP][3];
        float nwt[PP][3];
        request(nbase < n ? nbase : base, nid, nwt);      // (no next trip: the request repeats this one, unused)
        float r[CGT][PP];
#pragma unroll
        for (int p = 0; p < PP; p++) {
#pragma unroll
            for (int q = 0; q < CGT / 4; q++) {
                const float4 a = *reinterpret_cast<const float4*>(spm + (size_t)id[p][0] * CGT + 4 * q);
                const float4 bq = *reinterpret_cast<const float4*>(spm + (size_t)id[p][1] * CGT + 4 * q);
                const float4 cq = *reinterpret_cast<const float4*>(spm + (size_t)id[p][2] * CGT + 4 * q);
                const float w0 = wt[p][0], w1 = wt[p][1], w2 = wt[p][2];
                r[4 * q + 0][p] = __fadd_rn(__fadd_rn(__fmul_rn(w0, a.x), __fmul_rn(w1, bq.x)), __fmul_rn(w2, cq.x));
                r[4 * q + 1][p] = __fadd_rn(__fadd_rn(__fmul_rn(w0, a.y), __fmul_rn(w1, bq.y)), __fmul_rn(w2, cq.y));
                r[4 * q + 2][p] = __fadd_rn(__fadd_rn(__fmul_rn(w0, a.z), __fmul_rn(w1, bq.z)), __fmul_rn(w2, cq.z));
                r[4 * q + 3][p] = __fadd_rn(__fadd_rn(__fmul_rn(w0, a.w), __fmul_rn(w1, bq.w)), __fmul_rn(w2, cq.w));
            }
        }
        if (VEC) {
#pragma unroll
            for (int c = 0; c < CGT; c++) *reinterpret_cast<float4*>(o + (size_t)c * n + base) = make_float4(r[c][0], r[c][1], r[c][2], r[c][3]);
        } else {
#pragma unroll
            for (int p = 0; p < PP; p++) {
                const int i = base + p * stride;
                if (i < n)
#pragma unroll
                    for (int c = 0; c < CGT; c++) o[(size_t)c * n + i] = r[c][p];
            }
        }
        if (nbase >= n) break;
        base = nbase;
#pragma unroll
        for (int p = 0; p < PP; p++)
#pragma unroll
            for (int k = 0; k < 3; k++) { id[p][k] = nid[p][k]; wt[p][k] = nwt[p][k]; }
    }
}

__global__ __launch_bounds__(G_THREADS) void three_interp_grad_kernel(const float* __restrict__ grad_out,
                                                                      const int32_t* __restrict__ idx,
                                                                      const float* __restrict__ w, int C, int n, int m,
                                                                      float* __restrict__ grad_feat) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * G_THREADS + threadIdx.x;
    if (i >= n) return;
    const size_t o3 = ((size_t)b * n + i) * 3;
    const int i0 = idx[o3], i1 = idx[o3 + 1], i2 = idx[o3 + 2];
    const float w0 = w[o3], w1 = w[o3 + 1], w2 = w[o3 + 2];
    const float* g = grad_out + (size_t)b * C * n + i;
    float* gf = grad_feat + (size_t)b * C * m;
    for (int c = 0; c < C; c++) {
        float gv = g[(size_t)c * n];
        atomicAdd(gf + (size_t)c * m + i0, gv * w0);
        atomicAdd(gf + (size_t)c * m + i1, gv * w1);
        atomicAdd(gf + (size_t)c * m + i2, gv * w2);
    }
}

// ---- backward of grouping_operation, padding-aware ----------------------------------------------------------------
// ball_query pads a group with copies of its first hit, so most of a group's nsample gradient entries scatter to the SAME
// source point: the one-atomic-per-entry kernel above serialises them (16-32 lanes of a wave on one address; it was 20 % of
// the RPN training step).  Here a lane owns one group: the entries that map to the group's first index are summed in
// registers (one atomic for all of them), the remaining distinct hits get one atomic each.  Correct for ANY index tensor
// (entries are classified by comparing with the group's first index, not by assuming the padding pattern).
template <int NS>
__global__ __launch_bounds__(G_THREADS) void group_grad_kernel(const float* __restrict__ grad_out, const int32_t* __restrict__ idx,
                                                               int C, int N, int M, float* __restrict__ grad_feat) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = blockIdx.x * 64 + lane;
    const bool live = m < M;
    int id[NS];
    const int32_t* ip = idx + ((size_t)b * M + (live ? m : 0)) * NS;
#pragma unroll
    for (int q = 0; q < NS / 4; q++) {
        const int4 v = *reinterpret_cast<const int4*>(ip + 4 * q);
        id[4 * q] = v.x; id[4 * q + 1] = v.y; id[4 * q + 2] = v.z; id[4 * q + 3] = v.w;
    }
    unsigned long long other = 0;                    // bit s: entry s does NOT map to the group's first index
#pragma unroll
    for (int s = 1; s < NS; s++) other |= (unsigned long long)(id[s] != id[0]) << s;
    if (!live) other = 0;
    // wave-uniform bound of the positions that need their own atomic (ball query: the real hits are a short prefix)
    const unsigned long long any = __ballot(other != 0) ? 1 : 0;
    int last = 0;
    if (any) {
        int mine = other ? 64 - __builtin_clzll(other) : 0;
        last = wave_max_i32(mine);
    }
    float* gf = grad_feat + (size_t)b * C * N;
    for (int c = wave; c < C; c += G_THREADS / 64) {
        const float* g = grad_out + (((size_t)b * C + c) * M + (live ? m : 0)) * NS;
        float v[NS];
#pragma unroll
        for (int q = 0; q < NS / 4; q++) {
            const float4 t = *reinterpret_cast<const float4*>(g + 4 * q);
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        float first = v[0];
#pragma unroll
        for (int s = 1; s < NS; s++) first += ((other >> s) & 1ULL) ? 0.f : v[s];
        if (live) atomicAdd(gf + (size_t)c * N + id[0], first);
#pragma unroll
        for (int s = 1; s < NS; s++) {
            if (s >= last) break;                    // wave-uniform
            if ((other >> s) & 1ULL) atomicAdd(gf + (size_t)c * N + id[s], v[s]);
        }
    }
}

// ---- backward of three_interpolate through a channels-last accumulator --------------------------------------------
// grad_feat[b, c, idx[b,i,k]] += grad_out[b, c, i] * w[b,i,k].  In the (B,C,m) layout of the op surface the 64 lanes of a wave
// (64 unknown points) hit 64 unrelated cache lines per atomic instruction -- 18 % of the RPN training step.  With the
// accumulator channels-last, (B,m,C), the lanes are 64 consecutive CHANNELS of one known point: one or two cache lines per
// instruction.  grad_out tiles (64 channels x 64 points) are transposed through LDS on the way in, a second kernel
// transposes the accumulator back into the caller's (B,C,m) tensor.
__global__ __launch_bounds__(G_THREADS) void three_interp_grad_cl_kernel(const float* __restrict__ grad_out, const int32_t* __restrict__ idx,
                                                                         const float* __restrict__ w, int C, int n, int m,
                                                                         float* __restrict__ acc_cl) {
    __shared__ float tile[64][65];
    __shared__ int sid[64][3];
    __shared__ float sw[64][3];
    const int b = blockIdx.y, i0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 192) {
        const int i = threadIdx.x / 3, k = threadIdx.x - i * 3;
        const bool ok = i0 + i < n;
        sid[i][k] = ok ? idx[((size_t)b * n + i0 + i) * 3 + k] : 0;
        sw[i][k] = ok ? w[((size_t)b * n + i0 + i) * 3 + k] : 0.f;
    }
    float* acc = acc_cl + (size_t)b * m * C;
    for (int c0 = 0; c0 < C; c0 += 64) {
        __syncthreads();
        // coalesced over the points (lanes), 16 channels per wave
        for (int cc = wave; cc < 64; cc += 4) {
            const int c = c0 + cc;
            tile[cc][lane] = (c < C && i0 + lane < n) ? grad_out[((size_t)b * C + c) * n + i0 + lane] : 0.f;
        }
        __syncthreads();
        const int c = c0 + lane;
        if (c < C) {
            for (int ii = wave; ii < 64; ii += 4) {
                if (i0 + ii >= n) break;
                const float gv = tile[lane][ii];
                atomicAdd(acc + (size_t)sid[ii][0] * C + c, gv * sw[ii][0]);
                atomicAdd(acc + (size_t)sid[ii][1] * C + c, gv * sw[ii][1]);
                atomicAdd(acc + (size_t)sid[ii][2] * C + c, gv * sw[ii][2]);
            }
        }
    }
}

// (B, m, C) -> += into (B, C, m), 64 x 64 tiles through LDS
__global__ __launch_bounds__(G_THREADS) void transpose_add_kernel(const float* __restrict__ in_cl, int C, int m, float* __restrict__ out) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, k0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int kk = wave; kk < 64; kk += 4)
        tile[kk][lane] = (k0 + kk < m && c0 + lane < C) ? in_cl[((size_t)b * m + k0 + kk) * C + c0 + lane] : 0.f;
    __syncthreads();
    for (int cc = wave; cc < 64; cc += 4)
        if (c0 + cc < C && k0 + lane < m) out[((size_t)b * C + c0 + cc) * m + k0 + lane] += tile[lane][cc];
}

__global__ __launch_bounds__(G_THREADS) void gather_rows_kernel(const float* __restrict__ in, int ld_in,
                                                                const int32_t* __restrict__ idx, int N, int M, int C,
                                                                float* __restrict__ out) {
    const int b = blockIdx.y;
    const long e = (long)blockIdx.x * G_THREADS + threadIdx.x;      // element of the (M,C) output plane
    if (e >= (long)M * C) return;
    const int m = (int)(e / C), c = (int)(e - (long)m * C);
    out[((size_t)b * M + m) * C + c] = in[((size_t)b * N + idx[(size_t)b * M + m]) * ld_in + c];
}

static int check_bcn(const char* op, const void* a, const void* b, const void* c, int B, int C, int N, long J) {
    if (B < 0 || C < 0 || N <= 0 || J < 0) return prcnn_fail(PRCNN_EINVAL, "%s: bad shape B=%d C=%d N=%d J=%ld", op, B, C, N, J);
    if (B == 0 || C == 0 || J == 0) return PRCNN_OK;      // empty problem: pointers may legitimately be null
    if (!a || !b || !c) return prcnn_fail(PRCNN_EINVAL, "%s: null pointer", op);
    return PRCNN_OK;
}

PRCNN_API int prcnn_gather(const float* feat, const int32_t* idx, int B, int C, int N, int M, float* out,
                           prcnn_stream_t stream) {
    int rc = check_bcn("prcnn_gather", feat, idx, out, B, C, N, M);
    if (rc) return rc;
    if (B == 0 || C == 0 || M == 0) return PRCNN_OK;
    hipLaunchKernelGGL(gather_kernel, dim3(prcnn_divup(M, G_THREADS), B), dim3(G_THREADS), 0, (hipStream_t)stream, feat,
                       idx, C, N, M, out);
    PRCNN_LAUNCH_CHECK("prcnn_gather");
    return PRCNN_OK;
}

PRCNN_API int prcnn_gather_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M, float* grad_feat,
                                prcnn_stream_t stream) {
    int rc = check_bcn("prcnn_gather_grad", grad_out, idx, grad_feat, B, C, N, M);
    if (rc) return rc;
    if (B == 0 || C == 0 || M == 0) return PRCNN_OK;
    hipLaunchKernelGGL(gather_grad_kernel, dim3(prcnn_divup(M, G_THREADS), B), dim3(G_THREADS), 0, (hipStream_t)stream,
                       grad_out, idx, C, N, M, grad_feat);
    PRCNN_LAUNCH_CHECK("prcnn_gather_grad");
    return PRCNN_OK;
}

PRCNN_API int prcnn_group(const float* feat, const int32_t* idx, int B, int C, int N, int M, int nsample, float* out,
                          prcnn_stream_t stream) {
    long J = (long)M * nsample;
    int rc = check_bcn("prcnn_group", feat, idx, out, B, C, N, J);
    if (rc) return rc;
    if (B == 0 || C == 0 || J == 0) return PRCNN_OK;
    if (!launch_gather_pm(feat, idx, B, C, N, J, out, (hipStream_t)stream))
        hipLaunchKernelGGL(gather_kernel, dim3(prcnn_divup(J, G_THREADS), B), dim3(G_THREADS), 0, (hipStream_t)stream, feat,
                           idx, C, N, (int)J, out);
    PRCNN_LAUNCH_CHECK("prcnn_group");
    return PRCNN_OK;
}

PRCNN_API int prcnn_group_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M, int nsample,
                               float* grad_feat, prcnn_stream_t stream) {
    long J = (long)M * nsample;
    int rc = check_bcn("prcnn_group_grad", grad_out, idx, grad_feat, B, C, N, J);
    if (rc) return rc;
    if (B == 0 || C == 0 || J == 0) return PRCNN_OK;
    const bool vec = ((uintptr_t)grad_out & 15) == 0 && ((uintptr_t)idx & 15) == 0;
    if (vec && (nsample == 16 || nsample == 32 || nsample == 64)) {          // padding-aware kernel (one lane per group)
        dim3 grid(prcnn_divup(M, 64), B);
        if (nsample == 16) hipLaunchKernelGGL(group_grad_kernel<16>, grid, dim3(G_THREADS), 0, (hipStream_t)stream, grad_out, idx, C, N, M, grad_feat);
        else if (nsample == 32) hipLaunchKernelGGL(group_grad_kernel<32>, grid, dim3(G_THREADS), 0, (hipStream_t)stream, grad_out, idx, C, N, M, grad_feat);
        else hipLaunchKernelGGL(group_grad_kernel<64>, grid, dim3(G_THREADS), 0, (hipStream_t)stream, grad_out, idx, C, N, M, grad_feat);
    } else {
        hipLaunchKernelGGL(gather_grad_kernel, dim3(prcnn_divup(J, G_THREADS), B), dim3(G_THREADS), 0, (hipStream_t)stream,
                           grad_out, idx, C, N, (int)J, grad_feat);
    }
    PRCNN_LAUNCH_CHECK("prcnn_group_grad");
    return PRCNN_OK;
}

PRCNN_API int prcnn_three_interp(const float* feat, const int32_t* idx, const float* weight, int B, int C, int m, int n,
                                 float* out, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && C >= 0 && m > 0 && n >= 0, "prcnn_three_interp: bad shape B=%d C=%d m=%d n=%d", B, C, m, n);
    if (B == 0 || C == 0 || n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(feat && idx && weight && out, "prcnn_three_interp: null pointer");
    // LDS-staged kernel when a useful number of source rows fits (<= 128 KB) and there are enough points to amortise the staging
    const bool no_lds = getenv("PRCNN_INTERP_DIRECT") != nullptr;                        // A/B switch (same bits)
    int CG = (128 * 1024 / 4) / m;
    if (CG > C) CG = C;
    if (CG > 16) CG = 16;                                                                 // (more workgroups beat longer rows)
    const char* lay = getenv("PRCNN_INTERP_LAYOUT");                                       // "rows": channel-major LDS rows (A/B switch)
    const char* cge = getenv("PRCNN_INTERP_CGT");                                          // A/B switch: channels per workgroup (same bits)
    const int CGT = (cge && atoi(cge) == 4) ? 4 : (CG >= 8 ? 8 : 4);
    if (!no_lds && !(lay && lay[0] == 'r') && CG >= 4 && C % CGT == 0 && n >= 2 * m && (long)B * (C / CGT) >= 16) {
        static PrcnnLdsLimit lim[4];
        const size_t bytes = (size_t)CGT * m * sizeof(float);
        const bool vec = (n % 4 == 0) && (((uintptr_t)idx | (uintptr_t)weight | (uintptr_t)out) & 15) == 0;
#define TI_LAUNCH(G, V, L)                                                                                                        \
        do {                                                                                                                      \
            if (!lim[L].raise((const void*)three_interp_pm_kernel<G, V>, 128 * 1024))                                             \
                return prcnn_fail(PRCNN_EHIP, "prcnn_three_interp: cannot raise the dynamic LDS limit");                          \
            hipLaunchKernelGGL((three_interp_pm_kernel<G, V>), dim3(C / G, B), dim3(TI_THREADS), bytes, (hipStream_t)stream, feat, idx, weight, \
                               C, m, n, out);                                                                                     \
        } while (0)
        if (CGT == 8) { if (vec) TI_LAUNCH(8, true, 0); else TI_LAUNCH(8, false, 1); }
        else { if (vec) TI_LAUNCH(4, true, 2); else TI_LAUNCH(4, false, 3); }
#undef TI_LAUNCH
        PRCNN_LAUNCH_CHECK("prcnn_three_interp(point-major lds)");
        return PRCNN_OK;
    }
    if (!no_lds && CG >= 4 && n >= 2 * m && (long)B * prcnn_divup(C, CG) >= 16) {
        static PrcnnLdsLimit lim;
        if (!lim.raise((const void*)three_interp_lds_kernel, 128 * 1024))
            return prcnn_fail(PRCNN_EHIP, "prcnn_three_interp: cannot raise the dynamic LDS limit");
        hipLaunchKernelGGL(three_interp_lds_kernel, dim3(prcnn_divup(C, CG), B), dim3(TI_THREADS), (size_t)CG * m * sizeof(float),
                           (hipStream_t)stream, feat, idx, weight, C, m, n, CG, out);
        PRCNN_LAUNCH_CHECK("prcnn_three_interp(lds)");
        return PRCNN_OK;
    }
    hipLaunchKernelGGL(three_interp_kernel, dim3(prcnn_divup(n, G_THREADS), B), dim3(G_THREADS), 0, (hipStream_t)stream,
                       feat, idx, weight, C, m, n, out);
    PRCNN_LAUNCH_CHECK("prcnn_three_interp");
    return PRCNN_OK;
}

PRCNN_API int prcnn_three_interp_grad(const float* grad_out, const int32_t* idx, const float* weight, int B, int C,
                                      int n, int m, float* grad_feat, float* workspace, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && C >= 0 && m > 0 && n >= 0, "prcnn_three_interp_grad: bad shape");
    if (B == 0 || C == 0 || n == 0) return PRCNN_OK;
    PRCNN_REQUIRE(grad_out && idx && weight && grad_feat, "prcnn_three_interp_grad: null pointer");
    if (workspace) {            // channels-last accumulator (B, m, C), zeroed here, transposed into grad_feat afterwards
        hipStream_t s = (hipStream_t)stream;
        if (hipMemsetAsync(workspace, 0, (size_t)B * m * C * sizeof(float), s) != hipSuccess)
            return prcnn_fail(PRCNN_EHIP, "prcnn_three_interp_grad: cannot clear the workspace");
        hipLaunchKernelGGL(three_interp_grad_cl_kernel, dim3(prcnn_divup(n, 64), B), dim3(G_THREADS), 0, s, grad_out, idx, weight, C, n, m, workspace);
        hipLaunchKernelGGL(transpose_add_kernel, dim3(prcnn_divup(m, 64), prcnn_divup(C, 64), B), dim3(G_THREADS), 0, s, workspace, C, m, grad_feat);
        PRCNN_LAUNCH_CHECK("prcnn_three_interp_grad");
        return PRCNN_OK;
    }
    hipLaunchKernelGGL(three_interp_grad_kernel, dim3(prcnn_divup(n, G_THREADS), B), dim3(G_THREADS), 0,
                       (hipStream_t)stream, grad_out, idx, weight, C, n, m, grad_feat);
    PRCNN_LAUNCH_CHECK("prcnn_three_interp_grad");
    return PRCNN_OK;
}

PRCNN_API int prcnn_gather_rows(const float* in_cl, int ld_in, const int32_t* idx, int B, int N, int M, int C,
                                float* out, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && N > 0 && M >= 0 && C > 0 && ld_in >= C, "prcnn_gather_rows: bad shape");
    if (B == 0 || M == 0) return PRCNN_OK;
    PRCNN_REQUIRE(in_cl && idx && out, "prcnn_gather_rows: null pointer");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(prcnn_divup((long)M * C, G_THREADS), B), dim3(G_THREADS), 0,
                       (hipStream_t)stream, in_cl, ld_in, idx, N, M, C, out);
    PRCNN_LAUNCH_CHECK("prcnn_gather_rows");
    return PRCNN_OK;
}
