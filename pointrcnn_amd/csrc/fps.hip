// fps.hip -- furthest point sampling for gfx950.
//
// Replaces pointnet2_cuda.furthest_point_sampling_wrapper [UPSTREAM, not in tree]; semantics per
// SURVEY Appendix A.1 / oracle prcnn_cpu_fps: start at index 0, temp=1e10, squared distances with
// individually rounded fp32 ops, arg-max ties -> LOWEST point index.
//
// Design (latency-bound op: npoint dependent iterations): one workgroup per frame, the frame's points
// and running min-distances live in VGPRs (PPT points per lane: x,y,z,t = 4*PPT registers) so an
// iteration touches no memory except one 32-byte LDS slot per wave.  Per iteration:
//   lane-local min-update + arg-max over PPT points   (VALU, ILP = PPT)
//   wave arg-max: DPP max of the float bits, then DPP min of the candidate indices (12 VALU)
//   the winning lane's coordinates are pulled out with v_readlane (uniform register index)
//   wave partials (val, idx, x, y, z) -> LDS slot[iter&1][wave]; ONE s_barrier; every wave re-reduces
//   the <=16 partials redundantly inside one DPP row.  Slots are double-buffered by iteration parity,
//   which is what makes a single barrier per iteration sufficient.
// Single-wave configurations (N <= 1024) skip LDS and the barrier entirely.
#include "common.h"

template <int PPT> struct fvec_t { typedef float type __attribute__((ext_vector_type(PPT))); };
template <> struct fvec_t<1> { typedef float type __attribute__((ext_vector_type(2))); };  // avoid 1-wide vectors

template <int BLOCK, int PPT>
__global__ __launch_bounds__(BLOCK) void fps_reg_kernel(const float* __restrict__ xyz, int N, int npoint,
                                                        int32_t* __restrict__ idx_out) {
    constexpr int NW = BLOCK / 64;
    typedef typename fvec_t<PPT>::type fvec;
    __shared__ float slot[2][NW][8];   // val(bits), idx(bits), x, y, z, pad...

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    int32_t* __restrict__ out = idx_out + (size_t)b * npoint;

    fvec px, py, pz, pt;
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        int k = i * BLOCK + tid;
        bool ok = k < N;
        px[i] = ok ? p[k * 3 + 0] : 0.f;
        py[i] = ok ? p[k * 3 + 1] : 0.f;
        pz[i] = ok ? p[k * 3 + 2] : 0.f;
        pt[i] = ok ? 1e10f : -1.0f;      // padding lanes can never win the (signed) arg-max
    }
    if (tid == 0 && npoint > 0) out[0] = 0;
    float x0 = p[0], y0 = p[1], z0 = p[2];

    for (int j = 1; j < npoint; j++) {
        float best = -2.0f;
        int bi = 0;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            float d = sqdist3(px[i], py[i], pz[i], x0, y0, z0);
            float t = d < pt[i] ? d : pt[i];
            pt[i] = t;
            if (t > best) { best = t; bi = i; }     // strict '>' keeps the lowest k of this lane
        }
        // wave arg-max with lowest-index tie break
        int vb = __float_as_int(best);              // best >= 0 or -1/-2: signed int order == float order
        int wmax = wave_max_i32(vb);
        int cand = (vb == wmax) ? (bi * BLOCK + tid) : 0x7fffffff;
        int widx = wave_min_i32(cand);
        int istar = __builtin_amdgcn_readfirstlane(widx / BLOCK);   // register slot of the winner (uniform)
        int owner = widx & 63;                                       // its lane (BLOCK % 64 == 0)
        float sx = px[istar], sy = py[istar], sz = pz[istar];
        float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sx), owner));
        float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), owner));
        float wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sz), owner));
        int gidx;
        if (NW == 1) {
            gidx = widx; x0 = wx; y0 = wy; z0 = wz;
        } else {
            float* s = slot[j & 1][wave];
            if (lane == 0) {
                s[0] = __int_as_float(wmax); s[1] = __int_as_float(widx);
                s[2] = wx; s[3] = wy; s[4] = wz;
            }
            __syncthreads();
            const float* r = slot[j & 1][lane < NW ? lane : 0];
            int v = lane < NW ? __float_as_int(r[0]) : (int)0x80000000;
            int id = __float_as_int(r[1]);
            float rx = r[2], ry = r[3], rz = r[4];
            int gmax = row0_max_i32(v);
            int c2 = (v == gmax) ? id : 0x7fffffff;
            gidx = row0_min_i32(c2);
            int wwin = (gidx % BLOCK) >> 6;          // wave that owns the winner (uniform)
            x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rx), wwin));
            y0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ry), wwin));
            z0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rz), wwin));
        }
        if (tid == 0) out[j] = gidx;
    }
}

// HBM/L2-resident fallback for N > 16384: points and temp are re-read every iteration.
__global__ __launch_bounds__(1024) void fps_mem_kernel(const float* __restrict__ xyz, int N, int npoint,
                                                       float* __restrict__ tmp, int32_t* __restrict__ idx_out) {
    constexpr int NW = 16;
    __shared__ int sval[2][NW], sidx[2][NW];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    float* __restrict__ t = tmp + (size_t)b * N;
    int32_t* __restrict__ out = idx_out + (size_t)b * npoint;
    for (int k = tid; k < N; k += 1024) t[k] = 1e10f;
    if (tid == 0 && npoint > 0) out[0] = 0;
    int old = 0;
    for (int j = 1; j < npoint; j++) {
        float x0 = p[old * 3], y0 = p[old * 3 + 1], z0 = p[old * 3 + 2];
        float best = -2.0f;
        int bk = 0x7fffffff;
        for (int k = tid; k < N; k += 1024) {
            float d = sqdist3(p[k * 3], p[k * 3 + 1], p[k * 3 + 2], x0, y0, z0);
            float v = t[k];
            v = d < v ? d : v;
            t[k] = v;
            if (v > best) { best = v; bk = k; }
        }
        int vb = __float_as_int(best);
        int wmax = wave_max_i32(vb);
        int widx = wave_min_i32(vb == wmax ? bk : 0x7fffffff);
        if (lane == 0) { sval[j & 1][wave] = wmax; sidx[j & 1][wave] = widx; }
        __syncthreads();
        int v = lane < NW ? sval[j & 1][lane] : (int)0x80000000;
        int id = lane < NW ? sidx[j & 1][lane] : 0x7fffffff;
        int gmax = row0_max_i32(v);
        old = row0_min_i32(v == gmax ? id : 0x7fffffff);
        if (tid == 0) out[j] = old;
    }
}

template <int BLOCK, int PPT>
static void launch_fps(const float* xyz, int B, int N, int npoint, int32_t* idx, hipStream_t s) {
    hipLaunchKernelGGL((fps_reg_kernel<BLOCK, PPT>), dim3(B), dim3(BLOCK), 0, s, xyz, N, npoint, idx);
}

PRCNN_API int prcnn_fps(const float* xyz, int B, int N, int npoint, float* tmp, int32_t* idx, prcnn_stream_t stream) {
    PRCNN_REQUIRE(xyz && idx, "prcnn_fps: null pointer");
    PRCNN_REQUIRE(B >= 0 && N > 0 && npoint >= 0, "prcnn_fps: bad shape B=%d N=%d npoint=%d", B, N, npoint);
    PRCNN_REQUIRE(npoint <= N, "prcnn_fps: npoint %d > N %d", npoint, N);
    if (B == 0 || npoint == 0) return PRCNN_OK;
    hipStream_t s = (hipStream_t)stream;
    if (N <= 64) launch_fps<64, 1>(xyz, B, N, npoint, idx, s);
    else if (N <= 128) launch_fps<64, 2>(xyz, B, N, npoint, idx, s);
    else if (N <= 256) launch_fps<64, 4>(xyz, B, N, npoint, idx, s);
    else if (N <= 512) launch_fps<64, 8>(xyz, B, N, npoint, idx, s);
    else if (N <= 1024) launch_fps<64, 16>(xyz, B, N, npoint, idx, s);
    else if (N <= 2048) launch_fps<256, 8>(xyz, B, N, npoint, idx, s);
    else if (N <= 4096) launch_fps<256, 16>(xyz, B, N, npoint, idx, s);
    else if (N <= 8192) launch_fps<1024, 8>(xyz, B, N, npoint, idx, s);
    else if (N <= 16384) launch_fps<1024, 16>(xyz, B, N, npoint, idx, s);
    else {
        PRCNN_REQUIRE(tmp, "prcnn_fps: N=%d > 16384 needs the (B,N) tmp buffer", N);
        hipLaunchKernelGGL(fps_mem_kernel, dim3(B), dim3(1024), 0, s, xyz, N, npoint, tmp, idx);
    }
    PRCNN_LAUNCH_CHECK("prcnn_fps");
    return PRCNN_OK;
}
