// fps.hip -- furthest point sampling for gfx950.
//
// Replaces pointnet2_cuda.furthest_point_sampling_wrapper [UPSTREAM, not in tree]; semantics per
// SURVEY Appendix A.1 / oracle prcnn_cpu_fps: start at index 0, temp=1e10, squared distances with
// individually rounded fp32 ops, arg-max ties -> LOWEST point index.
//
// Design (latency-bound op: npoint dependent iterations): one workgroup per frame, the frame's points
// and running min-distances live in VGPRs so an iteration touches no memory except one 32-byte LDS slot per
// wave.  Point ownership is LANE-MAJOR (lane t of the block owns points t*PPT .. t*PPT+PPT-1), so "lowest
// point index among the maxima" == lowest wave, then lowest lane, then lowest register slot -- which lets
// the arg-max tie-break run on the scalar unit (ballot + find-first-set) instead of a second cross-lane
// reduction.  Per iteration:
//   lane-local min-update + running max over PPT points      (10 VALU per point, ILP = PPT)
//   wave max of the float bits: 6 DPP steps (v_max_i32_dpp)   -> wmax in an SGPR
//   owner lane  = ffs(ballot(lane max == wmax))               (SALU)
//   owner slot  = first i with ballot(t[i] == wmax) bit set at the owner lane (PPT v_cmp + SALU)
//   owner's coordinates via uniform register index + v_readlane
//   wave partials (val, idx, x, y, z) -> LDS slot[iter&1][wave]; ONE s_barrier; every wave re-reduces the
//   <=16 partials redundantly (4 DPP steps + ballot).  Slots are double-buffered by iteration parity, which
//   is what makes a single barrier per iteration sufficient.
// Single-wave configurations (N <= 1024) skip LDS and the barrier entirely.
#include "lds_sort.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int PPT> struct fvec_t { typedef float type __attribute__((ext_vector_type(PPT))); };
template <> struct fvec_t<1> { typedef float type __attribute__((ext_vector_type(2))); };  // avoid 1-wide vectors

// v = max(v, v shifted by the DPP pattern) in ONE VALU op.  Lanes whose DPP source is invalid (or whose row
// is masked off) keep v.  The leading s_nop covers the VALU-write -> DPP-read hazard hipcc cannot see
// inside an asm statement.
#define FPS_DPP_MAX(v, ctrl) asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl : "+v"(v))

__device__ __forceinline__ int wave_max_i32_fused(int v) {
    FPS_DPP_MAX(v, "row_shr:1 row_mask:0xf bank_mask:0xf");
    FPS_DPP_MAX(v, "row_shr:2 row_mask:0xf bank_mask:0xf");
    FPS_DPP_MAX(v, "row_shr:4 row_mask:0xf bank_mask:0xf");
    FPS_DPP_MAX(v, "row_shr:8 row_mask:0xf bank_mask:0xf");
    FPS_DPP_MAX(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    FPS_DPP_MAX(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    asm volatile("s_nop 1");
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int row0_max_i32_fused(int v) {          // first 16 lanes only, result from lane 15
    FPS_DPP_MAX(v, "row_shr:1 row_mask:0xf bank_mask:0xf");
    FPS_DPP_MAX(v, "row_shr:2 row_mask:0xf bank_mask:0xf");
    FPS_DPP_MAX(v, "row_shr:4 row_mask:0xf bank_mask:0xf");
    FPS_DPP_MAX(v, "row_shr:8 row_mask:0xf bank_mask:0xf");
    asm volatile("s_nop 1");
    return __builtin_amdgcn_readlane(v, 15);
}

// Wave maximum of `best` AND, per lane, the bit mask of the slots holding the lane's own maximum (bit i = pt[i] == best), in one
// hand-scheduled block (round 5).  A slot costs two VALU: v_cmp_eq into an SGPR pair, then v_addc_co acc = acc + acc + hit, i.e.
// (acc << 1) | hit -- where "lowest slot holding the maximum" + "how many slots hold it" cost three per slot each way before (compare,
// select the slot number, count).  The slot steps are issued BETWEEN the six DPP steps of the maximum, whose VALU-write -> DPP-read
// hazard (two wait states) they cover: no s_nop.  The owner lane's mask is read out with one v_readlane; its lowest set bit is the slot,
// its population count says whether the maximum is unique in the lane.  An updating wave issues ~one instruction per 4-5 cycles, so this
// is latency of every sample (fps_slot_kernel<16>: -24 VALU, -6 s_nop per update).
//
// Round 6 (advisor finding): gfx940-class targets need TWO wait states between a VALU write of an SGPR / VCC and a VALU read of it
// (hipcc pads exactly that with `s_nop 1` in its own code: v_cmp_*_e64 s[..] -> v_cndmask ... s[..]); nothing pads an asm string.
// The round-5 blocks compared into VCC and consumed it in the very next instruction (no wait state), and the tail of the
// two-accumulator form left one.  All 577 GPU tests passed on that code -- the hardware evidently forwards in the cases they hit --
// but the ISA does not promise it.  Now every compare writes one of two (four in the tail of the two-accumulator form) SGPR pairs
// and its v_addc is issued at least two instructions later:   C0 C1 D A0 C2 D A1 C3 D A2 ...   (C = compare, A = add-with-carry of
// the compare two slots back, D = a DPP step of the maximum; a D is also two instructions after the previous D).  Same instruction
// count as before except one s_nop 0 before the last A of the 8- and 16-slot forms.
#define FPS_C(s, p) "v_cmp_eq_f32_e64 " s ", " p ", %[best]\n\t"
#define FPS_A(s) "v_addc_co_u32_e64 %[acc], vcc, %[acc], %[acc], " s "\n\t"
#define FPS_DPPS(ctrl) "v_max_i32_dpp %[t], %[t], %[t] " ctrl "\n\t"
#define FPS_D1 FPS_DPPS("row_shr:1 row_mask:0xf bank_mask:0xf")
#define FPS_D2 FPS_DPPS("row_shr:2 row_mask:0xf bank_mask:0xf")
#define FPS_D3 FPS_DPPS("row_shr:4 row_mask:0xf bank_mask:0xf")
#define FPS_D4 FPS_DPPS("row_shr:8 row_mask:0xf bank_mask:0xf")
#define FPS_D5 FPS_DPPS("row_bcast:15 row_mask:0xa bank_mask:0xf")
#define FPS_D6 FPS_DPPS("row_bcast:31 row_mask:0xc bank_mask:0xf")
#define FPS_SA "%[sa]"
#define FPS_SB "%[sb]"
template <int PPT> struct WaveMaxEq;
template <> struct WaveMaxEq<16> {
    template <class V> static __device__ __forceinline__ int run(const V& pt, float best, unsigned& eqbits) {
        int t; unsigned acc; unsigned long long sa, sb;
        asm volatile("v_mov_b32 %[t], %[best]\n\tv_mov_b32 %[acc], 0\n\t"
                     FPS_C(FPS_SA, "%[p15]") FPS_C(FPS_SB, "%[p14]") FPS_D1
                     FPS_A(FPS_SA) FPS_C(FPS_SA, "%[p13]") FPS_D2
                     FPS_A(FPS_SB) FPS_C(FPS_SB, "%[p12]") FPS_D3
                     FPS_A(FPS_SA) FPS_C(FPS_SA, "%[p11]") FPS_D4
                     FPS_A(FPS_SB) FPS_C(FPS_SB, "%[p10]") FPS_D5
                     FPS_A(FPS_SA) FPS_C(FPS_SA, "%[p9]") FPS_D6
                     FPS_A(FPS_SB) FPS_C(FPS_SB, "%[p8]")
                     FPS_A(FPS_SA) FPS_C(FPS_SA, "%[p7]")
                     FPS_A(FPS_SB) FPS_C(FPS_SB, "%[p6]")
                     FPS_A(FPS_SA) FPS_C(FPS_SA, "%[p5]")
                     FPS_A(FPS_SB) FPS_C(FPS_SB, "%[p4]")
                     FPS_A(FPS_SA) FPS_C(FPS_SA, "%[p3]")
                     FPS_A(FPS_SB) FPS_C(FPS_SB, "%[p2]")
                     FPS_A(FPS_SA) FPS_C(FPS_SA, "%[p1]")
                     FPS_A(FPS_SB) FPS_C(FPS_SB, "%[p0]")
                     FPS_A(FPS_SA) "s_nop 0\n\t" FPS_A(FPS_SB)
                     : [t] "=&v"(t), [acc] "=&v"(acc), [sa] "=&s"(sa), [sb] "=&s"(sb)
                     : [best] "v"(best), [p0] "v"(pt[0]), [p1] "v"(pt[1]), [p2] "v"(pt[2]), [p3] "v"(pt[3]), [p4] "v"(pt[4]), [p5] "v"(pt[5]),
                       [p6] "v"(pt[6]), [p7] "v"(pt[7]), [p8] "v"(pt[8]), [p9] "v"(pt[9]), [p10] "v"(pt[10]), [p11] "v"(pt[11]),
                       [p12] "v"(pt[12]), [p13] "v"(pt[13]), [p14] "v"(pt[14]), [p15] "v"(pt[15])
                     : "vcc");
        eqbits = acc;
        return t;                                         // lane 63 holds the wave maximum (twenty VALU after the last DPP step)
    }
};
template <> struct WaveMaxEq<8> {
    template <class V> static __device__ __forceinline__ int run(const V& pt, float best, unsigned& eqbits) {
        int t; unsigned acc; unsigned long long sa, sb;
        asm volatile("v_mov_b32 %[t], %[best]\n\tv_mov_b32 %[acc], 0\n\t"
                     FPS_C(FPS_SA, "%[p7]") FPS_C(FPS_SB, "%[p6]") FPS_D1
                     FPS_A(FPS_SA) FPS_C(FPS_SA, "%[p5]") FPS_D2
                     FPS_A(FPS_SB) FPS_C(FPS_SB, "%[p4]") FPS_D3
                     FPS_A(FPS_SA) FPS_C(FPS_SA, "%[p3]") FPS_D4
                     FPS_A(FPS_SB) FPS_C(FPS_SB, "%[p2]") FPS_D5
                     FPS_A(FPS_SA) FPS_C(FPS_SA, "%[p1]") FPS_D6
                     FPS_A(FPS_SB) FPS_C(FPS_SB, "%[p0]")
                     FPS_A(FPS_SA) "s_nop 0\n\t" FPS_A(FPS_SB)
                     : [t] "=&v"(t), [acc] "=&v"(acc), [sa] "=&s"(sa), [sb] "=&s"(sb)
                     : [best] "v"(best), [p0] "v"(pt[0]), [p1] "v"(pt[1]), [p2] "v"(pt[2]), [p3] "v"(pt[3]), [p4] "v"(pt[4]), [p5] "v"(pt[5]),
                       [p6] "v"(pt[6]), [p7] "v"(pt[7])
                     : "vcc");
        eqbits = acc;
        return t;
    }
};
template <> struct WaveMaxEq<4> {
    template <class V> static __device__ __forceinline__ int run(const V& pt, float best, unsigned& eqbits) {
        int t; unsigned acc; unsigned long long sa, sb;
        asm volatile("v_mov_b32 %[t], %[best]\n\tv_mov_b32 %[acc], 0\n\t"
                     FPS_C(FPS_SA, "%[p3]") FPS_C(FPS_SB, "%[p2]") FPS_D1
                     FPS_A(FPS_SA) FPS_C(FPS_SA, "%[p1]") FPS_D2
                     FPS_A(FPS_SB) FPS_C(FPS_SB, "%[p0]") FPS_D3
                     FPS_A(FPS_SA) FPS_A(FPS_SB) FPS_D4
                     "s_nop 1\n\t" FPS_D5
                     "s_nop 1\n\t" FPS_D6
                     "s_nop 1\n\t"
                     : [t] "=&v"(t), [acc] "=&v"(acc), [sa] "=&s"(sa), [sb] "=&s"(sb)
                     : [best] "v"(best), [p0] "v"(pt[0]), [p1] "v"(pt[1]), [p2] "v"(pt[2]), [p3] "v"(pt[3])
                     : "vcc");
        eqbits = acc;
        return t;
    }
};

// WaveMaxEq<16> with TWO accumulators (round 5): a v_addc then reads a compare issued two to three instructions earlier and an
// accumulator written two instructions earlier -- the one-accumulator form is a chain of instructions each waiting for its
// predecessor's result (measured ~6 cycles per instruction on a lone wave against 4 for independent ones).  a0 collects slots 15..8,
// a1 slots 7..0.  Every compare is two or more instructions ahead of the v_addc that reads its SGPR pair (see above): in the groups
// around a DPP step by construction, in the tail (no DPP step left) by comparing the last four slots into four pairs first.
#define FPS_EQ2(pa, pb) "v_cmp_eq_f32_e64 %[sa], " pa ", %[best]\n\tv_cmp_eq_f32_e64 %[sb], " pb ", %[best]\n\t"
#define FPS_AC2 "v_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[sa]\n\tv_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[sb]\n\t"
template <class V> __device__ __forceinline__ int wave_max_eq2_16(const V& pt, float best, unsigned& eqbits) {
    int t; unsigned a0, a1; unsigned long long sa, sb, sc, sd;
    asm volatile("v_mov_b32 %[t], %[best]\n\tv_mov_b32 %[a0], 0\n\tv_mov_b32 %[a1], 0\n\t"
                 FPS_EQ2("%[p15]", "%[p7]") FPS_D1 FPS_AC2
                 FPS_EQ2("%[p14]", "%[p6]") FPS_D2 FPS_AC2
                 FPS_EQ2("%[p13]", "%[p5]") FPS_D3 FPS_AC2
                 FPS_EQ2("%[p12]", "%[p4]") FPS_D4 FPS_AC2
                 FPS_EQ2("%[p11]", "%[p3]") FPS_D5 FPS_AC2
                 FPS_EQ2("%[p10]", "%[p2]") FPS_D6 FPS_AC2
                 FPS_EQ2("%[p9]", "%[p1]")
                 "v_cmp_eq_f32_e64 %[sc], %[p8], %[best]\n\tv_cmp_eq_f32_e64 %[sd], %[p0], %[best]\n\t"
                 FPS_AC2
                 "v_addc_co_u32_e64 %[a0], vcc, %[a0], %[a0], %[sc]\n\tv_addc_co_u32_e64 %[a1], vcc, %[a1], %[a1], %[sd]\n\t"
                 : [t] "=&v"(t), [a0] "=&v"(a0), [a1] "=&v"(a1), [sa] "=&s"(sa), [sb] "=&s"(sb), [sc] "=&s"(sc), [sd] "=&s"(sd)
                 : [best] "v"(best), [p0] "v"(pt[0]), [p1] "v"(pt[1]), [p2] "v"(pt[2]), [p3] "v"(pt[3]), [p4] "v"(pt[4]), [p5] "v"(pt[5]),
                   [p6] "v"(pt[6]), [p7] "v"(pt[7]), [p8] "v"(pt[8]), [p9] "v"(pt[9]), [p10] "v"(pt[10]), [p11] "v"(pt[11]),
                   [p12] "v"(pt[12]), [p13] "v"(pt[13]), [p14] "v"(pt[14]), [p15] "v"(pt[15])
                 : "vcc");
    eqbits = (a0 << 8) | a1;
    return t;
}

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
        int k = tid * PPT + i;            // lane-major ownership
        bool ok = k < N;
        px[i] = ok ? p[k * 3 + 0] : 0.f;
        py[i] = ok ? p[k * 3 + 1] : 0.f;
        pz[i] = ok ? p[k * 3 + 2] : 0.f;
        pt[i] = ok ? 1e10f : -1.0f;      // padding can never win the (signed) arg-max
    }
    if (tid == 0 && npoint > 0) out[0] = 0;
    float x0 = p[0], y0 = p[1], z0 = p[2];

    for (int j = 1; j < npoint; j++) {
        float best = -2.0f;
        if (PPT >= 2) {
            // packed fp32: plain v_add/v_mul_f32 issue at ~3.6 cycles per wave64 op on CDNA4, v_pk_* at the full
            // rate.  Same individually rounded operations per component (no FMA), two points per instruction.
            const f32x2 qx = {x0, x0}, qy = {y0, y0}, qz = {z0, z0};
#pragma unroll
            for (int i = 0; i + 1 < PPT; i += 2) {
                f32x2 dx = (f32x2){px[i], px[i + 1]} - qx;
                f32x2 dy = (f32x2){py[i], py[i + 1]} - qy;
                f32x2 dz = (f32x2){pz[i], pz[i + 1]} - qz;
                f32x2 d = (dx * dx + dy * dy) + dz * dz;
                float t0 = __builtin_fminf(pt[i], d.x), t1 = __builtin_fminf(pt[i + 1], d.y);
                pt[i] = t0; pt[i + 1] = t1;
                best = __builtin_fmaxf(best, __builtin_fmaxf(t0, t1));
            }
        } else {
            float d = sqdist3(px[0], py[0], pz[0], x0, y0, z0);
            float t = __builtin_fminf(pt[0], d);
            pt[0] = t;
            best = t;
        }
        // wave arg-max, ties -> lowest point index == lowest lane, then lowest slot
        const int wmax = wave_max_i32_fused(__float_as_int(best));       // >= 0, or -1/-2: int order == float order
        const float wmaxf = __int_as_float(wmax);
        const int owner = __builtin_ctzll(__ballot(best == wmaxf));      // lowest lane holding the maximum
        // lowest slot holding the maximum: every lane finds the lowest slot holding ITS maximum (2 VALU per slot), the owner's
        // answer is read out -- a ballot + scalar shift / test / select per slot cost 5 instructions each, and a lone
        // latency-bound wave pays ~4 cycles per instruction whatever unit runs it (tools/fps_timing.py)
        int myslot = PPT - 1;
#pragma unroll
        for (int i = PPT - 2; i >= 0; i--) myslot = (pt[i] == best) ? i : myslot;
        const int istar = __builtin_amdgcn_readlane(myslot, owner);
        const int widx = (wave * 64 + owner) * PPT + istar;
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
                s[0] = wmaxf; s[1] = __int_as_float(widx);
                s[2] = wx; s[3] = wy; s[4] = wz;
            }
            __syncthreads();
            const float* r = slot[j & 1][lane < NW ? lane : 0];
            int v = lane < NW ? __float_as_int(r[0]) : (int)0x80000000;
            int id = __float_as_int(r[1]);
            float rx = r[2], ry = r[3], rz = r[4];
            const int gmax = row0_max_i32_fused(v);
            const int wwin = __builtin_ctzll(__ballot(v == gmax));       // lowest wave holding the global maximum
            gidx = __builtin_amdgcn_readlane(id, wwin);
            x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rx), wwin));
            y0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ry), wwin));
            z0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rz), wwin));
        }
        if (tid == 0) out[j] = gidx;
    }
}

// =====================================================================================================
// Spatially pruned FPS (exact).  A pre-pass sorts the frame's points by Morton code, so that the 1024 (or 256) points
// a wave owns form a compact spatial group; the wave keeps its group's bounding box.  For a new sample q the value
//     L = ((gx*gx + gy*gy) + gz*gz),   g_a = max(lo_a - q_a, q_a - hi_a, 0)
// computed with the same individually rounded fp32 operations as the distance itself satisfies L <= d(p, q) for every
// point p of the group (fp32 subtraction, multiplication and addition are monotone, and |p_a - q_a| >= g_a exactly).
// If L >= the group's current maximum of min-distances, then d >= t for all its points and NO min-distance changes:
// the wave skips its update and re-publishes its cached candidate.  Late in the sampling most waves skip (76 % of the
// wave-updates on the benchmark clouds), and an updating wave has its SIMD's VALU to itself.
// Results are bit-identical to the un-pruned kernel: ties are resolved on ORIGINAL point indices.
// Measured (MI355X, bs32 x 16384 -> 4096): 5.9 -> 5.2 ms standalone.  While the neighbour searches were full scans the
// FPS chain hid behind them and this kernel was ~3 % slower with 3 batches in flight (the lone updating wave per SIMD
// gets a smaller share of issue slots under contention); with the searches on the grid (grid.hip) the FPS chain is the
// longest dependency of a batch and the shorter chain is worth +8 % RPN throughput, so the host layer uses it by
// default (PRCNN_FPS_PRUNED=0 selects the plain kernel).
// Selected by passing the (B,N) `tmp` buffer with 2048 < N <= 16384.
// =====================================================================================================
// finite stand-in for infinity (this file is built with -ffinite-math-only); FPS_BIG^2 * 3 still fits fp32
#define FPS_BIG 1.0e18f
// FPS_V: build-time A/B switches of two round-5 changes to the sample loop (bit 1: early priority for the winner's wave and its Morton
// neighbours, bit 2: two-accumulator slot masks + tree maximum + one-compare uniqueness test); default: both on
#ifndef FPS_V
#define FPS_V 6
#endif
#ifndef FPS_PAIR_ASM
#define FPS_PAIR_ASM 1          // 0: the compiler's order of the pair update (A/B switch of the build)
#endif

__device__ __forceinline__ unsigned morton_spread6(unsigned v) {       // 6 bits -> every third bit
    v &= 0x3fu;
    v = (v | (v << 8)) & 0x0000300fu;
    v = (v | (v << 4)) & 0x000030c3u;
    v = (v | (v << 2)) & 0x00009249u;
    return v;
}

// One workgroup per frame: perm[s] = original index of the s-th point in Morton order (stable on the index).
// 32-bit keys (18-bit Morton code << 14 | index; N <= 16 384) sorted by the shared LDS bitonic sort (lds_sort.h: strides <= 8 in
// registers); NP = power of two >= N (<= 16384 -> 68 KB of LDS); blockDim = max(64, NP / 16) threads.
// Round 6: the keys were 64 bits (30-bit code, 32-bit index) -- the sort is bound by LDS bandwidth (55 passes over the keys), so half
// the key is half the time (95 -> ~50 us per 16 384-point frame set) and half the LDS (136 -> 68 KB, which kept every other
// workgroup off the CU).  The order only decides how compact a slot's 64 points are, never a result (ties go to the ORIGINAL index
// inside the FPS kernels): 64 cells per axis leave ~1 point per occupied cell at 16 384 points.
__global__ __launch_bounds__(1024) void fps_sort_kernel(const float* __restrict__ xyz, int N, int NP, int32_t* __restrict__ perm) {
    extern __shared__ unsigned keys[];
    __shared__ float red[6][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    float lo[3] = {FPS_BIG, FPS_BIG, FPS_BIG}, hi[3] = {-FPS_BIG, -FPS_BIG, -FPS_BIG};
    for (int k = tid; k < N; k += blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; a++) { float v = p[k * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
        if (lane == 0) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    }
    __syncthreads();
    float scale[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float l = red[a][0], h = red[3 + a][0];
        for (int w = 1; w < nw; w++) { l = fminf(l, red[a][w]); h = fmaxf(h, red[3 + a][w]); }
        lo[a] = l;
        scale[a] = (h > l) ? 63.0f / (h - l) : 0.f;
    }
    unsigned v[16];
    if (tid * 16 < NP) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int k = tid * 16 + e;
            unsigned key = ~0u;                             // padding sorts to the end
            if (k < N) {
                unsigned qx = (unsigned)((p[k * 3 + 0] - lo[0]) * scale[0]);
                unsigned qy = (unsigned)((p[k * 3 + 1] - lo[1]) * scale[1]);
                unsigned qz = (unsigned)((p[k * 3 + 2] - lo[2]) * scale[2]);
                unsigned m = (morton_spread6(qx) << 2) | (morton_spread6(qz) << 1) | morton_spread6(qy);
                key = (m << 14) | (unsigned)k;
            }
            v[e] = key;
        }
    }
    block_sort16(v, keys, NP, tid);
    if (tid * 16 < NP) {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int k = tid * 16 + e;
            if (k < N) perm[(size_t)b * N + k] = (int32_t)(v[e] & 0x3fffu);
        }
    }
}

#define FPS_DPP_MIN(v, ctrl) asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 " ctrl : "+v"(v))
#define FPS_DPP_FMIN(v, ctrl) asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " ctrl : "+v"(v))
#define FPS_DPP_FMAX(v, ctrl) asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl : "+v"(v))
#define FPS_WAVE_REDUCE(MACRO, v)                                   \
    MACRO(v, "row_shr:1 row_mask:0xf bank_mask:0xf");               \
    MACRO(v, "row_shr:2 row_mask:0xf bank_mask:0xf");               \
    MACRO(v, "row_shr:4 row_mask:0xf bank_mask:0xf");               \
    MACRO(v, "row_shr:8 row_mask:0xf bank_mask:0xf");               \
    MACRO(v, "row_bcast:15 row_mask:0xa bank_mask:0xf");            \
    MACRO(v, "row_bcast:31 row_mask:0xc bank_mask:0xf");            \
    asm volatile("s_nop 1");

__device__ __forceinline__ int wave_min_i32_fused(int v) { FPS_WAVE_REDUCE(FPS_DPP_MIN, v) return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ float wave_min_f32_fused(float v) {
    FPS_WAVE_REDUCE(FPS_DPP_FMIN, v) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f32_fused(float v) {
    FPS_WAVE_REDUCE(FPS_DPP_FMAX, v) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int row0_min_i32_fused(int v) {
    FPS_DPP_MIN(v, "row_shr:1 row_mask:0xf bank_mask:0xf");
    FPS_DPP_MIN(v, "row_shr:2 row_mask:0xf bank_mask:0xf");
    FPS_DPP_MIN(v, "row_shr:4 row_mask:0xf bank_mask:0xf");
    FPS_DPP_MIN(v, "row_shr:8 row_mask:0xf bank_mask:0xf");
    asm volatile("s_nop 1");
    return __builtin_amdgcn_readlane(v, 15);
}

// ---- the per-sample exchange of the workgroup kernels (fps_pruned_kernel, fps_slot_kernel) ----
// Exchange table, three rotating buffers of 64 words: words 0-15 / 16-31 / 32-47 = x / y / z of the 16 waves' candidates, words
// 48-49 = the 64-bit cell the candidates are folded into with ONE ds_max_u64 per wave: key = [value, order-preserving | 2^28-1 -
// original index | wave], the largest key is the largest value, ties -> lowest original index (reading all 16 candidates and
// reducing them twice by DPP was ~60 instructions per wave and sample).  The buffer of sample j is j % 3; wave 0 clears the next
// one's cell while nobody can still be reading it (its readers passed the previous barrier).  Round 5: ONE ds_read_b32 per lane
// after the barrier fetches the whole buffer, so cell and coordinates arrive together (rounds 3-4 read a 16 x 4 slot table and the
// cell with two loads that were meant to fly together, but the register allocator reused the cell's upper half as the second
// address: two dependent LDS round trips on the critical path of every sample; found by reading the ISA).
#define FPS_XT_DECL __shared__ __attribute__((aligned(256))) unsigned xt[3][64]
// The candidate a wave publishes is kept in the form the publish needs -- coordinates and the 64-bit key in VGPRs (fps_cand_t,
// rebuilt only when the wave updates) -- so that a wave that did not update re-publishes with five instructions (address, two
// coordinate stores, the atomic, wave 0's clear) instead of rebuilding key and operands from scalars every sample (~17).
struct fps_cand_t { float x, y, z; unsigned long long key; };
__device__ __forceinline__ void fps_cand_set(fps_cand_t& c, int wave, float cx, float cy, float cz, float cval, int corig) {
    const unsigned hi = (unsigned)__float_as_int(cval) ^ 0x80000000u;
    const unsigned lo = ((0xFFFFFFFu - (unsigned)min(corig, 0xFFFFFFF)) << 4) | (unsigned)wave;
    c.x = cx; c.y = cy; c.z = cz; c.key = ((unsigned long long)hi << 32) | lo;
    asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.key));          // uniform values, but they stay in vector registers
}
__device__ __forceinline__ void fps_xt_publish(unsigned (*xt)[64], int cb, int wave, const fps_cand_t& c) {
    unsigned* t = xt[cb];
    t[wave] = __float_as_uint(c.x); t[16 + wave] = __float_as_uint(c.y); t[32 + wave] = __float_as_uint(c.z);
    // (one lane, one instruction: atomicMax() would be wrapped in the compiler's wave-aggregation loop, ~25 more instructions per
    //  wave and sample on the critical path)
    asm volatile("ds_max_u64 %0, %1 offset:192" : : "v"((unsigned)(size_t)&t[0]), "v"(c.key) : "memory");
    if (wave == 0) *(unsigned long long*)&xt[cb == 2 ? 0 : cb + 1][48] = 0ULL;
}
// after the barrier: lane L reads word L of the sample's buffer; the cell's low word (original index | wave) is lane 48's, the
// winner's coordinates are lanes w, 16 + w, 32 + w's.  Returns the winner's original index.
__device__ __forceinline__ int fps_xt_collect(unsigned (*xt)[64], int cb, int lane, float& x0, float& y0, float& z0, int& wl) {
    const int tv = (int)xt[cb][lane];
    const unsigned klo = (unsigned)__builtin_amdgcn_readlane(tv, 48);
    wl = (int)(klo & 15u);
    x0 = __int_as_float(__builtin_amdgcn_readlane(tv, wl));
    y0 = __int_as_float(__builtin_amdgcn_readlane(tv, wl | 16));
    z0 = __int_as_float(__builtin_amdgcn_readlane(tv, wl | 32));
    return (int)(0xFFFFFFFu - (klo >> 4));
}

#ifdef PRCNN_FPS_TIMING            // dev build (tools/fps_timing.py): per-wave cycle sums of the loop's phases, frame 0
__device__ unsigned long long prcnn_fps_dbg[16 * 8 + 16 + 16 * 16];      // pruned kernel: [0, 144); slot kernel: 16 words per wave from 144
PRCNN_API int prcnn_fps_timing_read(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(prcnn_fps_dbg), sizeof(prcnn_fps_dbg)) == hipSuccess ? 0 : -1;
}
#define FPS_T(...) __VA_ARGS__
// the slot kernel takes time stamps on ONE wave per build (PRCNN_FPS_TIMING_WAVE; -1 = every wave): sixteen waves reading the clock at
// the same points serialise on the scalar cache and measure each other
#ifndef PRCNN_FPS_TIMING_WAVE
#define PRCNN_FPS_TIMING_WAVE -1
#endif
#define FPS_NOW(w) ((PRCNN_FPS_TIMING_WAVE < 0 || (w) == PRCNN_FPS_TIMING_WAVE) ? __builtin_readcyclecounter() : 0ULL)
#else
#define FPS_T(...)
#endif

template <int PPT>
__global__ __launch_bounds__(1024) void fps_pruned_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ perm,
                                                          int N, int npoint, int32_t* __restrict__ idx_out) {
    constexpr int BLOCK = 1024, NW = 16;
    typedef typename fvec_t<PPT>::type fvec;
    FPS_XT_DECL;
    // original indices of the points a lane holds: only the winner's is ever needed, so they live in LDS (slot-major:
    // s_po[i * 1024 + tid]) instead of PPT more VGPRs per lane -- at 96 VGPRs the four FPS waves of a SIMD left 128
    // registers, too few for ANY of the MLP kernels (160-216), i.e. a CU hosting an FPS workgroup was lost to them
    extern __shared__ int s_po[];
    __builtin_amdgcn_s_setprio(2);     // the serial chain every batch waits for: its few instructions go first (3 while a wave updates)

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);          // the wave's index in a scalar register (loop-invariant)
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    const int32_t* __restrict__ pm = perm + (size_t)b * N;
    int32_t* __restrict__ out = idx_out + (size_t)b * npoint;

    fvec px, py, pz, pt;
    float lox = FPS_BIG, loy = FPS_BIG, loz = FPS_BIG, hix = -FPS_BIG, hiy = -FPS_BIG, hiz = -FPS_BIG;
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        int s = tid * PPT + i;                      // position in Morton order
        bool ok = s < N;
        int o = ok ? pm[s] : 0x7fffffff;
        s_po[i * BLOCK + tid] = o;
        px[i] = ok ? p[o * 3 + 0] : 0.f;
        py[i] = ok ? p[o * 3 + 1] : 0.f;
        pz[i] = ok ? p[o * 3 + 2] : 0.f;
        pt[i] = ok ? 1e10f : -1.0f;
        if (ok) {
            lox = fminf(lox, px[i]); hix = fmaxf(hix, px[i]);
            loy = fminf(loy, py[i]); hiy = fmaxf(hiy, py[i]);
            loz = fminf(loz, pz[i]); hiz = fmaxf(hiz, pz[i]);
        }
    }
    // the wave's bounding box (uniform); an all-padding wave keeps the empty box (+-1e18: L ~ 3e36 stays finite)
    lox = wave_min_f32_fused(lox); loy = wave_min_f32_fused(loy); loz = wave_min_f32_fused(loz);
    hix = wave_max_f32_fused(hix); hiy = wave_max_f32_fused(hiy); hiz = wave_max_f32_fused(hiz);

    if (tid == 0 && npoint > 0) out[0] = 0;
    if (tid < 192) (&xt[0][0])[tid] = 0u;
    __syncthreads();
    float x0 = p[0], y0 = p[1], z0 = p[2];
    // cached candidate of this wave (uniform): value, original index, coordinates
    float cval = 1e10f; int corig = 0x7fffffff; float cx = 0.f, cy = 0.f, cz = 0.f;
    fps_cand_t cand; fps_cand_set(cand, wave, cx, cy, cz, cval, corig);
    bool first = true;
    int wprev = -1;                            // the wave whose candidate is the current sample

    FPS_T(unsigned long long t_upd = 0, t_wait = 0, t_red = 0, n_upd = 0, t_u1 = 0, t_u2 = 0, t_u3 = 0, t_u4 = 0; unsigned long long t0 = __builtin_readcyclecounter();)
    FPS_T(const unsigned long long t_begin = t0;)
    int cb = 1;                                // exchange cell of sample j: j % 3
    for (int j = 1; j < npoint; j++) {
        // lower bound of the distance from the new sample to anything in this wave's box
        float gx = fmaxf(fmaxf(lox - x0, x0 - hix), 0.f);
        float gy = fmaxf(fmaxf(loy - y0, y0 - hiy), 0.f);
        float gz = fmaxf(fmaxf(loz - z0, z0 - hiz), 0.f);
        float L = __fadd_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)), __fmul_rn(gz, gz));
        const bool update = first || __builtin_amdgcn_readfirstlane(__float_as_int(L)) < __float_as_int(cval) ;
        // (L and cval are >= 0 or cval = -1 for an all-padding wave: the int compare is the float compare; L < cval
        //  means some point MAY change.  L >= cval => provably nothing changes.)
        if (update) {
            // the sample waits for the updating wave(s): ahead of the three waves that share the SIMD and are still
            // working through their own bound test / exchange read (oldest-first arbitration otherwise: tools/fps_timing.py
            // shows the fourth wave of a SIMD taking 2-3x as long for the same instructions)
            __builtin_amdgcn_s_setprio(3);
            FPS_T(unsigned long long u0 = __builtin_readcyclecounter();)
            float best = -2.0f;
            if (PPT >= 2) {
                const f32x2 qx = {x0, x0}, qy = {y0, y0}, qz = {z0, z0};
#pragma unroll
                for (int i = 0; i + 1 < PPT; i += 2) {
                    f32x2 dx = (f32x2){px[i], px[i + 1]} - qx;
                    f32x2 dy = (f32x2){py[i], py[i + 1]} - qy;
                    f32x2 dz = (f32x2){pz[i], pz[i + 1]} - qz;
                    f32x2 d = (dx * dx + dy * dy) + dz * dz;
                    float t0 = __builtin_fminf(pt[i], d.x), t1 = __builtin_fminf(pt[i + 1], d.y);
                    pt[i] = t0; pt[i + 1] = t1;
                    best = __builtin_fmaxf(best, __builtin_fmaxf(t0, t1));
                }
            } else {
                float d = sqdist3(px[0], py[0], pz[0], x0, y0, z0);
                float t = __builtin_fminf(pt[0], d);
                pt[0] = t; best = t;
            }
            FPS_T(unsigned long long u1 = __builtin_readcyclecounter(); t_u1 += u1 - u0;)
            // Round 6: a candidate whose own distance did not shrink is still the wave's maximum (see fps_slot_kernel): republish it as it is
            bool shrunk = wave_u == wprev || first;
            if (!shrunk) {
                const float ex = __fsub_rn(cx, x0), ey = __fsub_rn(cy, y0), ez = __fsub_rn(cz, z0);
                shrunk = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)) < cval;
            }
            if (shrunk) {
            unsigned eqbits;
            const int wvec = WaveMaxEq<PPT>::run(pt, best, eqbits);          // lane 63 = the wave maximum
            // every lane fetches the original index of ITS OWN candidate (lowest slot holding the lane's maximum) as soon as the slot
            // masks exist; the owner's word is read out below, after the scalar search and the coordinate selects, so the LDS round
            // trip is hidden (rounds 3-4 read s_po at the owner's address inside the fast path, after the search: an exposed round
            // trip per update -- a load whose only use sits in a branch is sunk into it; the sched_barriers keep the order written)
            const int wmax = __builtin_amdgcn_readlane(wvec, 63);          // first: the load's address arithmetic fills its wait states
            const int myorig = s_po[__builtin_ctz(eqbits) * BLOCK + tid];
            __builtin_amdgcn_sched_barrier(0);
            const float wmaxf = __int_as_float(wmax);
            FPS_T(unsigned long long u2 = __builtin_readcyclecounter(); t_u2 += u2 - u1;)
            // candidates = (lane, slot) with t == wmax; the one with the LOWEST ORIGINAL index wins.  Fast path (a
            // unique maximum, the overwhelmingly common case).  An updating wave is usually ALONE on its SIMD and issues one
            // instruction every ~4-5 cycles, so its instruction count is the latency of the whole sample (measured with
            // tools/fps_timing.py: ~1450 cycles per update, of which a per-slot ballot + scalar select chain took ~600).
            // Here every lane finds the lowest slot holding ITS maximum and how many slots do (3 VALU per slot, no scalar
            // chain); one ballot finds the lanes holding the wave maximum; unique lane with a unique slot = fast path.
            const unsigned long long anym = __ballot(best == wmaxf);
            const int owner0 = __builtin_ctzll(anym);
            const unsigned ownbits = (unsigned)__builtin_amdgcn_readlane((int)eqbits, owner0);       // the owner lane's slots holding the maximum
            const int total = (__popcll(anym) == 1 && __popc(ownbits) == 1) ? 1 : 2;
            int istar = __builtin_ctz(ownbits);
            // the fast path's selects, taken before the branch (wasted on the rare tie path) so that they too run under the load
            const float fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[istar]), owner0));
            const float fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[istar]), owner0));
            const float fz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[istar]), owner0));
            __builtin_amdgcn_sched_barrier(0);
            const int corig_fast = __builtin_amdgcn_readlane(myorig, owner0);
            FPS_T(unsigned long long u3 = __builtin_readcyclecounter(); t_u3 += u3 - u2;)
            if (total == 1) {
                corig = corig_fast; cx = fx; cy = fy; cz = fz;
            } else {                                  // exact ties (duplicated points, lattices): rare, any cost is fine
                int bo = 0x7fffffff; float bx = 0.f, by = 0.f, bz = 0.f;
                if (best == wmaxf) {
#pragma unroll
                    for (int i = PPT - 1; i >= 0; i--) {
                        const int oi = s_po[i * BLOCK + tid];
                        if (pt[i] == wmaxf && oi <= bo) { bo = oi; bx = px[i]; by = py[i]; bz = pz[i]; }
                    }
                }
                corig = wave_min_i32_fused(bo);
                const int owner = __builtin_ctzll(__ballot(bo == corig));
                cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx), owner));
                cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(by), owner));
                cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bz), owner));
            }
            cval = wmaxf;
            first = false;
            fps_cand_set(cand, wave, cx, cy, cz, cval, corig);
            FPS_T(t_u4 += __builtin_readcyclecounter() - u3;)
            }
            __builtin_amdgcn_s_setprio(2);
        }
#if FPS_V & 2
        else __builtin_amdgcn_s_setprio(2);
#endif
        FPS_T(unsigned long long t1 = __builtin_readcyclecounter(); t_upd += t1 - t0; n_upd += update ? 1 : 0;)
        // Exchange (fps_xt_publish / fps_xt_collect above): candidates folded into the sample's cell, one barrier, one LDS read
        if (lane == 0) fps_xt_publish(xt, cb, wave, cand);
        __syncthreads();
        FPS_T(unsigned long long t2 = __builtin_readcyclecounter(); t_wait += t2 - t1;)
        int wwin;
        const int gorig = fps_xt_collect(xt, cb, lane, x0, y0, z0, wwin);
#if FPS_V & 2
        if ((unsigned)(wave_u - wwin + 1) <= 2u) __builtin_amdgcn_s_setprio(3);          // see fps_slot_kernel
#endif
        wprev = wwin;
        cb = cb == 2 ? 0 : cb + 1;
        if (tid == 0) out[j] = gorig;
        FPS_T(t0 = __builtin_readcyclecounter(); t_red += t0 - t2;)
    }
    FPS_T(if (b == 0 && lane == 0) { unsigned long long* d = prcnn_fps_dbg + wave * 8; d[0] = t_upd; d[1] = t_wait; d[2] = t_red; d[3] = n_upd;
                                     d[4] = __builtin_readcyclecounter() - t_begin; d[5] = t_u1; d[6] = t_u2; d[7] = t_u3; prcnn_fps_dbg[128 + wave] = t_u4; })
}

// =====================================================================================================
// Two-level pruning (fps_slot_kernel): fps_pruned_kernel with SLOT-major point ownership -- slot i of a wave is 64 consecutive
// Morton points, a pair of slots a 128-point cluster with its own box -- so that the update of a wave the sample reaches into
// touches only the pairs it can reach (a uniform branch per pair) instead of all 16 slots.  The pairs are tested against the
// wave's largest distance, which bounds every point's: no per-pair maxima to maintain (round 2's version kept them exact with a
// DPP reduction per slot and lost more than it skipped).  Same arithmetic and tie rule: bit-identical sample sets.
// =====================================================================================================
template <int PPT, int NW>
__global__ __launch_bounds__(NW * 64) void fps_slot_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ perm,
                                                          int N, int npoint, int32_t* __restrict__ idx_out) {
    constexpr int BLOCK = NW * 64;          // NW waves x 64 lanes x PPT points (16 x 16 for 16 384 points; the 4 096-point level on 8 x 8: 716, on 4 x 16: 748 vs 653 us for 16 waves x 4 points -- not used)
    typedef typename fvec_t<PPT>::type fvec;
    FPS_XT_DECL;
    // original indices of the points a lane holds: only the winner's is ever needed, so they live in LDS (slot-major:
    // s_po[i * 1024 + tid]) instead of PPT more VGPRs per lane -- at 96 VGPRs the four FPS waves of a SIMD left 128
    // registers, too few for ANY of the MLP kernels (160-216), i.e. a CU hosting an FPS workgroup was lost to them
    extern __shared__ int s_po[];
    __builtin_amdgcn_s_setprio(2);     // the serial chain every batch waits for: its few instructions go first (3 while a wave updates)

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);          // the wave's index in a scalar register (loop-invariant)
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    const int32_t* __restrict__ pm = perm + (size_t)b * N;
    int32_t* __restrict__ out = idx_out + (size_t)b * npoint;

    fvec px, py, pz, pt;
    float lox = FPS_BIG, loy = FPS_BIG, loz = FPS_BIG, hix = -FPS_BIG, hiy = -FPS_BIG, hiz = -FPS_BIG;
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        int s = (wave * PPT + i) * 64 + lane;       // position in Morton order, SLOT-major: slot i of a wave = 64 consecutive points
        bool ok = s < N;
        int o = ok ? pm[s] : 0x7fffffff;
        s_po[i * BLOCK + tid] = o;
        px[i] = ok ? p[o * 3 + 0] : 0.f;
        py[i] = ok ? p[o * 3 + 1] : 0.f;
        pz[i] = ok ? p[o * 3 + 2] : 0.f;
        pt[i] = ok ? 1e10f : -1.0f;
        if (ok) {
            lox = fminf(lox, px[i]); hix = fmaxf(hix, px[i]);
            loy = fminf(loy, py[i]); hiy = fmaxf(hiy, py[i]);
            loz = fminf(loz, pz[i]); hiz = fmaxf(hiz, pz[i]);
        }
    }
    // the wave's bounding box (uniform); an all-padding wave keeps the empty box (+-1e18: L ~ 3e36 stays finite)
    lox = wave_min_f32_fused(lox); loy = wave_min_f32_fused(loy); loz = wave_min_f32_fused(loz);
    hix = wave_max_f32_fused(hix); hiy = wave_max_f32_fused(hiy); hiz = wave_max_f32_fused(hiz);

    // second level: the box of every PAIR of slots (128 consecutive Morton points), kept by lane g = pair index
    float glx = FPS_BIG, gly = FPS_BIG, glz = FPS_BIG, ghx = -FPS_BIG, ghy = -FPS_BIG, ghz = -FPS_BIG;
#pragma unroll
    for (int g = 0; g < PPT / 2; g++) {
        float ax_ = FPS_BIG, ay_ = FPS_BIG, az_ = FPS_BIG, bx_ = -FPS_BIG, by_ = -FPS_BIG, bz_ = -FPS_BIG;
#pragma unroll
        for (int i = 2 * g; i < 2 * g + 2; i++)
            if (pt[i] >= 0.f) {
                ax_ = fminf(ax_, px[i]); bx_ = fmaxf(bx_, px[i]);
                ay_ = fminf(ay_, py[i]); by_ = fmaxf(by_, py[i]);
                az_ = fminf(az_, pz[i]); bz_ = fmaxf(bz_, pz[i]);
            }
        ax_ = wave_min_f32_fused(ax_); ay_ = wave_min_f32_fused(ay_); az_ = wave_min_f32_fused(az_);
        bx_ = wave_max_f32_fused(bx_); by_ = wave_max_f32_fused(by_); bz_ = wave_max_f32_fused(bz_);
        if (lane == g) { glx = ax_; gly = ay_; glz = az_; ghx = bx_; ghy = by_; ghz = bz_; }
    }
    if (tid == 0 && npoint > 0) out[0] = 0;
    if (tid < 192) (&xt[0][0])[tid] = 0u;
    __syncthreads();
    float x0 = p[0], y0 = p[1], z0 = p[2];
    // cached candidate of this wave (uniform): value, original index, coordinates
    float cval = 1e10f; int corig = 0x7fffffff; float cx = 0.f, cy = 0.f, cz = 0.f;
    fps_cand_t cand; fps_cand_set(cand, wave, cx, cy, cz, cval, corig);
    bool first = true;
    int wprev = -1;                            // the wave whose candidate is the current sample
    FPS_T(unsigned long long q_test = 0, q_dist = 0, q_max = 0, q_search = 0, q_pub = 0, q_coll = 0, q_nupd = 0, q_npair = 0; unsigned long long q0 = FPS_NOW(wave); const unsigned long long q_begin = q0;)
    int cb = 1;                                // exchange cell of sample j: j % 3
    for (int j = 1; j < npoint; j++) {
        // which pairs of slots can the new sample reach?  Lane g tests pair g's box (a lower bound of the distance to anything in it)
        // against the WAVE's largest distance -- no pair keeps a maximum of its own: the wave's bounds every point's, so
        // "bound >= cval" proves no change.  The eight pair boxes together are also a tighter test of the whole wave than its one box.
        unsigned live = (1u << (PPT / 2)) - 1u;
        if (!first) {
            const float hx = fmaxf(fmaxf(glx - x0, x0 - ghx), 0.f), hy = fmaxf(fmaxf(gly - y0, y0 - ghy), 0.f), hz = fmaxf(fmaxf(glz - z0, z0 - ghz), 0.f);
            const float Lg = __fadd_rn(__fadd_rn(__fmul_rn(hx, hx), __fmul_rn(hy, hy)), __fmul_rn(hz, hz));
            live = (unsigned)__ballot(Lg < cval) & ((1u << (PPT / 2)) - 1u);
        }
        FPS_T(unsigned long long q1 = FPS_NOW(wave); q_test += q1 - q0;)
        if (live != 0u) {
            FPS_T(q_nupd++; q_npair += __builtin_popcount(live);)
            // the sample waits for the updating wave(s): ahead of the three waves that share the SIMD and are still
            // working through their own bound test / exchange read (oldest-first arbitration otherwise: tools/fps_timing.py
            // shows the fourth wave of a SIMD taking 2-3x as long for the same instructions)
            __builtin_amdgcn_s_setprio(3);
            const f32x2 qx = {x0, x0}, qy = {y0, y0}, qz = {z0, z0};
#pragma unroll
            for (int g = 0; g < PPT / 2; g++) {
                if ((live >> g) & 1u) {                          // wave-uniform
                    const int i = 2 * g;
#if FPS_PAIR_ASM
                    // the pair's two squared distances, hand-ordered: a packed fp32 result read by the NEXT instruction costs a wait
                    // state; with three temporaries only the final sum waits (the compiler's two-temporary order needs three to four)
                    f32x2 d, tb, tc;
                    asm("v_pk_add_f32 %[a], %[px], %[qx] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                        "v_pk_add_f32 %[b], %[py], %[qy] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                        "v_pk_add_f32 %[c], %[pz], %[qz] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                        "v_pk_mul_f32 %[a], %[a], %[a]\n\t"
                        "v_pk_mul_f32 %[b], %[b], %[b]\n\t"
                        "v_pk_mul_f32 %[c], %[c], %[c]\n\t"
                        "v_pk_add_f32 %[a], %[a], %[b]\n\t"
                        "s_nop 0\n\t"
                        "v_pk_add_f32 %[a], %[a], %[c]"
                        : [a] "=&v"(d), [b] "=&v"(tb), [c] "=&v"(tc)
                        : [px] "v"((f32x2){px[i], px[i + 1]}), [py] "v"((f32x2){py[i], py[i + 1]}), [pz] "v"((f32x2){pz[i], pz[i + 1]}),
                          [qx] "s"(qx), [qy] "s"(qy), [qz] "s"(qz));
#else
                    f32x2 dx = (f32x2){px[i], px[i + 1]} - qx;
                    f32x2 dy = (f32x2){py[i], py[i + 1]} - qy;
                    f32x2 dz = (f32x2){pz[i], pz[i + 1]} - qz;
                    f32x2 d = (dx * dx + dy * dy) + dz * dz;
#endif
                    pt[i] = __builtin_fminf(pt[i], d.x); pt[i + 1] = __builtin_fminf(pt[i + 1], d.y);
                }
            }
            FPS_T(unsigned long long q2 = FPS_NOW(wave); q_dist += q2 - q1;)
            // Round 6: did the candidate's own distance shrink?  (the pair update's arithmetic, operation for operation, on the candidate's
            // coordinates; a pair that was not live cannot have changed it: its bound is >= cval.)  If not, it is still the wave's maximum --
            // every other distance only shrank, and it beat its equals on the index rule already -- and the wave republishes it as it is:
            // no lane maximum, slot masks, wave maximum or owner search (2/3 of the updates of the waves next to the winner's; the
            // winner's own wave always recomputes: its candidate IS the sample).
            bool shrunk = wave_u == wprev || first;
            if (!shrunk) {
                const float ex = __fsub_rn(cx, x0), ey = __fsub_rn(cy, y0), ez = __fsub_rn(cz, z0);
                shrunk = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)) < cval;
            }
            if (shrunk) {
#if (FPS_V & 4)
            float best;
            if (PPT == 16) {                         // a tree: neighbouring instructions are independent (a chain waits ~6 cycles per link)
                const float m0 = __builtin_fmaxf(__builtin_fmaxf(pt[0], pt[1]), pt[2]), m1 = __builtin_fmaxf(__builtin_fmaxf(pt[3], pt[4]), pt[5]);
                const float m2 = __builtin_fmaxf(__builtin_fmaxf(pt[6], pt[7]), pt[8]), m3 = __builtin_fmaxf(__builtin_fmaxf(pt[9], pt[10]), pt[11]);
                const float m4 = __builtin_fmaxf(__builtin_fmaxf(pt[12], pt[13]), pt[14]);
                best = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m0, m1), m2), __builtin_fmaxf(__builtin_fmaxf(m3, m4), pt[15]));
            } else {
                best = pt[0];
#pragma unroll
                for (int i = 1; i < PPT; i++) best = __builtin_fmaxf(best, pt[i]);
            }
#else
            float best = pt[0];
#pragma unroll
            for (int i = 1; i < PPT; i++) best = __builtin_fmaxf(best, pt[i]);
#endif
            unsigned eqbits;
#if (FPS_V & 4)
            const int wvec = PPT == 16 ? wave_max_eq2_16(pt, best, eqbits) : WaveMaxEq<PPT>::run(pt, best, eqbits);
#else
            const int wvec = WaveMaxEq<PPT>::run(pt, best, eqbits);          // lane 63 = the wave maximum
#endif
            // every lane fetches the original index of ITS OWN candidate (lowest slot holding the lane's maximum) as soon as the slot
            // masks exist; the owner's word is read out below, after the scalar search and the coordinate selects, so the LDS round
            // trip is hidden (rounds 3-4 read s_po at the owner's address inside the fast path, after the search: an exposed round
            // trip per update -- a load whose only use sits in a branch is sunk into it; the sched_barriers keep the order written)
            const int wmax = __builtin_amdgcn_readlane(wvec, 63);          // first: the load's address arithmetic fills its wait states
            const int myorig = s_po[__builtin_ctz(eqbits) * BLOCK + tid];
            __builtin_amdgcn_sched_barrier(0);
            FPS_T(unsigned long long q3 = FPS_NOW(wave); q_max += q3 - q2;)
            const float wmaxf = __int_as_float(wmax);
            // candidates = (lane, slot) with t == wmax; the one with the LOWEST ORIGINAL index wins.  Fast path (a
            // unique maximum, the overwhelmingly common case).  An updating wave is usually ALONE on its SIMD and issues one
            // instruction every ~4-5 cycles, so its instruction count is the latency of the whole sample (measured with
            // tools/fps_timing.py: ~1450 cycles per update, of which a per-slot ballot + scalar select chain took ~600).
            // Here every lane finds the lowest slot holding ITS maximum and how many slots do (3 VALU per slot, no scalar
            // chain); one ballot finds the lanes holding the wave maximum; unique lane with a unique slot = fast path.
            const unsigned long long anym = __ballot(best == wmaxf);
            const int owner0 = __builtin_ctzll(anym);
            const unsigned ownbits = (unsigned)__builtin_amdgcn_readlane((int)eqbits, owner0);       // the owner lane's slots holding the maximum
#if (FPS_V & 4)
            const int total = (__popcll(anym) + __popc(ownbits) == 2) ? 1 : 2;        // both counts are >= 1
#else
            const int total = (__popcll(anym) == 1 && __popc(ownbits) == 1) ? 1 : 2;
#endif
            int istar = __builtin_ctz(ownbits);
            // the fast path's selects, taken before the branch (wasted on the rare tie path) so that they too run under the load
            const float fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[istar]), owner0));
            const float fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[istar]), owner0));
            const float fz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[istar]), owner0));
            __builtin_amdgcn_sched_barrier(0);
            const int corig_fast = __builtin_amdgcn_readlane(myorig, owner0);
            if (total == 1) {
                corig = corig_fast; cx = fx; cy = fy; cz = fz;
            } else {                                  // exact ties (duplicated points, lattices): rare, any cost is fine
                int bo = 0x7fffffff; float bx = 0.f, by = 0.f, bz = 0.f;
                if (best == wmaxf) {
#pragma unroll
                    for (int i = PPT - 1; i >= 0; i--) {
                        const int oi = s_po[i * BLOCK + tid];
                        if (pt[i] == wmaxf && oi <= bo) { bo = oi; bx = px[i]; by = py[i]; bz = pz[i]; }
                    }
                }
                corig = wave_min_i32_fused(bo);
                const int owner = __builtin_ctzll(__ballot(bo == corig));
                cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx), owner));
                cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(by), owner));
                cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bz), owner));
            }
            cval = wmaxf;
            first = false;
            fps_cand_set(cand, wave, cx, cy, cz, cval, corig);
            FPS_T(q_search += FPS_NOW(wave) - q3;)
            }
            __builtin_amdgcn_s_setprio(2);
        }
#if FPS_V & 2
        else __builtin_amdgcn_s_setprio(2);
#endif
        // Exchange (fps_xt_publish / fps_xt_collect above): candidates folded into the sample's cell, one barrier, one LDS read
        FPS_T(unsigned long long q4 = FPS_NOW(wave);)
        if (lane == 0) fps_xt_publish(xt, cb, wave, cand);
        __syncthreads();
        FPS_T(unsigned long long q5 = FPS_NOW(wave); q_pub += q5 - q4;)
        int wwin;
        const int gorig = fps_xt_collect(xt, cb, lane, x0, y0, z0, wwin);
#if FPS_V & 2
        // the winner's wave certainly updates next (the sample is one of its points) and its Morton neighbours (on the two adjacent SIMDs)
        // probably do: they take their bound test ahead of the three waves they share a SIMD with instead of in arrival order
        if ((unsigned)(wave_u - wwin + 1) <= 2u) __builtin_amdgcn_s_setprio(3);          // scalar compare + branch
#endif
        wprev = wwin;
        cb = cb == 2 ? 0 : cb + 1;
        if (tid == 0) out[j] = gorig;
        FPS_T(q0 = FPS_NOW(wave); q_coll += q0 - q5;)
    }
    FPS_T(if (b == 0 && lane == 0) { unsigned long long* d = prcnn_fps_dbg + 144 + wave * 16; d[0] = q_test; d[1] = q_dist; d[2] = q_max; d[3] = q_search; d[4] = q_pub;
                                     d[5] = q_coll; d[6] = q_nupd; d[7] = q_npair; d[8] = FPS_NOW(wave) - q_begin; })
}

// =====================================================================================================
// Two samples per exchange (fps_batch_kernel, round 6).  The sample loop is a serial chain -- update, publish, barrier, collect -- of
// ~1 700 cycles, and only the 2-4 waves a sample reaches do anything in it.  But the table of candidates already holds the NEXT sample
// in most rounds: let W1 be the wave of the largest candidate (sample j) and W2 the wave of the largest among the other fifteen,
// value v2.  Sample j + 1 is W2's candidate, provably and before any update has run, when
//   (a) sample j cannot change W2's points: W2 is not in the REACH MASK published with W1's candidate -- the waves whose box bound
//       against the candidate was below their maximum when the candidate was found (maxima only shrink: a superset of today's);
//   (b) v2 > s(W1), an upper bound of the SECOND largest min-distance of W1's points (published too): once sample j is taken W1's
//       best point drops to 0 and every other only shrinks, so W1's next candidate is <= s(W1);
//   (c) no third wave holds v2 too (every other wave's next candidate is <= its current one <= v2; a tie would go to the index rule).
// Distances only shrink, so these are exact: the sample SEQUENCE is the one-at-a-time sequence (tools/fps_batch_sim.py replays the rule
// on the host against plain FPS: 1.74 samples per exchange on the uniform and the LiDAR-like clouds).  When the proof fails the round
// carries one sample, as before.  A wave in neither sample's reach mask skips the round with scalar work only; a reached wave tests
// its own pairs, applies both samples to the pairs each can reach, then finds its new candidate, second value and reach mask ONCE --
// and not at all when its candidate's own distance did not shrink (then it is still the maximum, and its mask and bound still hold).
// Every wave runs the proof after the barrier (same table, same decision).
// MEASURED (16 384 -> 4 096, bs32, tools/fps_timing.py): exact -- every FPS test passes on it -- and 1.74 samples per round as predicted,
// but a round costs 3 200 cycles against 1 775 for a one-sample round of fps_slot_kernel: 3.09 vs 3.02 ms.  A wave that shares its SIMD
// with three others issues one instruction per 8-16 cycles whatever its kind, so the ~30 instructions of the proof cost every wave
// 500-700 cycles per round, and the second value + reach mask + survival test add ~700 to a full update (1 750 vs 970).  Two other
// arrangements of the same idea: the wave-box test inside the proof instead of a published mask (47 VALU in the proof, nothing added to
// the update): 2.94 ms, -2.5 %; the last wave to publish running the proof alone for everybody (a returning LDS add as the arrival
// count, result through the buffer): 730 cycles of a lone wave's dependent instructions on the critical path, 3.15 ms, +4 %.  Opt-in
// (PRCNN_FPS_BATCH=1), kept as the worked-out form of "more than one sample per barrier".
// Exchange buffer (three rotate): words 4w.. = {x, y, z, value} of wave w, 64+4w.. = {second value, original index, reach mask, -},
// 128-129 = the 64-bit maximum cell of fps_slot_kernel.
// =====================================================================================================
struct fps_bcand_t { float x, y, z, v, s; unsigned orig, mask; unsigned long long key; };
__device__ __forceinline__ void fps_bcand_set(fps_bcand_t& c, int wave, float cx, float cy, float cz, float cval, float csec, int corig, unsigned cmask) {
    const unsigned hi = (unsigned)__float_as_int(cval) ^ 0x80000000u;
    const unsigned o = (unsigned)min(corig, 0xFFFFFFF);
    c.x = cx; c.y = cy; c.z = cz; c.v = cval; c.s = csec; c.orig = o; c.mask = cmask; c.key = ((unsigned long long)hi << 32) | ((0xFFFFFFFu - o) << 4) | (unsigned)wave;
    asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.v), "+v"(c.s), "+v"(c.orig), "+v"(c.mask), "+v"(c.key));
}
#define FPS_BT_WORDS 160

template <int PPT, int NW>
__global__ __launch_bounds__(NW * 64) void fps_batch_kernel(const float* __restrict__ xyz, const int32_t* __restrict__ perm,
                                                           int N, int npoint, int32_t* __restrict__ idx_out) {
    static_assert(NW == 16 && PPT == 16, "rows of 16 lanes = the 16 waves; 8 pair boxes");
    constexpr int BLOCK = NW * 64;
    typedef typename fvec_t<PPT>::type fvec;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(256))) unsigned xt[3][FPS_BT_WORDS];
    __shared__ __attribute__((aligned(32))) float wbox[16][8];          // {lo x y z, hi x y z, -, -} of every wave
    extern __shared__ int s_po[];
    __builtin_amdgcn_s_setprio(2);

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    const int32_t* __restrict__ pm = perm + (size_t)b * N;
    int32_t* __restrict__ out = idx_out + (size_t)b * npoint;

    fvec px, py, pz, pt;
    float lox = FPS_BIG, loy = FPS_BIG, loz = FPS_BIG, hix = -FPS_BIG, hiy = -FPS_BIG, hiz = -FPS_BIG;
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        int s = (wave * PPT + i) * 64 + lane;
        bool ok = s < N;
        int o = ok ? pm[s] : 0x7fffffff;
        s_po[i * BLOCK + tid] = o;
        px[i] = ok ? p[o * 3 + 0] : 0.f;
        py[i] = ok ? p[o * 3 + 1] : 0.f;
        pz[i] = ok ? p[o * 3 + 2] : 0.f;
        pt[i] = ok ? 1e10f : -1.0f;
        if (ok) {
            lox = fminf(lox, px[i]); hix = fmaxf(hix, px[i]);
            loy = fminf(loy, py[i]); hiy = fmaxf(hiy, py[i]);
            loz = fminf(loz, pz[i]); hiz = fmaxf(hiz, pz[i]);
        }
    }
    lox = wave_min_f32_fused(lox); loy = wave_min_f32_fused(loy); loz = wave_min_f32_fused(loz);
    hix = wave_max_f32_fused(hix); hiy = wave_max_f32_fused(hiy); hiz = wave_max_f32_fused(hiz);
    if (lane == 0) { float* wb = wbox[wave]; wb[0] = lox; wb[1] = loy; wb[2] = loz; wb[3] = hix; wb[4] = hiy; wb[5] = hiz; }

    // the pair box a lane tests after the barrier: lanes 0-7 pair l15 against sample 1, lanes 16-23 against sample 2 (the rest: an empty box)
    float glx = FPS_BIG, gly = FPS_BIG, glz = FPS_BIG, ghx = -FPS_BIG, ghy = -FPS_BIG, ghz = -FPS_BIG;
#pragma unroll
    for (int g = 0; g < PPT / 2; g++) {
        float ax_ = FPS_BIG, ay_ = FPS_BIG, az_ = FPS_BIG, bx_ = -FPS_BIG, by_ = -FPS_BIG, bz_ = -FPS_BIG;
#pragma unroll
        for (int i = 2 * g; i < 2 * g + 2; i++)
            if (pt[i] >= 0.f) {
                ax_ = fminf(ax_, px[i]); bx_ = fmaxf(bx_, px[i]);
                ay_ = fminf(ay_, py[i]); by_ = fmaxf(by_, py[i]);
                az_ = fminf(az_, pz[i]); bz_ = fmaxf(bz_, pz[i]);
            }
        ax_ = wave_min_f32_fused(ax_); ay_ = wave_min_f32_fused(ay_); az_ = wave_min_f32_fused(az_);
        bx_ = wave_max_f32_fused(bx_); by_ = wave_max_f32_fused(by_); bz_ = wave_max_f32_fused(bz_);
        if (lane < 32 && l15 == g) { glx = ax_; gly = ay_; glz = az_; ghx = bx_; ghy = by_; ghz = bz_; }
    }
    if (tid == 0 && npoint > 0) out[0] = 0;
    // every buffer starts with value 1e10 for every wave (the reach mask of the first candidates is taken against "the previous table")
    if (tid < 3 * FPS_BT_WORDS) (&xt[0][0])[tid] = (tid % FPS_BT_WORDS < 64 && (tid & 3) == 3) ? __float_as_uint(1e10f) : 0u;
    __syncthreads();
    const bool row_s2 = (lane & 16) != 0;          // odd rows look at sample 2

    float x1 = p[0], y1 = p[1], z1 = p[2], x2 = 0.f, y2 = 0.f, z2 = 0.f;
    float cval = 1e10f, csec = 1e10f; int corig = 0x7fffffff; float cx = 0.f, cy = 0.f, cz = 0.f;
    fps_bcand_t cand; fps_bcand_set(cand, wave, cx, cy, cz, cval, csec, corig, 0xFFFFu);
    unsigned live1 = (1u << (PPT / 2)) - 1u, live2 = 0u;          // round 0: sample 0 reaches everything
    bool second = false, winner = false;          // winner: one of this round's samples is this wave's candidate
    FPS_T(unsigned long long q_test = 0, q_dist = 0, q_max = 0, q_search = 0, q_pub = 0, q_coll = 0, q_nupd = 0, q_npair = 0, q_rounds = 0, q_proof = 0, q_nproof = 0, q_nfull = 0; unsigned long long q0 = FPS_NOW(wave); const unsigned long long q_begin = q0;)
    int cb = 1;
    int j = 1;
    while (j < npoint) {
        FPS_T(unsigned long long q1 = FPS_NOW(wave); q_rounds++;)
        if ((live1 | live2) != 0u) {
            FPS_T(q_nupd++; q_npair += __builtin_popcount(live1) + __builtin_popcount(live2);)
#define FPS_BATCH_PAIRS(LIVE, QX, QY, QZ)                                                                                          \
            {                                                                                                                      \
                const f32x2 qx = {QX, QX}, qy = {QY, QY}, qz = {QZ, QZ};                                                           \
                _Pragma("unroll") for (int g = 0; g < PPT / 2; g++) {                                                              \
                    if (((LIVE) >> g) & 1u) {                                                                                      \
                        const int i = 2 * g;                                                                                       \
                        f32x2 d, tb, tc;                                                                                           \
                        asm("v_pk_add_f32 %[a], %[px], %[qx] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                       \
                            "v_pk_add_f32 %[b], %[py], %[qy] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                       \
                            "v_pk_add_f32 %[c], %[pz], %[qz] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                       \
                            "v_pk_mul_f32 %[a], %[a], %[a]\n\t"                                                                    \
                            "v_pk_mul_f32 %[b], %[b], %[b]\n\t"                                                                    \
                            "v_pk_mul_f32 %[c], %[c], %[c]\n\t"                                                                    \
                            "v_pk_add_f32 %[a], %[a], %[b]\n\t"                                                                    \
                            "s_nop 0\n\t"                                                                                          \
                            "v_pk_add_f32 %[a], %[a], %[c]"                                                                        \
                            : [a] "=&v"(d), [b] "=&v"(tb), [c] "=&v"(tc)                                                           \
                            : [px] "v"((f32x2){px[i], px[i + 1]}), [py] "v"((f32x2){py[i], py[i + 1]}), [pz] "v"((f32x2){pz[i], pz[i + 1]}), \
                              [qx] "s"(qx), [qy] "s"(qy), [qz] "s"(qz));                                                           \
                        pt[i] = __builtin_fminf(pt[i], d.x); pt[i + 1] = __builtin_fminf(pt[i + 1], d.y);                          \
                    }                                                                                                              \
                }                                                                                                                  \
            }
            if (live1 != 0u) FPS_BATCH_PAIRS(live1, x1, y1, z1)
            if (live2 != 0u) FPS_BATCH_PAIRS(live2, x2, y2, z2)
#undef FPS_BATCH_PAIRS
            FPS_T(unsigned long long q2 = FPS_NOW(wave); q_dist += q2 - q1;)
            // did the candidate's own distance shrink?  (the pair update's arithmetic, operation for operation, on the candidate's
            // coordinates; a pair that was not live cannot have changed it: its bound is >= cval)  If not it is still the wave's maximum --
            // every other distance only shrank -- and the published second value and reach mask still bound what they bound.
            const float e1x = __fsub_rn(cx, x1), e1y = __fsub_rn(cy, y1), e1z = __fsub_rn(cz, z1);
            const float e1 = __fadd_rn(__fadd_rn(__fmul_rn(e1x, e1x), __fmul_rn(e1y, e1y)), __fmul_rn(e1z, e1z));
            const float e2x = __fsub_rn(cx, x2), e2y = __fsub_rn(cy, y2), e2z = __fsub_rn(cz, z2);
            const float e2 = __fadd_rn(__fadd_rn(__fmul_rn(e2x, e2x), __fmul_rn(e2y, e2y)), __fmul_rn(e2z, e2z));
            const bool shrunk = winner || (e1 < cval) || (second && e2 < cval) || cval >= 1e10f;
            if (shrunk) {
            FPS_T(q_nfull++;)
            // operands of the reach mask, consumed after the search: the waves' boxes and their values in the PREVIOUS table (the one
            // this round's samples came from; today's values are <= those)
            const f32x4 wlo = *reinterpret_cast<const f32x4*>(&wbox[l15][0]);
            const f32x2 whi = *reinterpret_cast<const f32x2*>(&wbox[l15][4]);
            const float pval = __uint_as_float(xt[cb == 0 ? 2 : cb - 1][l15 * 4 + 3]);
            const float m0 = __builtin_fmaxf(__builtin_fmaxf(pt[0], pt[1]), pt[2]), m1 = __builtin_fmaxf(__builtin_fmaxf(pt[3], pt[4]), pt[5]);
            const float m2 = __builtin_fmaxf(__builtin_fmaxf(pt[6], pt[7]), pt[8]), m3 = __builtin_fmaxf(__builtin_fmaxf(pt[9], pt[10]), pt[11]);
            const float m4 = __builtin_fmaxf(__builtin_fmaxf(pt[12], pt[13]), pt[14]);
            const float best = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(m0, m1), m2), __builtin_fmaxf(__builtin_fmaxf(m3, m4), pt[15]));
            unsigned eqbits;
            const int wvec = wave_max_eq2_16(pt, best, eqbits);
            const int wmax = __builtin_amdgcn_readlane(wvec, 63);
            const int myorig = s_po[__builtin_ctz(eqbits) * BLOCK + tid];
            __builtin_amdgcn_sched_barrier(0);
            FPS_T(unsigned long long q3 = FPS_NOW(wave); q_max += q3 - q2;)
            const float wmaxf = __int_as_float(wmax);
            const unsigned long long anym = __ballot(best == wmaxf);
            const int owner0 = __builtin_ctzll(anym);
            const unsigned ownbits = (unsigned)__builtin_amdgcn_readlane((int)eqbits, owner0);
            const int total = (__popcll(anym) + __popc(ownbits) == 2) ? 1 : 2;
            int istar = __builtin_ctz(ownbits);
            const float fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[istar]), owner0));
            const float fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[istar]), owner0));
            const float fz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[istar]), owner0));
            __builtin_amdgcn_sched_barrier(0);
            const int corig_fast = __builtin_amdgcn_readlane(myorig, owner0);
            if (total == 1) {
                corig = corig_fast; cx = fx; cy = fy; cz = fz;
                // the second value: the lane maxima of the other lanes, and the owner lane's with its slot istar left out
                const float keep = pt[istar];
                pt[istar] = -2.0f;
                const float n0 = __builtin_fmaxf(__builtin_fmaxf(pt[0], pt[1]), pt[2]), n1 = __builtin_fmaxf(__builtin_fmaxf(pt[3], pt[4]), pt[5]);
                const float n2 = __builtin_fmaxf(__builtin_fmaxf(pt[6], pt[7]), pt[8]), n3 = __builtin_fmaxf(__builtin_fmaxf(pt[9], pt[10]), pt[11]);
                const float n4 = __builtin_fmaxf(__builtin_fmaxf(pt[12], pt[13]), pt[14]);
                const float lane2 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(n0, n1), n2), __builtin_fmaxf(__builtin_fmaxf(n3, n4), pt[15]));
                pt[istar] = keep;
                csec = wave_max_f32_fused(lane == owner0 ? lane2 : best);
            } else {
                int bo = 0x7fffffff; float bx = 0.f, by = 0.f, bz = 0.f;
                if (best == wmaxf) {
#pragma unroll
                    for (int i = PPT - 1; i >= 0; i--) {
                        const int oi = s_po[i * BLOCK + tid];
                        if (pt[i] == wmaxf && oi <= bo) { bo = oi; bx = px[i]; by = py[i]; bz = pz[i]; }
                    }
                }
                corig = wave_min_i32_fused(bo);
                const int owner = __builtin_ctzll(__ballot(bo == corig));
                cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bx), owner));
                cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(by), owner));
                cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bz), owner));
                csec = wmaxf;                      // several points hold the maximum: the wave's next candidate may equal this one
            }
            cval = wmaxf;
            // the candidate's reach mask: lane w tests wave w's box against it (the bound fps_slot_kernel uses for its own wave)
            const float hx = fmaxf(fmaxf(wlo.x - cx, cx - wlo.w), 0.f), hy = fmaxf(fmaxf(wlo.y - cy, cy - whi.x), 0.f), hz = fmaxf(fmaxf(wlo.z - cz, cz - whi.y), 0.f);
            const float Lw = __fadd_rn(__fadd_rn(__fmul_rn(hx, hx), __fmul_rn(hy, hy)), __fmul_rn(hz, hz));
            const unsigned cmask = ((unsigned)__ballot(Lw < pval) & 0xFFFFu) | (1u << wave_u);
            fps_bcand_set(cand, wave, cx, cy, cz, cval, csec, corig, cmask);
            FPS_T(q_search += FPS_NOW(wave) - q3;)
            }
            __builtin_amdgcn_s_setprio(2);
        }
        FPS_T(unsigned long long q4 = FPS_NOW(wave);)
        unsigned* t = xt[cb];
        if (lane == 0) {
            *reinterpret_cast<f32x4*>(t + wave * 4) = (f32x4){cand.x, cand.y, cand.z, cand.v};
            *reinterpret_cast<u32x4*>(t + 64 + wave * 4) = (u32x4){__float_as_uint(cand.s), cand.orig, cand.mask, 0u};
            asm volatile("ds_max_u64 %0, %1 offset:512" : : "v"((unsigned)(size_t)&t[0]), "v"(cand.key) : "memory");
            if (wave == 0) *(unsigned long long*)&xt[cb == 2 ? 0 : cb + 1][128] = 0ULL;
        }
        __syncthreads();
        FPS_T(unsigned long long q5 = FPS_NOW(wave); q_pub += q5 - q4;)
        // ---- collect + the proof for a second sample (every wave, the same table: the same decision) ----
        const f32x4 ta = *reinterpret_cast<const f32x4*>(t + l15 * 4);                 // {x, y, z, value} of wave l15
        const u32x4 tb = *reinterpret_cast<const u32x4*>(t + 64 + l15 * 4);            // {second value, original index, reach mask, -}
        const unsigned klo = (unsigned)__builtin_amdgcn_readfirstlane((int)t[128]);
        const int w1 = (int)(klo & 15u);
        const int gorig1 = (int)(0xFFFFFFFu - (klo >> 4));
        if ((unsigned)(wave_u - w1 + 1) <= 2u) __builtin_amdgcn_s_setprio(3);          // the winner's wave and its Morton neighbours go first
        const int s1 = __builtin_amdgcn_readlane((int)tb.x, w1);
        const unsigned mask1 = (unsigned)__builtin_amdgcn_readlane((int)tb.z, w1);
        // the largest value among the other fifteen waves (float bits compare as integers: values are >= 0, or -1 for an all-padding wave)
        const int tv = (l15 == w1) ? (int)0x80000000 : __float_as_int(ta.w);
        const int v2 = row0_max_i32_fused(tv);
        const unsigned m2nd = (unsigned)__ballot(tv == v2) & 0xFFFFu;
        const int w2 = __builtin_ctz(m2nd);
        const int gorig2 = __builtin_amdgcn_readlane((int)tb.y, w2);
        const unsigned mask2 = (unsigned)__builtin_amdgcn_readlane((int)tb.z, w2);
        second = (__builtin_popcount(m2nd) == 1) && (v2 > s1) && (v2 > 0) && (j + 1 < npoint) && ((mask1 >> w2) & 1u) == 0u;
        const unsigned reached = mask1 | (second ? mask2 : 0u);
        winner = wave_u == w1 || (second && wave_u == w2);
        live1 = 0u; live2 = 0u;
        if ((reached >> wave_u) & 1u) {
            __builtin_amdgcn_s_setprio(3);
            x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ta.x), w1));
            y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ta.y), w1));
            z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ta.z), w1));
            x2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ta.x), w2));
            y2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ta.y), w2));
            z2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ta.z), w2));
            const float qx = row_s2 ? x2 : x1, qy = row_s2 ? y2 : y1, qz = row_s2 ? z2 : z1;
            const float hx = fmaxf(fmaxf(glx - qx, qx - ghx), 0.f), hy = fmaxf(fmaxf(gly - qy, qy - ghy), 0.f), hz = fmaxf(fmaxf(glz - qz, qz - ghz), 0.f);
            const float Lg = __fadd_rn(__fadd_rn(__fmul_rn(hx, hx), __fmul_rn(hy, hy)), __fmul_rn(hz, hz));
            const unsigned reach = (unsigned)__ballot(Lg < cval);
            live1 = reach & 0xFFu;
            live2 = second ? (reach >> 16) & 0xFFu : 0u;
            if ((live1 | live2) == 0u) __builtin_amdgcn_s_setprio(2);
        } else __builtin_amdgcn_s_setprio(2);
        cb = cb == 2 ? 0 : cb + 1;
        if (tid == 0) { out[j] = gorig1; if (second) out[j + 1] = gorig2; }
        j += second ? 2 : 1;
        FPS_T(q0 = FPS_NOW(wave); q_coll += q0 - q5;)
    }
    FPS_T(if (b == 0 && lane == 0) { unsigned long long* d = prcnn_fps_dbg + 144 + wave * 16; d[0] = q_test; d[1] = q_dist; d[2] = q_max; d[3] = q_search; d[4] = q_pub;
                                     d[5] = q_coll; d[6] = q_nupd; d[7] = q_npair; d[8] = FPS_NOW(wave) - q_begin; d[9] = q_rounds; d[10] = q_proof; d[11] = q_nproof; d[12] = q_nfull; })
}

// =====================================================================================================
// N > 16384 (BASELINE config 5: 65 536 points per frame): the frame does not fit one workgroup's registers (1024 threads x
// 16 points), and re-reading it from L2 every iteration (fps_mem_kernel below) costs ~12 us per sample.  Here a frame is
// owned by S = ceil(N / 16384) workgroups, each keeping its 16384-point slice and running min-distances in VGPRs exactly
// as fps_reg_kernel does; per sample the S workgroups exchange their local candidates through L2:
//   * every workgroup publishes {value, global index, x, y, z} as five naturally aligned 8-byte {data, tag} granules,
//     each written by ONE device-scope (sc1) store, tag = the sample number -- a granule is either the old or the new
//     pair, never torn, so no separate flag / release fence is needed (MI355X_MICROARCH.md, hand-off price list);
//   * wave 0 of every workgroup polls the 5 S granules of its frame (one lane each, L1-bypassing loads) until all carry
//     the current tag, reduces them (max value, ties -> lowest slice == lowest point index) and broadcasts the winner
//     through LDS.  Slots are double-buffered by sample parity: a workgroup can run at most one sample ahead of the
//     slowest one of its frame (it needs that one's NEXT publication to proceed).
// The launcher clears the slots (hipMemsetAsync, tag 0xFFFFFFFF is never a sample number).  All workgroups of a frame
// must become resident for the frame to progress; they are consecutive in the grid, so at most the last dispatched frame
// waits for a CU, and it gets one when any other frame finishes.  The poll is bounded: a workgroup that does not see its
// partners within ~2^20 polls (about a second) gives up and fills the rest of its output with -1 (a hang would take the GPU down with it).
// Results are bit-identical to the single-workgroup kernels (same distance arithmetic, same tie rule).
// =====================================================================================================
#define FPS_MULTI_SLICE 16384
#define FPS_MULTI_MAX_SPLIT 12          // 5 * S polling lanes must fit one wave
#define FPS_MULTI_SPIN_LIMIT (1 << 20)

__global__ __launch_bounds__(1024) void fps_multi_kernel(const float* __restrict__ xyz, int N, int npoint, int S,
                                                         unsigned long long* __restrict__ slots, int32_t* __restrict__ idx_out,
                                                         int* __restrict__ timed_out) {
    constexpr int BLOCK = 1024, PPT = 16, NW = BLOCK / 64;
    typedef typename fvec_t<PPT>::type fvec;
    __shared__ float slot[2][NW][8];
    __shared__ float win[2][4];            // idx(bits), x, y, z of the frame-wide winner
    const int b = blockIdx.x / S, sl = blockIdx.x - b * S;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* __restrict__ pf = xyz + (size_t)b * N * 3;
    const int base = sl * FPS_MULTI_SLICE;
    const float* __restrict__ p = pf + (size_t)base * 3;
    const int n_loc = min(FPS_MULTI_SLICE, N - base);
    int32_t* __restrict__ out = idx_out + (size_t)b * npoint;
    unsigned long long* __restrict__ fslots = slots + (size_t)b * 2 * S * 5;          // [parity][slice][5]

    fvec px, py, pz, pt;
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        int k = tid * PPT + i;
        bool ok = k < n_loc;
        px[i] = ok ? p[k * 3 + 0] : 0.f;
        py[i] = ok ? p[k * 3 + 1] : 0.f;
        pz[i] = ok ? p[k * 3 + 2] : 0.f;
        pt[i] = ok ? 1e10f : -1.0f;
    }
    if (sl == 0 && tid == 0 && npoint > 0) out[0] = 0;
    float x0 = pf[0], y0 = pf[1], z0 = pf[2];
    bool dead = false;

    for (int j = 1; j < npoint; j++) {
        float best = -2.0f;
        const f32x2 qx = {x0, x0}, qy = {y0, y0}, qz = {z0, z0};
#pragma unroll
        for (int i = 0; i + 1 < PPT; i += 2) {
            f32x2 dx = (f32x2){px[i], px[i + 1]} - qx;
            f32x2 dy = (f32x2){py[i], py[i + 1]} - qy;
            f32x2 dz = (f32x2){pz[i], pz[i + 1]} - qz;
            f32x2 d = (dx * dx + dy * dy) + dz * dz;
            float t0 = __builtin_fminf(pt[i], d.x), t1 = __builtin_fminf(pt[i + 1], d.y);
            pt[i] = t0; pt[i + 1] = t1;
            best = __builtin_fmaxf(best, __builtin_fmaxf(t0, t1));
        }
        const int wmax = wave_max_i32_fused(__float_as_int(best));
        const float wmaxf = __int_as_float(wmax);
        const int owner = __builtin_ctzll(__ballot(best == wmaxf));
        int myslot = PPT - 1;                                             // (as fps_reg_kernel: per-lane lowest slot, owner's read out)
#pragma unroll
        for (int i = PPT - 2; i >= 0; i--) myslot = (pt[i] == best) ? i : myslot;
        const int istar = __builtin_amdgcn_readlane(myslot, owner);
        const int widx = (wave * 64 + owner) * PPT + istar;
        float sx = px[istar], sy = py[istar], sz = pz[istar];
        float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sx), owner));
        float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sy), owner));
        float wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sz), owner));
        float* s = slot[j & 1][wave];
        if (lane == 0) { s[0] = wmaxf; s[1] = __int_as_float(widx); s[2] = wx; s[3] = wy; s[4] = wz; }
        __syncthreads();
        if (wave == 0) {
            // workgroup-level candidate (as fps_reg_kernel), then the exchange with the frame's other slices
            const float* r = slot[j & 1][lane < NW ? lane : 0];
            int v = lane < NW ? __float_as_int(r[0]) : (int)0x80000000;
            const int gmax = row0_max_i32_fused(v);
            const int wwin = __builtin_ctzll(__ballot(v == gmax));
            const float* rw = slot[j & 1][wwin];
            unsigned long long* mine = fslots + ((size_t)(j & 1) * S + sl) * 5;
            if (lane < 5) {
                unsigned data = lane == 0 ? (unsigned)gmax : lane == 1 ? (unsigned)(base + __float_as_int(rw[1])) : __float_as_uint(rw[lane]);
                __hip_atomic_store(mine + lane, ((unsigned long long)(unsigned)j << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const int ng = 5 * S;
            const unsigned long long* src = fslots + (size_t)(j & 1) * S * 5 + (lane < ng ? lane : 0);
            unsigned long long g = 0;
            int spins = 0;
            while (!dead) {
                g = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = lane >= ng || (unsigned)(g >> 32) == (unsigned)j;
                if (__all(ok)) break;
                if (++spins > FPS_MULTI_SPIN_LIMIT) dead = true;
                __builtin_amdgcn_s_sleep(1);
            }
            // lane 5 t holds slice t's value: frame-wide max, ties -> lowest slice (= lowest point index)
            const int val = (lane < ng && lane % 5 == 0) ? (int)(unsigned)g : (int)0x80000000;
            const int fmax = wave_max_i32(val);
            const int wl = __builtin_ctzll(__ballot(val == fmax));                // lane 5 * winning slice
            const unsigned lo = (unsigned)g;
            if (lane == 0) {
                win[j & 1][0] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)lo, wl + 1));
                win[j & 1][1] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)lo, wl + 2));
                win[j & 1][2] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)lo, wl + 3));
                win[j & 1][3] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)lo, wl + 4));
                if (dead) win[j & 1][0] = __int_as_float(-1);
            }
        }
        __syncthreads();
        const int gidx = __float_as_int(win[j & 1][0]);
        x0 = win[j & 1][1]; y0 = win[j & 1][2]; z0 = win[j & 1][3];
        if (gidx < 0) {                              // a partner never showed up: give up loudly (-1 indices), do not hang
            for (int k = j + tid; k < npoint; k += BLOCK) if (sl == 0) out[k] = -1;
            // sticky, host-visible (pinned, device-mapped word): the NEXT prcnn_fps / prcnn_fps_status call reports it
            if (tid == 0 && timed_out) __hip_atomic_store(timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        if (sl == 0 && tid == 0) out[j] = gidx;
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

// The multi-slice kernel's poll is bounded; a slice that gave up marks this pinned host word (device-mapped, written with a
// system-scope store), which the host can read without synchronising with the stream.  -1 indices are never consumed
// silently: the next prcnn_fps call and prcnn_fps_status() return PRCNN_EHIP once the mark is set.
static int* fps_timeout_word() {
    static int* word = [] {
        int* w = nullptr;
        if (hipHostMalloc((void**)&w, sizeof(int), hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return (int*)nullptr; }
        *w = 0;
        return w;
    }();
    return word;
}

PRCNN_API int prcnn_fps_status(void) {
    int* w = fps_timeout_word();
    if (w && __atomic_exchange_n(w, 0, __ATOMIC_RELAXED) != 0)
        return prcnn_fail(PRCNN_EHIP, "prcnn_fps: a multi-workgroup launch (N > 16384) gave up waiting for a partner slice; "
                                      "its output holds -1 indices (were all ceil(N/16384) workgroups of a frame able to become resident?)");
    return PRCNN_OK;
}

PRCNN_API int prcnn_fps(const float* xyz, int B, int N, int npoint, float* tmp, int32_t* idx, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && N > 0 && npoint >= 0, "prcnn_fps: bad shape B=%d N=%d npoint=%d", B, N, npoint);
    PRCNN_REQUIRE(npoint <= N, "prcnn_fps: npoint %d > N %d", npoint, N);
    if (B == 0 || npoint == 0) return PRCNN_OK;          // empty problem: pointers may legitimately be null
    PRCNN_REQUIRE(xyz && idx, "prcnn_fps: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (tmp && N > 2048 && N <= 16384) {
        // spatially pruned path: tmp (B,N) 4-byte entries holds the Morton-order permutation
        int NP = 1;
        while (NP < N) NP <<= 1;
        int32_t* perm = reinterpret_cast<int32_t*>(tmp);
        static PrcnnLdsLimit sort_attr;
        if (!sort_attr.raise((const void*)fps_sort_kernel, 80 * 1024))
            return prcnn_fail(PRCNN_EHIP, "prcnn_fps: cannot raise the dynamic LDS limit of the sort kernel");
        const int sort_threads = NP / 16 < 64 ? 64 : NP / 16;
        hipLaunchKernelGGL(fps_sort_kernel, dim3(B), dim3(sort_threads), lds_sort_bytes(NP, sizeof(unsigned)), s, xyz, N, NP, perm);
        PRCNN_LAUNCH_CHECK("prcnn_fps(sort)");
        static PrcnnLdsLimit pruned_attr;
        if (!pruned_attr.raise((const void*)fps_pruned_kernel<16>, 16 * 4096))
            return prcnn_fail(PRCNN_EHIP, "prcnn_fps: cannot raise the dynamic LDS limit of the pruned kernel");
        // two-level pruning (fps_slot_kernel): pays at 16 points per lane (16 384 -> 4 096: 3.82 -> 3.56 ms), not at 4 or 8 (+4..9 %: the
        // pair tests and the separate running-maximum pass cost what the skipped slots save); PRCNN_FPS_SLOTS=0 is the A/B switch (same bits)
        const char* slots_env = getenv("PRCNN_FPS_SLOTS");              // read per call: the tests flip it in-process
        const bool slots = slots_env == nullptr || atoi(slots_env) != 0;
        static PrcnnLdsLimit slot_attr;
        if (slots && N > 8192 && !slot_attr.raise((const void*)fps_slot_kernel<16, 16>, 16 * 4096))
            return prcnn_fail(PRCNN_EHIP, "prcnn_fps: cannot raise the dynamic LDS limit of the slot kernel");
        // two samples per exchange where the second is provable (fps_batch_kernel; same bits): measured 2-3 % SLOWER than the slot kernel
        // (DESIGN.md 8), so it is opt-in, PRCNN_FPS_BATCH=1
        const char* batch_env = getenv("PRCNN_FPS_BATCH");
        const bool batch = batch_env != nullptr && atoi(batch_env) != 0;
        static PrcnnLdsLimit batch_attr;
        if (slots && batch && N > 8192 && !batch_attr.raise((const void*)fps_batch_kernel<16, 16>, 16 * 4096))
            return prcnn_fail(PRCNN_EHIP, "prcnn_fps: cannot raise the dynamic LDS limit of the batch kernel");
        if (slots && batch && N > 8192) hipLaunchKernelGGL((fps_batch_kernel<16, 16>), dim3(B), dim3(1024), 16 * 4096, s, xyz, perm, N, npoint, idx);
        else if (slots && N > 8192) hipLaunchKernelGGL((fps_slot_kernel<16, 16>), dim3(B), dim3(1024), 16 * 4096, s, xyz, perm, N, npoint, idx);
        // (4 096 -> 1 024 with the sort, round 6: 16 waves x 4 points per lane 578 us; the same kernel on 8 waves x 8: 629, on 4 waves x 16: 757)
        else if (N <= 4096) hipLaunchKernelGGL((fps_pruned_kernel<4>), dim3(B), dim3(1024), 4 * 4096, s, xyz, perm, N, npoint, idx);
        else if (N <= 8192) hipLaunchKernelGGL((fps_pruned_kernel<8>), dim3(B), dim3(1024), 8 * 4096, s, xyz, perm, N, npoint, idx);
        else hipLaunchKernelGGL((fps_pruned_kernel<16>), dim3(B), dim3(1024), 16 * 4096, s, xyz, perm, N, npoint, idx);
    }
    else if (N <= 64) launch_fps<64, 1>(xyz, B, N, npoint, idx, s);
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
        const int S = prcnn_divup(N, FPS_MULTI_SLICE);
        // multi-workgroup register-resident kernel: the exchange slots (B * 2 * S * 5 granules of 8 bytes) live at the start
        // of tmp, which is 4 N bytes per frame >= 80 S bytes.  Needs every slice of a frame resident at once: up to 256 CUs.
        static const bool use_mem = getenv("PRCNN_FPS_MEM") != nullptr;       // A/B switch: the L2 re-read kernel (same bits)
        if (!use_mem && S <= FPS_MULTI_MAX_SPLIT && ((uintptr_t)tmp & 7) == 0) {
            const size_t slot_bytes = (size_t)B * 2 * S * 5 * sizeof(unsigned long long);
            if (prcnn_fill_words(tmp, 0xFFFFFFFFu, slot_bytes / 4, s) != hipSuccess) return prcnn_fail(PRCNN_EHIP, "prcnn_fps: cannot clear the exchange slots");
            if (int st = prcnn_fps_status()) return st;               // an earlier launch of this kind timed out: say so now
            hipLaunchKernelGGL(fps_multi_kernel, dim3(B * S), dim3(1024), 0, s, xyz, N, npoint, S, reinterpret_cast<unsigned long long*>(tmp), idx,
                               fps_timeout_word());
        } else {
            hipLaunchKernelGGL(fps_mem_kernel, dim3(B), dim3(1024), 0, s, xyz, N, npoint, tmp, idx);
        }
    }
    PRCNN_LAUNCH_CHECK("prcnn_fps");
    return PRCNN_OK;
}

// =====================================================================================================
// Upstream tie ORDER (optional mode).  The upstream CUDA kernel runs T = min(1024, largest power of
// two <= N) threads per frame; thread t scans k = t, t+T, ... keeping its first maximum (strict >), and the tree
// reduction over threads keeps the lower thread on ties.  Among equal maxima the winner is therefore
// argmin (k mod T, k) -- not the lowest k of the canonical rule.  This kernel reproduces exactly that order (same
// distance arithmetic as every other FPS kernel here), so that an upstream build can be compared index for index even
// on clouds with duplicate points / lattices, the only inputs on which the two rules differ.  Not a fast path: the
// running min-distances live in `tmp` (HBM/L2).
// =====================================================================================================
__global__ __launch_bounds__(1024) void fps_upstream_order_kernel(const float* __restrict__ xyz, int N, int npoint, int T,
                                                                  float* __restrict__ tmp, int32_t* __restrict__ idx_out) {
    constexpr int NWMAX = 16;
    __shared__ int sval[2][NWMAX], stid[2][NWMAX], sidx[2][NWMAX];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    const float* __restrict__ p = xyz + (size_t)b * N * 3;
    float* __restrict__ t = tmp + (size_t)b * N;
    int32_t* __restrict__ out = idx_out + (size_t)b * npoint;
    for (int k = tid; k < N; k += blockDim.x) t[k] = 1e10f;
    if (tid == 0 && npoint > 0) out[0] = 0;
    __syncthreads();
    int old = 0;
    for (int j = 1; j < npoint; j++) {
        const float x0 = p[old * 3], y0 = p[old * 3 + 1], z0 = p[old * 3 + 2];
        float best = -2.0f;
        int bk = 0x7fffffff;
        if (tid < T) {
            for (int k = tid; k < N; k += T) {
                const float d = sqdist3(p[k * 3], p[k * 3 + 1], p[k * 3 + 2], x0, y0, z0);
                float v = t[k];
                v = d < v ? d : v;
                t[k] = v;
                if (v > best) { best = v; bk = k; }          // strict: the thread's FIRST maximum
            }
        }
        const int vb = __float_as_int(best);
        const int wmax = wave_max_i32(vb);
        const int wtid = wave_min_i32(vb == wmax ? tid : 0x7fffffff);           // ties -> lowest thread
        const int wk = __shfl(bk, wtid & 63);
        if (lane == 0) { sval[j & 1][wave] = wmax; stid[j & 1][wave] = wtid; sidx[j & 1][wave] = wk; }
        __syncthreads();
        const int v = lane < nw ? sval[j & 1][lane] : (int)0x80000000;
        const int td = lane < nw ? stid[j & 1][lane] : 0x7fffffff;
        const int id = lane < nw ? sidx[j & 1][lane] : 0;
        const int gmax = row0_max_i32(v);
        const int gtid = row0_min_i32(v == gmax ? td : 0x7fffffff);
        const int win = __builtin_ctzll(__ballot(v == gmax && td == gtid));
        old = __builtin_amdgcn_readlane(id, win);
        if (tid == 0) out[j] = old;
    }
}

PRCNN_API int prcnn_fps_order(const float* xyz, int B, int N, int npoint, int order, float* tmp, int32_t* idx, prcnn_stream_t stream) {
    if (order == PRCNN_FPS_ORDER_CANONICAL) return prcnn_fps(xyz, B, N, npoint, tmp, idx, stream);
    PRCNN_REQUIRE(order == PRCNN_FPS_ORDER_UPSTREAM, "prcnn_fps_order: unknown order %d", order);
    PRCNN_REQUIRE(B >= 0 && N > 0 && npoint >= 0 && npoint <= N, "prcnn_fps_order: bad shape B=%d N=%d npoint=%d", B, N, npoint);
    if (B == 0 || npoint == 0) return PRCNN_OK;
    PRCNN_REQUIRE(xyz && idx && tmp, "prcnn_fps_order: null pointer (the upstream-order kernel needs the (B,N) tmp buffer)");
    int T = 1;
    while (T * 2 <= N && T < 1024) T <<= 1;
    const int block = T < 64 ? 64 : T;
    hipLaunchKernelGGL(fps_upstream_order_kernel, dim3(B), dim3(block), 0, (hipStream_t)stream, xyz, N, npoint, T, tmp, idx);
    PRCNN_LAUNCH_CHECK("prcnn_fps_order");
    return PRCNN_OK;
}
