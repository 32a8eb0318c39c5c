// lds_sort.h -- workgroup-wide bitonic sort of 64-bit (or 32-bit) keys held in LDS, 16 keys per owning thread.
//
// Strides >= 16 are compare-exchanged through LDS (one barrier per stride), strides 8..1 and the first four stages
// entirely in the owning thread's registers: 16 384 keys take 55 LDS passes + 10 register phases instead of the 105
// LDS passes of a textbook bitonic network.  LDS layout: key i lives at word i + (i >> 4) (one pad word per 16 keys),
// so that a thread streaming its 16 consecutive keys and the strided compare-exchange passes are both conflict-free;
// 16 384 keys + padding = 136 KB of the CU's 160 KB.  Used by proposal.hip (score order) and fps.hip (Morton order).
#pragma once
#include "common.h"

typedef unsigned long long u64;

__device__ __forceinline__ int lds_phys(int i) { return i + (i >> 4); }     // one pad word per 16 keys

template <class K>
__device__ __forceinline__ void cswap(K& a, K& b, bool up) {
    const bool gt = a > b;
    if (gt == up) { K t = a; a = b; b = t; }
}

// strides 8,4,2,1 of one merge stage on the 16 keys a thread owns (all 16 share the direction once k >= 32)
template <class K>
__device__ __forceinline__ void merge16(K (&v)[16], bool up) {
#pragma unroll
    for (int j = 8; j >= 1; j >>= 1)
#pragma unroll
        for (int e = 0; e < 16; e++)
            if ((e & j) == 0) cswap(v[e], v[e | j], up);
}

// One compare-exchange pass of stride j >= 16 inside merge stage k: the thread's 8 pairs are p = q * nact + t, element
// i = ((p & ~(j-1)) << 1) | (p & (j-1)).  While j <= nact (all but the last few passes of the largest stages) the low bits of p are t's,
// so i = i0(t, j) + q * 2 nact and, 2 nact being a multiple of 16, its padded address is lds_phys(i0) + q * (2 nact + nact / 8): one
// address computation per pass instead of eight (round 6: a pass was ~160 instructions per thread of mostly index arithmetic -- the
// sort is issue-bound, not LDS-bound: halving the keys had bought 11 %).  All 16 keys are read before any is written back.
template <class K>
__device__ __forceinline__ void lds_pass(K* keys, int nact, int t, int j, int k) {
    K a[8], c[8];
    if (j <= nact && nact >= 8) {
        const int i0 = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int D = 2 * nact, DP = D + (D >> 4);
        K* pa = keys + lds_phys(i0);
        K* pb = keys + lds_phys(i0 + j);
#pragma unroll
        for (int q = 0; q < 8; q++) { a[q] = pa[q * DP]; c[q] = pb[q * DP]; }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const bool up = ((i0 + q * D) & k) == 0;
            if ((a[q] > c[q]) == up) { pa[q * DP] = c[q]; pb[q * DP] = a[q]; }
        }
    } else {
        int ia[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int p = q * nact + t;
            ia[q] = ((p & ~(j - 1)) << 1) | (p & (j - 1));
            a[q] = keys[lds_phys(ia[q])]; c[q] = keys[lds_phys(ia[q] + j)];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const bool up = (ia[q] & k) == 0;
            if ((a[q] > c[q]) == up) { keys[lds_phys(ia[q])] = c[q]; keys[lds_phys(ia[q] + j)] = a[q]; }
        }
    }
}

// bytes of LDS `keys` needs for Npad keys (Npad a power of two >= 16)
static inline size_t lds_sort_bytes(int Npad, size_t key_bytes = sizeof(u64)) { return ((size_t)(Npad + (Npad >> 4)) + 1) * key_bytes; }

// Sort Npad keys ascending.  Thread t < Npad/16 owns positions 16t .. 16t+15: it passes their keys in v and receives the
// sorted keys of the same positions back in v (the sorted sequence is also left in LDS, lds_phys layout).  EVERY thread of
// the workgroup must call this (barriers inside); threads with t >= Npad/16 only take part in the barriers.
template <class K>
__device__ __forceinline__ void block_sort16(K (&v)[16], K* keys, int Npad, int t) {
    const int nact = Npad >> 4;
    const bool active = t < nact;
    if (active) {
        // stages k = 2..16 entirely in registers; element i sorts ascending when (i & k) == 0
#pragma unroll
        for (int k = 2; k <= 16; k <<= 1)
#pragma unroll
            for (int j = k >> 1; j >= 1; j >>= 1)
#pragma unroll
                for (int e = 0; e < 16; e++)
                    if ((e & j) == 0) cswap(v[e], v[e | j], k < 16 ? ((e & k) == 0) : ((t & 1) == 0));
#pragma unroll
        for (int e = 0; e < 16; e++) keys[lds_phys(t * 16 + e)] = v[e];
    }
    __syncthreads();
    for (int k = 32; k <= Npad; k <<= 1) {
        for (int j = k >> 1; j >= 16; j >>= 1) {        // strides >= 16 through LDS: 8 pairs per owning thread
            if (active) lds_pass(keys, nact, t, j, k);
            __syncthreads();
        }
        if (active) {                                   // strides 8..1 in registers
#pragma unroll
            for (int e = 0; e < 16; e++) v[e] = keys[lds_phys(t * 16 + e)];
            merge16(v, ((t * 16) & k) == 0);
#pragma unroll
            for (int e = 0; e < 16; e++) keys[lds_phys(t * 16 + e)] = v[e];
        }
        __syncthreads();
    }
}

// One merge stage on Npad (<= 16384) keys that form a bitonic sequence: strides Npad/2 .. 16 through LDS, 8 .. 1 in registers,
// ascending.  Same calling convention as block_sort16 (v holds the thread's 16 keys on entry and on exit).
__device__ __forceinline__ void block_merge16(u64 (&v)[16], u64* keys, int Npad, int t) {
    const int nact = Npad >> 4;
    const bool active = t < nact;
    if (active) {
#pragma unroll
        for (int e = 0; e < 16; e++) keys[lds_phys(t * 16 + e)] = v[e];
    }
    __syncthreads();
    for (int j = Npad >> 1; j >= 16; j >>= 1) {
        if (active) lds_pass(keys, nact, t, j, 1 << 30);          // (i & 2^30) == 0: ascending everywhere
        __syncthreads();
    }
    if (active) {
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = keys[lds_phys(t * 16 + e)];
        merge16(v, true);
    }
}
