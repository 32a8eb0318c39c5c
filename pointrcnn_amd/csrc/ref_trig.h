// ref_trig.h -- float sinf / cosf / atan2f with the arithmetic of the reference's HOST libm, for device code.
//
// The reference's iou3d / roipool3d sources call cos(), sin(), atan2() on floats (lib/utils/iou3d/src/iou3d_kernel.cu:56,
// 104-106,135-136, lib/utils/roipool3d/src/roipool3d.cpp:89, roipool3d_kernel.cu:22); the pin of this repository is those sources
// compiled for the host (the test pin), i.e. glibc's float routines.  A device libm differs from glibc by ulps -- 1.3 % of all
// angles give a different cosf -- which flips point-in-box and NMS decisions that sit within rounding of their threshold.  These
// are restatements of glibc 2.35's algorithms that return the SAME BITS:
//   sinf / cosf  (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h: ARM optimized-routines): argument reduction by pi/2 and a
//                degree-7/8 polynomial, everything in double, the multiply-adds FUSED exactly where the x86-64 FMA build of glibc
//                (the ifunc variant every FMA-capable host selects) fuses them.  Verified exhaustively on the build host: all
//                2 246 049 792 floats with |x| < 120 give bit-identical results for both functions (tools/ref_trig_check.c);
//                |x| >= 120 (no box angle) falls back to the double routine rounded once.
//   atan2f       (e_atan2f.c + s_atanf.c: the fdlibm float routines, plain float arithmetic, no fused operations): verified
//                exhaustively for atanf (all positive floats) and on 4e8 random (y, x) pairs for atan2f.
// Polynomial coefficients are the published ones (read back from the host's libm.so.6 for the check).  Everything here is
// individually rounded IEEE arithmetic: the library is built with -ffp-contract=off and correctly rounded float division.
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#define RT_FN __host__ __device__ __forceinline__
#else
#define RT_FN static inline
#endif

typedef struct RtSinCos { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } RtSinCos;

RT_FN uint32_t rt_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
RT_FN float rt_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
RT_FN uint32_t rt_abstop12(float x) { return (rt_asuint(x) >> 20) & 0x7ff; }

// table 0: quadrants where the cosine polynomial is positive; table 1 = the same with the cosine coefficients negated
RT_FN void rt_table(int neg, RtSinCos* tp) {
    RtSinCos t;
    const double sg = neg ? -1.0 : 1.0;
    t.sign[0] = 1.0; t.sign[1] = -1.0; t.sign[2] = -1.0; t.sign[3] = 1.0;
    t.hpi_inv = 0x1.45F306DC9C883p+23; t.hpi = 0x1.921FB54442D18p0;
    t.c0 = sg * 0x1p0; t.c1 = sg * -0x1.ffffffd0c621cp-2; t.c2 = sg * 0x1.55553e1068f19p-5; t.c3 = sg * -0x1.6c087e89a359dp-10;
    t.c4 = sg * 0x1.99343027bf8c3p-16;
    t.s1 = -0x1.555545995a603p-3; t.s2 = 0x1.1107605230bc4p-7; t.s3 = -0x1.994eb3774cf24p-13;
    *tp = t;
}

RT_FN float rt_poly(double x, double x2, const RtSinCos* pp, int n) {
    const RtSinCos p = *pp;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = fma(x2, p.s3, p.s2);
        const double x7 = x3 * x2;
        const double s = fma(x3, p.s1, x);
        return (float)fma(x7, s1, s);
    }
    const double x4 = x2 * x2;
    const double c2 = fma(x2, p.c4, p.c3);
    const double c1 = fma(x2, p.c1, p.c0);
    const double x6 = x4 * x2;
    const double c = fma(x4, p.c2, c1);
    return (float)fma(x6, c2, c);
}

RT_FN double rt_reduce_fast(double x, const RtSinCos* pp, int* np) {
    const RtSinCos p = *pp;
    const double r = x * p.hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return fma(-(double)n, p.hpi, x);
}

RT_FN float prcnn_ref_sinf(float y) {
    if (!(rt_abstop12(y) < rt_abstop12(120.0f))) return (float)sin((double)y);
    double x = y;
    RtSinCos p;
    if (rt_abstop12(y) < rt_abstop12(0x1.921FB6p-1f)) {
        if (rt_abstop12(y) < rt_abstop12(0x1p-12f)) return y;
        rt_table(0, &p);
        return rt_poly(x, x * x, &p, 0);
    }
    rt_table(0, &p);
    int n;
    x = rt_reduce_fast(x, &p, &n);
    const double s = ((n + 1) & 2) ? -1.0 : 1.0;          // sign[n & 3] of {1, -1, -1, 1} without an indexed table (a dynamically indexed local array is scratch memory on the GPU)
    if (n & 2) rt_table(1, &p);
    return rt_poly(x * s, x * x, &p, n);
}

RT_FN float prcnn_ref_cosf(float y) {
    if (!(rt_abstop12(y) < rt_abstop12(120.0f))) return (float)cos((double)y);
    double x = y;
    RtSinCos p;
    if (rt_abstop12(y) < rt_abstop12(0x1.921FB6p-1f)) {
        if (rt_abstop12(y) < rt_abstop12(0x1p-12f)) return 1.0f;
        rt_table(0, &p);
        return rt_poly(x, x * x, &p, 1);
    }
    rt_table(0, &p);
    int n;
    x = rt_reduce_fast(x, &p, &n);
    const double s = ((n + 2) & 2) ? -1.0 : 1.0;          // sign[(n + 1) & 3]
    if ((n + 1) & 2) rt_table(1, &p);
    return rt_poly(x * s, x * x, &p, n ^ 1);
}

RT_FN float rt_atanf(float x) {
    const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
    const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
    const float aT[11] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f,
                           -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f,
                           1.6285819933e-02f };
    const float one = 1.0f, huge = 1.0e30f;
    const int32_t hx = (int32_t)rt_asuint(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {
        if (ix < 0x31000000) { if (huge + x > one) return x; }
        id = -1;
    } else {
        x = rt_asfloat((uint32_t)ix);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
            else { id = 1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float hi = id == 0 ? atanhi[0] : id == 1 ? atanhi[1] : id == 2 ? atanhi[2] : atanhi[3];          // (selects, not an indexed table)
    const float lo = id == 0 ? atanlo[0] : id == 1 ? atanlo[1] : id == 2 ? atanlo[2] : atanlo[3];
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}

RT_FN float prcnn_ref_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)rt_asuint(x), ix = hx & 0x7fffffff, hy = (int32_t)rt_asuint(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return rt_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = rt_atanf(rt_asfloat(rt_asuint(y / x) & 0x7fffffffu));
    if (m == 0) return z;
    if (m == 1) return rt_asfloat(rt_asuint(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}
