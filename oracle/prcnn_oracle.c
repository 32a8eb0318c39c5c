/*
 * prcnn_oracle.c -- CPU ORACLE for the PointRCNN point-ops hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (pointrcnn_amd/) never imports, links or calls anything here.
 *
 * Every function restates, single-threaded and in plain C, the algorithm of
 * one operator on the reference's hot path, citing the reference file:line it
 * follows (paths relative to the reference checkout).  Build (see Makefile):
 *     gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC prcnn_oracle.c -lm
 * -ffp-contract=off matters: every fp32 product/sum below is individually
 * rounded (no FMA), which is the arithmetic contract the HIP kernels follow.
 *
 * PARITY PINNING STATUS
 *   roipool3d / pts_in_boxes3d : PINNED -- checked bit-for-bit against the
 *       reference's own lib/utils/roipool3d/src/roipool3d.cpp compiled in
 *       place (oracle/_ref, tests/test_oracle.py) and via golden
 *       fixtures generated from it (tests/golden/).
 *   iou3d (overlap / iou / nms): PINNED -- checked against the reference's
 *       lib/utils/iou3d/src/iou3d_kernel.cu device functions compiled for the
 *       host (oracle/_ref, trig_mode 0 == bit-exact) and golden fixtures.
 *   PointNet++ ops (fps, ball_query, group, gather, three_nn,
 *       three_interpolate, per-point MLP): PARITY UNPINNED -- the reference
 *       vendors them as an EMPTY git submodule (.gitmodules:1-4,
 *       sshaoshuai/Pointnet2.PyTorch, SHA unrecoverable).  The functions below
 *       restate the published algorithm under the canonical rules of SURVEY.md
 *       Appendix A, constrained by the reference call sites
 *       (lib/net/pointnet2_msg.py:27-34,44,61,66-68; lib/net/rcnn_net.py:33-41).
 *
 * trig_mode (roipool3d / iou3d):
 *   0 = "reference libm": cosf/sinf/atan2f exactly as the reference source
 *       spells them -- used to pin this restatement against oracle/_ref.
 *   1 = "canonical": cos/sin evaluated in double and rounded once to float,
 *       polygon vertices ordered by a division-only monotone surrogate of
 *       atan2 -- the contract the HIP kernels implement bit-for-bit (device
 *       libm differs from glibc by ULPs, so mode 0 cannot be bit-exact on a
 *       GPU; mode 1 can).  Mode 0 vs mode 1 agree to ~1e-6 relative.
 *   2 = "reference arithmetic, restated" (round 3; what the HIP kernels of roipool3d / iou3d / NMS / labels / RoI sampling
 *       implement now): pointrcnn_amd/csrc/ref_trig.h -- glibc's float sinf / cosf (double polynomial, FMA build) and atan2f
 *       (fdlibm float) restated operation by operation, bit-identical to the host libm for every float |x| < 120 (exhaustive,
 *       tools/ref_trig_check.c).  Mode 2 == mode 0 on any FMA-capable glibc 2.35 host; unlike mode 0 it does not depend on
 *       which libm the test machine has.
 */
#include <math.h>
#include <stdint.h>
#include "../pointrcnn_amd/csrc/ref_trig.h"   /* trig_mode 2: the kernels' restatement of glibc's sinf / cosf / atan2f */
#include <stdlib.h>
#include <string.h>

#define PRCNN_EXPORT __attribute__((visibility("default")))

static inline float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
    /* three rounded products summed left to right, no FMA. */
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float s = xx + yy;
    return s + zz;
}

/* ------------------------------------------------------------------ *
 * A.1 furthest_point_sample  [UPSTREAM pointnet2 sampling kernel; call
 * sites lib/net/pointnet2_msg.py:27-34 via PointnetSAModuleMSG]
 * xyz (B,N,3) f32 -> idx (B,npoint) i32.  tmp (B,N) scratch or NULL.
 * start index 0, temp=1e10, tie -> lowest point index.
 * ------------------------------------------------------------------ */
PRCNN_EXPORT void prcnn_cpu_fps(const float* xyz, int B, int N, int npoint, float* tmp, int* idx) {
    float* own = NULL;
    if (!tmp) { own = (float*)malloc(sizeof(float) * (size_t)B * N); tmp = own; }
    for (int b = 0; b < B; b++) {
        const float* p = xyz + (size_t)b * N * 3;
        float* t = tmp + (size_t)b * N;
        int* o = idx + (size_t)b * npoint;
        for (int k = 0; k < N; k++) t[k] = 1e10f;
        if (npoint <= 0) continue;
        int old = 0;
        o[0] = 0;
        for (int j = 1; j < npoint; j++) {
            float x0 = p[old * 3], y0 = p[old * 3 + 1], z0 = p[old * 3 + 2];
            float best = -1.0f;
            int besti = 0;
            for (int k = 0; k < N; k++) {
                float d = sqdist3(p[k * 3], p[k * 3 + 1], p[k * 3 + 2], x0, y0, z0);
                float v = d < t[k] ? d : t[k];
                t[k] = v;
                if (v > best) { best = v; besti = k; }
            }
            o[j] = besti;
            old = besti;
        }
    }
    free(own);
}

/* A.1 with the UPSTREAM tie order: the upstream kernel's T threads (T = min(1024, largest power of
 * two <= N)) each scan k = t, t+T, ... keeping their first maximum, the tree reduction keeps the lower thread on ties:
 * among equal maxima the winner is argmin (k mod T, k).  Emulated literally: per-thread best, then threads in order. */
PRCNN_EXPORT void prcnn_cpu_fps_upstream(const float* xyz, int B, int N, int npoint, int* idx) {
    int T = 1;
    while (T * 2 <= N && T < 1024) T <<= 1;
    float* t = (float*)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; b++) {
        const float* p = xyz + (size_t)b * N * 3;
        int* o = idx + (size_t)b * npoint;
        for (int k = 0; k < N; k++) t[k] = 1e10f;
        if (npoint <= 0) continue;
        int old = 0;
        o[0] = 0;
        for (int j = 1; j < npoint; j++) {
            float x0 = p[old * 3], y0 = p[old * 3 + 1], z0 = p[old * 3 + 2];
            float best = -1.0f;
            int besti = 0;
            for (int th = 0; th < T; th++) {
                float tb = -1.0f;
                int ti = 0;
                for (int k = th; k < N; k += T) {
                    float d = sqdist3(p[k * 3], p[k * 3 + 1], p[k * 3 + 2], x0, y0, z0);
                    float v = d < t[k] ? d : t[k];
                    t[k] = v;
                    if (v > tb) { tb = v; ti = k; }
                }
                if (tb > best) { best = tb; besti = ti; }
            }
            o[j] = besti;
            old = besti;
        }
    }
    free(t);
}

/* ------------------------------------------------------------------ *
 * "UPSTREAM ARITHMETIC" comparison mode (round 6; PARITY UNPINNED either way: the upstream source is absent, .gitmodules:1-4).
 * The upstream kernels write the squared distance as  (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)  and are built by nvcc,
 * whose default (-fmad=true) contracts a*a + b*b + c*c into  fma(c,c, fma(b,b, a*a)):  one rounding after dx*dx, then two fused
 * steps -- not the three rounded products of the canonical contract above.  arith 0 = canonical (sqdist3), arith 1 = that contracted
 * form.  The two differ in the last bit of some distances; indices differ only where that bit decides a comparison (a near-tie of two
 * running min-distances in FPS, a point within an ulp of the ball radius, the 3rd / 4th neighbour at nearly the same distance).
 * tests/test_gpu_arith_modes.py holds the kernels to these functions bit for bit in both modes and tools/arith_disagreement.py counts
 * how often the modes disagree on the bench clouds.
 * ------------------------------------------------------------------ */
static inline float sqdist3_mode(int arith, float ax, float ay, float az, float bx, float by, float bz) {
    if (!arith) return sqdist3(ax, ay, az, bx, by, bz);
    float dx = ax - bx, dy = ay - by, dz = az - bz;
    float xx = dx * dx;
    return fmaf(dz, dz, fmaf(dy, dy, xx));       /* libm fmaf: correctly rounded whatever the host (-ffp-contract=off never fuses by itself) */
}

/* A.1 with selectable tie order (0 canonical: lowest point index; 1 upstream: argmin (k mod T, k), see prcnn_cpu_fps_upstream) and
 * distance arithmetic */
PRCNN_EXPORT void prcnn_cpu_fps_mode(const float* xyz, int B, int N, int npoint, int order, int arith, int* idx) {
    int T = 1;
    if (order) while (T * 2 <= N && T < 1024) T <<= 1;
    float* t = (float*)malloc(sizeof(float) * (size_t)N);
    for (int b = 0; b < B; b++) {
        const float* p = xyz + (size_t)b * N * 3;
        int* o = idx + (size_t)b * npoint;
        for (int k = 0; k < N; k++) t[k] = 1e10f;
        if (npoint <= 0) continue;
        int old = 0;
        o[0] = 0;
        for (int j = 1; j < npoint; j++) {
            float x0 = p[old * 3], y0 = p[old * 3 + 1], z0 = p[old * 3 + 2];
            float best = -1.0f;
            int besti = 0;
            for (int th = 0; th < T; th++) {            /* T == 1: one pass in index order == the canonical rule */
                float tb = -1.0f;
                int ti = 0;
                for (int k = th; k < N; k += T) {
                    float d = sqdist3_mode(arith, p[k * 3], p[k * 3 + 1], p[k * 3 + 2], x0, y0, z0);
                    float v = d < t[k] ? d : t[k];
                    t[k] = v;
                    if (v > tb) { tb = v; ti = k; }
                }
                if (tb > best) { best = tb; besti = ti; }
            }
            o[j] = besti;
            old = besti;
        }
    }
    free(t);
}

PRCNN_EXPORT void prcnn_cpu_ball_query_arith(const float* xyz, const float* new_xyz, int B, int N, int M, float radius, int nsample,
                                             int arith, int* idx) {
    float r2 = radius * radius;
    for (int b = 0; b < B; b++) {
        const float* p = xyz + (size_t)b * N * 3;
        for (int m = 0; m < M; m++) {
            const float* q = new_xyz + ((size_t)b * M + m) * 3;
            int* o = idx + ((size_t)b * M + m) * nsample;
            for (int s = 0; s < nsample; s++) o[s] = 0;
            int cnt = 0;
            for (int k = 0; k < N && cnt < nsample; k++) {
                float d2 = sqdist3_mode(arith, q[0], q[1], q[2], p[k * 3], p[k * 3 + 1], p[k * 3 + 2]);
                if (d2 < r2) {
                    if (cnt == 0)
                        for (int s = 0; s < nsample; s++) o[s] = k;
                    o[cnt] = k;
                    cnt++;
                }
            }
        }
    }
}

PRCNN_EXPORT void prcnn_cpu_three_nn_arith(const float* unknown, const float* known, int B, int n, int m, int arith, float* dist2,
                                           int* idx) {
    for (int b = 0; b < B; b++) {
        const float* kn = known + (size_t)b * m * 3;
        for (int i = 0; i < n; i++) {
            const float* u = unknown + ((size_t)b * n + i) * 3;
            float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
            int i1 = 0, i2 = 0, i3 = 0;
            for (int k = 0; k < m; k++) {
                float d = sqdist3_mode(arith, u[0], u[1], u[2], kn[k * 3], kn[k * 3 + 1], kn[k * 3 + 2]);
                if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
                else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
                else if (d < b3) { b3 = d; i3 = k; }
            }
            float* od = dist2 + ((size_t)b * n + i) * 3;
            int* oi = idx + ((size_t)b * n + i) * 3;
            od[0] = b1; od[1] = b2; od[2] = b3;
            oi[0] = i1; oi[1] = i2; oi[2] = i3;
        }
    }
}

/* A.2 gather_operation: out[b,c,m] = feat[b,c,idx[b,m]] */
PRCNN_EXPORT void prcnn_cpu_gather(const float* feat, const int* idx, int B, int C, int N, int M, float* out) {
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int m = 0; m < M; m++)
                out[((size_t)b * C + c) * M + m] = feat[((size_t)b * C + c) * N + idx[(size_t)b * M + m]];
}

/* A.2 backward: scatter-add of grad_out (B,C,M) into grad_feat (B,C,N) (pre-zeroed by caller) */
PRCNN_EXPORT void prcnn_cpu_gather_grad(const float* grad_out, const int* idx, int B, int C, int N, int M, float* grad_feat) {
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int m = 0; m < M; m++)
                grad_feat[((size_t)b * C + c) * N + idx[(size_t)b * M + m]] += grad_out[((size_t)b * C + c) * M + m];
}

/* A.3 ball_query: first <=nsample points with d2 < r2 in ascending index order,
 * padded with the first hit; no hit -> zeros. */
PRCNN_EXPORT void prcnn_cpu_ball_query(const float* xyz, const float* new_xyz, int B, int N, int M,
                                       float radius, int nsample, int* idx) {
    float r2 = radius * radius;
    for (int b = 0; b < B; b++) {
        const float* p = xyz + (size_t)b * N * 3;
        for (int m = 0; m < M; m++) {
            const float* q = new_xyz + ((size_t)b * M + m) * 3;
            int* o = idx + ((size_t)b * M + m) * nsample;
            for (int s = 0; s < nsample; s++) o[s] = 0;
            int cnt = 0;
            for (int k = 0; k < N && cnt < nsample; k++) {
                float d2 = sqdist3(q[0], q[1], q[2], p[k * 3], p[k * 3 + 1], p[k * 3 + 2]);
                if (d2 < r2) {
                    if (cnt == 0)
                        for (int s = 0; s < nsample; s++) o[s] = k;
                    o[cnt] = k;
                    cnt++;
                }
            }
        }
    }
}

/* A.4 grouping_operation: out[b,c,m,s] = feat[b,c,idx[b,m,s]] */
PRCNN_EXPORT void prcnn_cpu_group(const float* feat, const int* idx, int B, int C, int N, int M, int ns, float* out) {
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int m = 0; m < M; m++)
                for (int s = 0; s < ns; s++)
                    out[(((size_t)b * C + c) * M + m) * ns + s] =
                        feat[((size_t)b * C + c) * N + idx[((size_t)b * M + m) * ns + s]];
}

PRCNN_EXPORT void prcnn_cpu_group_grad(const float* grad_out, const int* idx, int B, int C, int N, int M, int ns,
                                       float* grad_feat) {
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int m = 0; m < M; m++)
                for (int s = 0; s < ns; s++)
                    grad_feat[((size_t)b * C + c) * N + idx[((size_t)b * M + m) * ns + s]] +=
                        grad_out[(((size_t)b * C + c) * M + m) * ns + s];
}

/* A.5 three_nn: 3 smallest squared distances, strict '<' insertion (ties keep earlier k).
 * Returns SQUARED distances (the Python wrapper takes the sqrt, as upstream does). */
PRCNN_EXPORT void prcnn_cpu_three_nn(const float* unknown, const float* known, int B, int n, int m,
                                     float* dist2, int* idx) {
    for (int b = 0; b < B; b++) {
        const float* kn = known + (size_t)b * m * 3;
        for (int i = 0; i < n; i++) {
            const float* u = unknown + ((size_t)b * n + i) * 3;
            float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
            int i1 = 0, i2 = 0, i3 = 0;
            for (int k = 0; k < m; k++) {
                float d = sqdist3(u[0], u[1], u[2], kn[k * 3], kn[k * 3 + 1], kn[k * 3 + 2]);
                if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
                else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
                else if (d < b3) { b3 = d; i3 = k; }
            }
            float* od = dist2 + ((size_t)b * n + i) * 3;
            int* oi = idx + ((size_t)b * n + i) * 3;
            od[0] = b1; od[1] = b2; od[2] = b3;
            oi[0] = i1; oi[1] = i2; oi[2] = i3;
        }
    }
}

/* A.6 three_interpolate: out[b,c,i] = (w0*f[i0] + w1*f[i1]) + w2*f[i2] */
PRCNN_EXPORT void prcnn_cpu_three_interp(const float* feat, const int* idx, const float* w, int B, int C, int m, int n,
                                         float* out) {
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++) {
            const float* f = feat + ((size_t)b * C + c) * m;
            for (int i = 0; i < n; i++) {
                const int* ii = idx + ((size_t)b * n + i) * 3;
                const float* ww = w + ((size_t)b * n + i) * 3;
                float a0 = ww[0] * f[ii[0]], a1 = ww[1] * f[ii[1]], a2 = ww[2] * f[ii[2]];
                float s = a0 + a1;
                out[((size_t)b * C + c) * n + i] = s + a2;
            }
        }
}

PRCNN_EXPORT void prcnn_cpu_three_interp_grad(const float* grad_out, const int* idx, const float* w, int B, int C, int n,
                                              int m, float* grad_feat) {
    for (int b = 0; b < B; b++)
        for (int c = 0; c < C; c++)
            for (int i = 0; i < n; i++) {
                const int* ii = idx + ((size_t)b * n + i) * 3;
                const float* ww = w + ((size_t)b * n + i) * 3;
                float g = grad_out[((size_t)b * C + c) * n + i];
                float* gf = grad_feat + ((size_t)b * C + c) * m;
                gf[ii[0]] += g * ww[0];
                gf[ii[1]] += g * ww[1];
                gf[ii[2]] += g * ww[2];
            }
}

/* FP-module interpolation weights: w = 1/(sqrt(d2)+1e-8), normalised. */
PRCNN_EXPORT void prcnn_cpu_three_weights(const float* dist2, int count, float* w) {
    for (int i = 0; i < count; i++) {
        float r0 = 1.0f / (sqrtf(dist2[i * 3]) + 1e-8f);
        float r1 = 1.0f / (sqrtf(dist2[i * 3 + 1]) + 1e-8f);
        float r2 = 1.0f / (sqrtf(dist2[i * 3 + 2]) + 1e-8f);
        float s = r0 + r1;
        s = s + r2;
        w[i * 3] = r0 / s; w[i * 3 + 1] = r1 / s; w[i * 3 + 2] = r2 / s;
    }
}

/* Per-point MLP layer on row-major rows (channels-last): out[r,n] = act(sum_k a[r,k]*w[n,k] + bias[n]).
 * Accumulated in DOUBLE: this is the "true value" the fp32-MFMA kernel is compared to with a
 * tolerance (the kernel sums in a different k order, so bit-exactness is not the contract here). */
PRCNN_EXPORT void prcnn_cpu_linear_rows(const float* a, const float* w, const float* bias, int R, int K, int Nout,
                                        int relu, float* out) {
    for (int r = 0; r < R; r++)
        for (int n = 0; n < Nout; n++) {
            double acc = bias ? (double)bias[n] : 0.0;
            const float* ar = a + (size_t)r * K;
            const float* wr = w + (size_t)n * K;
            for (int k = 0; k < K; k++) acc += (double)ar[k] * (double)wr[k];
            float v = (float)acc;
            if (relu && v < 0.0f) v = 0.0f;
            out[(size_t)r * Nout + n] = v;
        }
}

/* ------------------------------------------------------------------ *
 * roipool3d  (lib/utils/roipool3d/src/roipool3d.cpp:82-195 is the
 * authoritative CPU code; GPU twin roipool3d_kernel.cu:14-28,97-194)
 * ------------------------------------------------------------------ */
static void box_trig(float angle, int trig_mode, float* cosa, float* sina) {
    if (trig_mode == 0) { *cosa = cosf(angle); *sina = sinf(angle); }      /* roipool3d.cpp:89 */
    else if (trig_mode == 2) { *cosa = prcnn_ref_cosf(angle); *sina = prcnn_ref_sinf(angle); }
    else { *cosa = (float)cos((double)angle); *sina = (float)sin((double)angle); }
}

/* roipool3d.cpp:82-95 pt_in_box3d_cpu */
static int pt_in_box3d_gate(float x, float y, float z, float cx, float bottom_y, float cz, float h, float w, float l,
                            float cosa, float sina, float max_dis) {
    float x_rot, z_rot, cy;
    cy = (float)((double)bottom_y - (double)h / 2.0);
    if ((fabsf(x - cx) > max_dis) || ((double)fabsf(y - cy) > (double)h / 2.0) || (fabsf(z - cz) > max_dis)) return 0;
    {
        float dx = x - cx, dz = z - cz;
        float a = dx * cosa, b = dz * (-sina);
        x_rot = a + b;
        a = dx * sina; b = dz * cosa;
        z_rot = a + b;
    }
    return ((double)x_rot >= -(double)l / 2.0) & ((double)x_rot <= (double)l / 2.0) &
           ((double)z_rot >= -(double)w / 2.0) & ((double)z_rot <= (double)w / 2.0);
}

static int pt_in_box3d(float x, float y, float z, float cx, float bottom_y, float cz, float h, float w, float l,
                       float cosa, float sina) {
    return pt_in_box3d_gate(x, y, z, cx, bottom_y, cz, h, w, l, cosa, sina, 10.0f);
}

/* roipool3d.cpp:97-125 pts_in_boxes3d_cpu: flags (M,N) int64 */
PRCNN_EXPORT void prcnn_cpu_pts_in_boxes3d(const float* pts, const float* boxes3d, int N, int M, int trig_mode,
                                           int64_t* flags) {
    for (int i = 0; i < M; i++) {
        const float* bx = boxes3d + (size_t)i * 7;
        float ca, sa;
        box_trig(bx[6], trig_mode, &ca, &sa);
        for (int j = 0; j < N; j++)
            flags[(size_t)i * N + j] =
                pt_in_box3d(pts[j * 3], pts[j * 3 + 1], pts[j * 3 + 2], bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], ca, sa);
    }
}

/* The point work of KittiRCNNDataset.apply_gt_aug_to_one_scene (lib/datasets/kitti_rcnn_dataset.py:484-489: one
 * pts_in_boxes3d_cpu scan per accepted object with h += 2, keep flag cleared; :501-507: pts_rect[src_pts_flag == 1] then
 * np.concatenate with the pasted points), per scene of a batch.  Rows past out_count are zero. */
PRCNN_EXPORT void prcnn_cpu_gt_aug_edit(const float* pts, const float* inten, const int* num_pts, const float* boxes, const int* num_boxes,
                                        float extra_h, const float* new_pts, const float* new_inten, const int* num_new, int B, int N,
                                        int K, int P, int trig_mode, float* out_pts, float* out_inten, int* out_count, int* removed) {
    for (int b = 0; b < B; b++) {
        const int n = num_pts ? (num_pts[b] < 0 ? 0 : (num_pts[b] < N ? num_pts[b] : N)) : N;
        const int k = num_boxes ? (num_boxes[b] < 0 ? 0 : (num_boxes[b] < K ? num_boxes[b] : K)) : K;
        const int np = num_new ? (num_new[b] < 0 ? 0 : (num_new[b] < P ? num_new[b] : P)) : P;
        float* op = out_pts + (size_t)b * (N + P) * 3;
        float* oi = out_inten ? out_inten + (size_t)b * (N + P) : 0;
        int kept = 0;
        for (int i = 0; i < N; i++)
            if (removed) removed[(size_t)b * N + i] = 0;
        for (int i = 0; i < n; i++) {
            const float* p = pts + ((size_t)b * N + i) * 3;
            int inside = 0;
            for (int q = 0; q < k; q++) {
                const float* bx = boxes + ((size_t)b * K + q) * 7;
                float ca, sa;
                const float h = bx[3] + extra_h;
                box_trig(bx[6], trig_mode, &ca, &sa);
                inside |= pt_in_box3d(p[0], p[1], p[2], bx[0], bx[1], bx[2], h, bx[4], bx[5], ca, sa);
            }
            if (removed) removed[(size_t)b * N + i] = inside;
            if (!inside) {
                op[kept * 3] = p[0]; op[kept * 3 + 1] = p[1]; op[kept * 3 + 2] = p[2];
                if (oi) oi[kept] = inten[(size_t)b * N + i];
                kept++;
            }
        }
        for (int i = 0; i < np; i++) {
            const float* q = new_pts + ((size_t)b * P + i) * 3;
            op[(kept + i) * 3] = q[0]; op[(kept + i) * 3 + 1] = q[1]; op[(kept + i) * 3 + 2] = q[2];
            if (oi) oi[kept + i] = new_inten[(size_t)b * P + i];
        }
        for (int i = kept + np; i < N + P; i++) {
            op[i * 3] = op[i * 3 + 1] = op[i * 3 + 2] = 0.f;
            if (oi) oi[i] = 0.f;
        }
        out_count[b] = kept + np;
    }
}

/* KittiRCNNDataset.generate_rpn_training_labels (lib/datasets/kitti_rcnn_dataset.py:365-394) with the analytic in-box
 * test above in place of the scipy Delaunay hull test: per frame, GT boxes applied in order; cls (B,N) int32 in {-1,0,1},
 * reg (B,N,7) [dx, dy, dz, h, w, l, ry].  enlarge_box3d (kitti_utils.py:150-160): h,w,l += 2*extra, y += extra. */
PRCNN_EXPORT void prcnn_cpu_rpn_labels(const float* pts, const float* gt, const int* num_gt, int B, int N, int G, float extra,
                                       int trig_mode, int* cls, float* reg) {
    for (int b = 0; b < B; b++) {
        const int g = num_gt ? (num_gt[b] < G ? (num_gt[b] < 0 ? 0 : num_gt[b]) : G) : G;
        for (int n = 0; n < N; n++) {
            const float* p = pts + ((size_t)b * N + n) * 3;
            int c = 0;
            float r[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < g; k++) {
                const float* bx = gt + ((size_t)b * G + k) * 7;
                float ca, sa;
                box_trig(bx[6], trig_mode, &ca, &sa);
                const float e2 = extra * 2;
                /* the reference tests against the box's convex hull (kitti_utils.in_hull): no 10 m centre-distance gate */
                const int fg = pt_in_box3d_gate(p[0], p[1], p[2], bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], ca, sa, INFINITY);
                const int en = pt_in_box3d_gate(p[0], p[1], p[2], bx[0], bx[1] + extra, bx[2], bx[3] + e2, bx[4] + e2, bx[5] + e2, ca, sa, INFINITY);
                if (fg) {
                    const float cy = bx[1] - bx[3] / 2;
                    c = 1;
                    r[0] = bx[0] - p[0]; r[1] = cy - p[1]; r[2] = bx[2] - p[2];
                    r[3] = bx[3]; r[4] = bx[4]; r[5] = bx[5]; r[6] = bx[6];
                }
                if (fg != en) c = -1;
            }
            cls[(size_t)b * N + n] = c;
            memcpy(reg + ((size_t)b * N + n) * 7, r, sizeof(r));
        }
    }
}

/* Batched pooling with the GPU op's output layout (roipool3d_kernel.cu:97-194 ==
 * roipool3d.cpp:127-195 semantics): out (B,M,S,3+C) f32 pre-zeroed, empty (B,M) i32.
 * boxes are ALREADY enlarged (roipool3d_utils.py:19). */
PRCNN_EXPORT void prcnn_cpu_roipool3d(const float* xyz, const float* boxes3d, const float* feat, int B, int N, int M,
                                      int C, int S, int trig_mode, float* out, int* empty) {
    int* sel = (int*)malloc(sizeof(int) * (size_t)(S > 0 ? S : 1));
    for (int b = 0; b < B; b++)
        for (int m = 0; m < M; m++) {
            const float* bx = boxes3d + ((size_t)b * M + m) * 7;
            const float* p = xyz + (size_t)b * N * 3;
            float ca, sa;
            box_trig(bx[6], trig_mode, &ca, &sa);
            int cnt = 0;
            for (int k = 0; k < N && cnt < S; k++)
                if (pt_in_box3d(p[k * 3], p[k * 3 + 1], p[k * 3 + 2], bx[0], bx[1], bx[2], bx[3], bx[4], bx[5], ca, sa))
                    sel[cnt++] = k;
            float* o = out + ((size_t)b * M + m) * S * (3 + C);
            if (cnt == 0) {
                empty[(size_t)b * M + m] = 1;
                memset(o, 0, sizeof(float) * (size_t)S * (3 + C));
                continue;
            }
            empty[(size_t)b * M + m] = 0;
            for (int s = 0; s < S; s++) {
                int k = sel[s < cnt ? s : s % cnt];
                float* row = o + (size_t)s * (3 + C);
                row[0] = p[k * 3]; row[1] = p[k * 3 + 1]; row[2] = p[k * 3 + 2];
                memcpy(row + 3, feat + ((size_t)b * N + k) * C, sizeof(float) * C);
            }
        }
    free(sel);
}

/* ------------------------------------------------------------------ *
 * iou3d  (lib/utils/iou3d/src/iou3d_kernel.cu:14-303)
 * ------------------------------------------------------------------ */
typedef struct { float x, y; } pt2;

static inline float cross2(pt2 a, pt2 b) {                 /* iou3d_kernel.cu:34-36 */
    float u = a.x * b.y, v = a.y * b.x;
    return u - v;
}
static inline float cross3(pt2 p1, pt2 p2, pt2 p0) {       /* iou3d_kernel.cu:38-40 */
    float a = (p1.x - p0.x) * (p2.y - p0.y);
    float b = (p2.x - p0.x) * (p1.y - p0.y);
    return a - b;
}
static inline float fmin2(float a, float b) { return a < b ? a : b; }   /* CUDA min/max(float,float) */
static inline float fmax2(float a, float b) { return a > b ? a : b; }

static int check_rect_cross(pt2 p1, pt2 p2, pt2 q1, pt2 q2) {           /* iou3d_kernel.cu:42-48 */
    return fmin2(p1.x, p2.x) <= fmax2(q1.x, q2.x) && fmin2(q1.x, q2.x) <= fmax2(p1.x, p2.x) &&
           fmin2(p1.y, p2.y) <= fmax2(q1.y, q2.y) && fmin2(q1.y, q2.y) <= fmax2(p1.y, p2.y);
}

/* iou3d_kernel.cu:50-65 check_in_box2d; (c,s) = (cos(-angle), sin(-angle)) */
static int check_in_box2d(const float* box, pt2 p, float c, float s) {
    const float MARGIN = 1e-5f;
    float center_x = (box[0] + box[2]) / 2, center_y = (box[1] + box[3]) / 2;
    float dx = p.x - center_x, dy = p.y - center_y;
    float t0 = dx * c, t1 = dy * s;
    float rot_x = (t0 + t1) + center_x;
    float t2 = (-dx) * s, t3 = dy * c;
    float rot_y = (t2 + t3) + center_y;
    return (rot_x > box[0] - MARGIN && rot_x < box[2] + MARGIN && rot_y > box[1] - MARGIN && rot_y < box[3] + MARGIN);
}

/* iou3d_kernel.cu:67-96 intersection */
static int seg_intersection(pt2 p1, pt2 p0, pt2 q1, pt2 q0, pt2* ans) {
    const float EPS = 1e-8f;
    if (check_rect_cross(p0, p1, q0, q1) == 0) return 0;
    float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > EPS) {
        float a = s5 * q0.x, b = s1 * q1.x;
        ans->x = (a - b) / (s5 - s1);
        a = s5 * q0.y; b = s1 * q1.y;
        ans->y = (a - b) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x;
        float u = p0.x * p1.y, v = p1.x * p0.y;
        float c0 = u - v;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x;
        u = q0.x * q1.y; v = q1.x * q0.y;
        float c1 = u - v;
        u = a0 * b1; v = a1 * b0;
        float D = u - v;
        u = b0 * c1; v = b1 * c0;
        ans->x = (u - v) / D;
        u = a1 * c0; v = a0 * c1;
        ans->y = (u - v) / D;
    }
    return 1;
}

/* iou3d_kernel.cu:98-102 rotate_around_center */
static pt2 rotate_around_center(pt2 center, float c, float s, pt2 p) {
    float dx = p.x - center.x, dy = p.y - center.y;
    float t0 = dx * c, t1 = dy * s;
    float nx = (t0 + t1) + center.x;
    float t2 = (-dx) * s, t3 = dy * c;
    float ny = (t2 + t3) + center.y;
    pt2 r = { nx, ny };
    return r;
}

/* canonical ordering key: strictly increasing in atan2(dy,dx) over (-pi, pi], IEEE +,-,/ only */
static float angle_key(float dx, float dy) {
    float ax = fabsf(dx), ay = fabsf(dy);
    float s = ax + ay;
    if (!(s > 0.0f)) return 0.0f;
    float t = dy / s;
    if (dx >= 0.0f) return t;
    return dy >= 0.0f ? 2.0f - t : -2.0f - t;
}

static void iou_trig(float angle, int trig_mode, float* c, float* s) {
    if (trig_mode == 0) { *c = cosf(angle); *s = sinf(angle); }             /* iou3d_kernel.cu:135-136 */
    else if (trig_mode == 2) { *c = prcnn_ref_cosf(angle); *s = prcnn_ref_sinf(angle); }
    else { *c = (float)cos((double)angle); *s = (float)sin((double)angle); }
}

/* iou3d_kernel.cu:108-212 box_overlap */
static float box_overlap(const float* box_a, const float* box_b, int trig_mode) {
    float a_x1 = box_a[0], a_y1 = box_a[1], a_x2 = box_a[2], a_y2 = box_a[3], a_angle = box_a[4];
    float b_x1 = box_b[0], b_y1 = box_b[1], b_x2 = box_b[2], b_y2 = box_b[3], b_angle = box_b[4];
    pt2 center_a = { (a_x1 + a_x2) / 2, (a_y1 + a_y2) / 2 };
    pt2 center_b = { (b_x1 + b_x2) / 2, (b_y1 + b_y2) / 2 };
    pt2 A[5] = { { a_x1, a_y1 }, { a_x2, a_y1 }, { a_x2, a_y2 }, { a_x1, a_y2 } };
    pt2 Bc[5] = { { b_x1, b_y1 }, { b_x2, b_y1 }, { b_x2, b_y2 }, { b_x1, b_y2 } };
    float ac, as, bc, bs, acn, asn, bcn, bsn;
    iou_trig(a_angle, trig_mode, &ac, &as);
    iou_trig(b_angle, trig_mode, &bc, &bs);
    if (trig_mode == 0) {   /* check_in_box2d evaluates cos(-angle), sin(-angle) itself (iou3d_kernel.cu:56) */
        acn = cosf(-a_angle); asn = sinf(-a_angle); bcn = cosf(-b_angle); bsn = sinf(-b_angle);
    } else if (trig_mode == 2) {
        acn = prcnn_ref_cosf(-a_angle); asn = prcnn_ref_sinf(-a_angle); bcn = prcnn_ref_cosf(-b_angle); bsn = prcnn_ref_sinf(-b_angle);
    } else {
        acn = ac; asn = -as; bcn = bc; bsn = -bs;
    }
    for (int k = 0; k < 4; k++) {
        A[k] = rotate_around_center(center_a, ac, as, A[k]);
        Bc[k] = rotate_around_center(center_b, bc, bs, Bc[k]);
    }
    A[4] = A[0];
    Bc[4] = Bc[0];

    pt2 cp[24];            /* the reference declares 16 (iou3d_kernel.cu:154); 24 = hard upper bound 16+8, avoids UB */
    pt2 poly_center = { 0, 0 };
    int cnt = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (seg_intersection(A[i + 1], A[i], Bc[j + 1], Bc[j], &cp[cnt])) {
                poly_center.x = poly_center.x + cp[cnt].x;
                poly_center.y = poly_center.y + cp[cnt].y;
                cnt++;
            }
    for (int k = 0; k < 4; k++) {
        if (check_in_box2d(box_a, Bc[k], acn, asn)) {
            poly_center.x = poly_center.x + Bc[k].x; poly_center.y = poly_center.y + Bc[k].y;
            cp[cnt++] = Bc[k];
        }
        if (check_in_box2d(box_b, A[k], bcn, bsn)) {
            poly_center.x = poly_center.x + A[k].x; poly_center.y = poly_center.y + A[k].y;
            cp[cnt++] = A[k];
        }
    }
    if (cnt == 0) return 0.0f;     /* reference: 0/0 centre, empty loops, area 0 (iou3d_kernel.cu:184-211) */
    poly_center.x /= cnt;
    poly_center.y /= cnt;

    float key[24];
    for (int i = 0; i < cnt; i++) {
        float dy = cp[i].y - poly_center.y, dx = cp[i].x - poly_center.x;
        key[i] = trig_mode == 0 ? atan2f(dy, dx) : (trig_mode == 2 ? prcnn_ref_atan2f(dy, dx) : angle_key(dx, dy));   /* iou3d_kernel.cu:104-106 */
    }
    for (int j = 0; j < cnt - 1; j++)                                        /* iou3d_kernel.cu:188-196 */
        for (int i = 0; i < cnt - j - 1; i++)
            if (key[i] > key[i + 1]) {
                pt2 t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
                float tk = key[i]; key[i] = key[i + 1]; key[i + 1] = tk;
            }
    float area = 0;
    for (int k = 0; k < cnt - 1; k++) {                                      /* iou3d_kernel.cu:206-211 */
        pt2 u = { cp[k].x - cp[0].x, cp[k].y - cp[0].y };
        pt2 v = { cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y };
        area += cross2(u, v);
    }
    return fabsf(area) / 2.0f;
}

static float iou_bev(const float* a, const float* b, int trig_mode) {      /* iou3d_kernel.cu:214-221 */
    float sa = (a[2] - a[0]) * (a[3] - a[1]);
    float sb = (b[2] - b[0]) * (b[3] - b[1]);
    float s_overlap = box_overlap(a, b, trig_mode);
    return s_overlap / fmaxf(sa + sb - s_overlap, 1e-8f);
}

static float iou_normal(const float* a, const float* b) {                  /* iou3d_kernel.cu:295-303 */
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    float interS = width * height;
    float Sa = (a[2] - a[0]) * (a[3] - a[1]);
    float Sb = (b[2] - b[0]) * (b[3] - b[1]);
    return interS / fmaxf(Sa + Sb - interS, 1e-8f);
}

/* iou3d_kernel.cu:223-234 / iou3d.cpp:31-50 */
PRCNN_EXPORT void prcnn_cpu_boxes_overlap_bev(const float* a, int Na, const float* b, int Nb, int trig_mode, float* out) {
    for (int i = 0; i < Na; i++)
        for (int j = 0; j < Nb; j++) out[(size_t)i * Nb + j] = box_overlap(a + i * 5, b + j * 5, trig_mode);
}
/* iou3d_kernel.cu:236-248 / iou3d.cpp:52-71 */
PRCNN_EXPORT void prcnn_cpu_boxes_iou_bev(const float* a, int Na, const float* b, int Nb, int trig_mode, float* out) {
    for (int i = 0; i < Na; i++)
        for (int j = 0; j < Nb; j++) out[(size_t)i * Nb + j] = iou_bev(a + i * 5, b + j * 5, trig_mode);
}

/* Greedy NMS on boxes ALREADY sorted by descending score (iou3d_utils.py:64-66):
 * iou3d_kernel.cu:250-292 (mask: row i vs col j>i, strict '>') + iou3d.cpp:100-119 (sweep).
 * kind 0 = rotated (iou_bev), 1 = normal (iou_normal).  Returns num_to_keep. */
PRCNN_EXPORT int prcnn_cpu_nms(const float* boxes, int N, float thresh, int kind, int trig_mode, int64_t* keep) {
    unsigned char* removed = (unsigned char*)calloc((size_t)(N > 0 ? N : 1), 1);
    int num = 0;
    for (int i = 0; i < N; i++) {
        if (removed[i]) continue;
        keep[num++] = i;
        for (int j = i + 1; j < N; j++) {
            if (removed[j]) continue;     /* OR-ing an already set bit is a no-op in the reference sweep */
            float v = kind == 0 ? iou_bev(boxes + (size_t)i * 5, boxes + (size_t)j * 5, trig_mode)
                                : iou_normal(boxes + (size_t)i * 5, boxes + (size_t)j * 5);
            if (v > thresh) removed[j] = 1;
        }
    }
    free(removed);
    return num;
}

/* full pairwise mask words exactly as the reference kernel lays them out
 * (iou3d_kernel.cu:250-292): mask[i*col_blocks + cb] bit t = iou(i, cb*64+t) > thr,
 * diagonal tile only t > i%64.  Used to pin the HIP mask kernel's upper triangle. */
PRCNN_EXPORT void prcnn_cpu_nms_mask(const float* boxes, int N, float thresh, int kind, int trig_mode, uint64_t* mask) {
    int cbn = (N + 63) / 64;
    for (int i = 0; i < N; i++)
        for (int cb = 0; cb < cbn; cb++) {
            uint64_t t = 0;
            int start = (i / 64 == cb) ? (i % 64) + 1 : 0;
            for (int tt = start; tt < 64 && cb * 64 + tt < N; tt++) {
                int j = cb * 64 + tt;
                float v = kind == 0 ? iou_bev(boxes + (size_t)i * 5, boxes + (size_t)j * 5, trig_mode)
                                    : iou_normal(boxes + (size_t)i * 5, boxes + (size_t)j * 5);
                if (v > thresh) t |= 1ULL << tt;
            }
            mask[(size_t)i * cbn + cb] = t;
        }
}

/* ======================================================================================================
 * Proposal stage: bin-based box decode + distance-split top-k + NMS + top-k.
 *   PINNED -- golden fixtures under tests/golden/proposal_ref.npz are produced by the reference's own
 *   lib/utils/bbox_transform.py and lib/rpn/proposal_layer.py imported on CPU (tests/golden/make_golden.py),
 *   with the compiled-extension calls inside iou3d_utils.nms_* routed to oracle/_ref.
 * ====================================================================================================== */

static inline int argmax_first(const float* v, int n) {       /* torch.argmax: first maximal value; NaN is maximal */
    int best = 0;
    for (int i = 1; i < n; i++) {
        if (v[best] != v[best]) break;                        /* NaN already found: it is the first maximum */
        if (v[i] > v[best] || v[i] != v[i]) best = i;
    }
    return best;
}

static inline float remainder_f32(float a, float b) {         /* torch.remainder (ATen BinaryOpsKernel: fmod + sign fix) */
    float m = fmodf(a, b);
    if (m != 0.0f && ((b < 0.0f) != (m < 0.0f))) m += b;
    return m;
}

/* lib/utils/bbox_transform.py:24-121 decode_bbox_target.  Every torch tensor op is one individually rounded
 * fp32 operation; Python-side scalars are computed in double and rounded to fp32 when they meet a tensor.
 *   roi: (N, roi_cols) with roi_cols 3 (xyz: RPN stage, proposal_layer.py:23) or 7 (RoI boxes: eval_rcnn.py:509)
 *   y_to_bottom: also apply proposal_layer.py:32  (y += h / 2)
 * returns 0, or -1 when C does not match the layout implied by the flags (the reference asserts, :106). */
PRCNN_EXPORT int prcnn_cpu_decode_bbox_target(const float* roi, int roi_cols, const float* reg, int N, int C,
                                              double loc_scope, double loc_bin_size, int num_head_bin, const float* anchor,
                                              int get_xz_fine, int get_y_by_bin, double loc_y_scope, double loc_y_bin_size,
                                              int get_ry_fine, int y_to_bottom, int trig_mode, float* out) {
    const int nb = (int)(loc_scope / loc_bin_size) * 2;       /* :42 */
    const int nyb = (int)(loc_y_scope / loc_y_bin_size) * 2;  /* :43 */
    int off = nb * 2;
    const int x_res_l = nb * 2, z_res_l = nb * 3;
    if (get_xz_fine) off = nb * 4;
    int y_bin_l = 0, y_res_l = 0, y_off = 0;
    if (get_y_by_bin) { y_bin_l = off; y_res_l = off + nyb; off += 2 * nyb; } else { y_off = off; off += 1; }
    const int ry_bin_l = off, ry_res_l = off + num_head_bin, size_l = off + 2 * num_head_bin;
    if (size_l + 3 != C || (roi_cols != 3 && roi_cols != 7)) return -1;
    const float lbs = (float)loc_bin_size, half_lbs = (float)(loc_bin_size / 2), scope = (float)loc_scope;
    const float lybs = (float)loc_y_bin_size, half_lybs = (float)(loc_y_bin_size / 2), yscope = (float)loc_y_scope;
    const double PI = 3.141592653589793;
    for (int i = 0; i < N; i++) {
        const float* r = reg + (size_t)i * C;
        const float* b = roi + (size_t)i * roi_cols;
        int xb = argmax_first(r, nb), zb = argmax_first(r + nb, nb);
        float px = ((float)xb * lbs + half_lbs) - scope;     /* :53-54 */
        float pz = ((float)zb * lbs + half_lbs) - scope;
        if (get_xz_fine) {                                    /* :56-67 */
            px = px + r[x_res_l + xb] * lbs;
            pz = pz + r[z_res_l + zb] * lbs;
        }
        float py;
        if (get_y_by_bin) {                                   /* :70-79 */
            int yb = argmax_first(r + y_bin_l, nyb);
            float y_res = r[y_res_l + yb] * lybs;
            py = (((float)yb * lybs + half_lybs) - yscope) + y_res;
            py = py + b[1];
        } else {
            py = b[1] + r[y_off];                             /* :84 */
        }
        int rb = argmax_first(r + ry_bin_l, num_head_bin);
        float rn = r[ry_res_l + rb], ry;
        if (get_ry_fine) {                                    /* :92-96 */
            double apc = (PI / 2) / num_head_bin;
            float ry_res = rn * (float)(apc / 2);
            ry = ((((float)rb * (float)apc) + (float)(apc / 2)) + ry_res) - (float)(PI / 4);
        } else {                                              /* :97-103 */
            double apc = (2 * PI) / num_head_bin;
            float ry_res = rn * (float)(apc / 2);
            ry = remainder_f32((float)rb * (float)apc + ry_res, (float)(2 * PI));
            if (ry > (float)PI) ry = ry - (float)(2 * PI);
        }
        float h = r[size_l] * anchor[0] + anchor[0];          /* :109-110 */
        float w = r[size_l + 1] * anchor[1] + anchor[1];
        float l = r[size_l + 2] * anchor[2] + anchor[2];
        if (roi_cols == 7) {                                  /* :116-119 -> rotate_pc_along_y_torch(box, -roi_ry) :5-21 */
            float ang = -b[6], ca, sa;
            if (trig_mode == 0) { ca = cosf(ang); sa = sinf(ang); }
            else { ca = (float)cos((double)ang); sa = (float)sin((double)ang); }
            float nx = px * ca + pz * (-sa);
            float nz = px * sa + pz * ca;
            px = nx; pz = nz;
            ry = ry + b[6];
        }
        px = px + b[0];                                       /* :120 */
        pz = pz + b[2];
        if (y_to_bottom) py = py + h / 2;                     /* proposal_layer.py:32 */
        float* o = out + (size_t)i * 7;
        o[0] = px; o[1] = py; o[2] = pz; o[3] = h; o[4] = w; o[5] = l; o[6] = ry;
    }
    return 0;
}

/* descending-score order, canonical total order: NaN first (torch.sort treats NaN as the largest value),
 * -0.0 == +0.0, ties by ascending index (torch.sort's tie order is unspecified unless stable=True). */
typedef struct { float s; int32_t i; } sort_rec;
static int sort_rec_cmp(const void* pa, const void* pb) {
    const sort_rec *a = (const sort_rec*)pa, *b = (const sort_rec*)pb;
    int an = a->s != a->s, bn = b->s != b->s;
    if (an != bn) return an ? -1 : 1;
    if (!an) { if (a->s > b->s) return -1; if (a->s < b->s) return 1; }
    return a->i < b->i ? -1 : (a->i > b->i ? 1 : 0);
}
PRCNN_EXPORT void prcnn_cpu_argsort_desc(const float* scores, int N, int32_t* order) {
    sort_rec* r = (sort_rec*)malloc(sizeof(sort_rec) * (size_t)(N > 0 ? N : 1));
    for (int i = 0; i < N; i++) { r[i].s = scores[i]; r[i].i = i; }
    qsort(r, (size_t)N, sizeof(sort_rec), sort_rec_cmp);
    for (int i = 0; i < N; i++) order[i] = r[i].i;
    free(r);
}

static void box3d_to_bev(const float* b, float* v) {          /* kitti_utils.py:134-147 */
    float half_l = b[5] / 2, half_w = b[4] / 2;
    v[0] = b[0] - half_l; v[1] = b[2] - half_w; v[2] = b[0] + half_l; v[3] = b[2] + half_w; v[4] = b[6];
}

/* greedy NMS over a candidate list (already in descending-score order), stopping after max_keep survivors:
 * identical keep set to prcnn_cpu_nms(...)[:max_keep] (a box is only ever tested against KEPT earlier boxes). */
static int nms_list(const float* boxes3d, const int32_t* cand, int n, float thresh, int kind, int trig_mode, int max_keep,
                    int32_t* kept) {
    float* kb = (float*)malloc(sizeof(float) * 5 * (size_t)(max_keep > 0 ? max_keep : 1));
    int nk = 0;
    for (int c = 0; c < n && nk < max_keep; c++) {
        float v[5];
        box3d_to_bev(boxes3d + (size_t)cand[c] * 7, v);
        int dead = 0;
        for (int k = 0; k < nk && !dead; k++) {
            float iou = kind == 0 ? iou_bev(kb + k * 5, v, trig_mode) : iou_normal(kb + k * 5, v);
            dead = iou > thresh;
        }
        if (!dead) { memcpy(kb + nk * 5, v, sizeof(v)); kept[nk++] = cand[c]; }
    }
    free(kb);
    return nk;
}

/* lib/rpn/proposal_layer.py:35-119 on already-decoded boxes (B,N,7) and raw scores (B,N).
 *   use_range=1: distance_based_proposal :58-117 -- area 1 = (r0, r1], area 2 = (r1, r2]; area 2 empty -> ranks
 *                [pre1, pre1+pre2) of area 1 (:91-98).  (The reference ASSERTS when area 1 is empty, :90; here an
 *                empty area simply contributes no proposals.)
 *   use_range=0: score_based_proposal :119-141 (pre2 = post2 = 0, always rotated NMS in the reference).
 * out_boxes (B, post1+post2, 7) / out_scores (B, post1+post2) are zero padded (:38-39); out_count (B) = rows filled. */
PRCNN_EXPORT void prcnn_cpu_proposal_layer(const float* scores, const float* boxes3d, int B, int N, int use_range, float r0,
                                           float r1, float r2, int pre1, int pre2, int post1, int post2, float thresh,
                                           int kind, int trig_mode, float* out_boxes, float* out_scores, int32_t* out_count) {
    const int post = post1 + post2;
    int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    int32_t* a1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    int32_t* a2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
    int32_t* kept = (int32_t*)malloc(sizeof(int32_t) * (size_t)(post > 0 ? post : 1));
    memset(out_boxes, 0, sizeof(float) * 7 * (size_t)B * post);
    memset(out_scores, 0, sizeof(float) * (size_t)B * post);
    for (int b = 0; b < B; b++) {
        const float* sc = scores + (size_t)b * N;
        const float* bx = boxes3d + (size_t)b * N * 7;
        prcnn_cpu_argsort_desc(sc, N, order);                 /* :35 */
        int n1 = 0, n2 = 0;
        for (int k = 0; k < N; k++) {
            float d = bx[(size_t)order[k] * 7 + 2];           /* :78 dist = z of the score-ordered proposals */
            if (!use_range) { a1[n1++] = order[k]; continue; }
            if (d > r0 && d <= r1) a1[n1++] = order[k];       /* :79,82 */
            if (d > r1 && d <= r2) a2[n2++] = order[k];
        }
        int tot = 0;
        for (int seg = 0; seg < 2; seg++) {
            const int32_t* cand; int n, pre = seg ? pre2 : pre1, postk = seg ? post2 : post1;
            if (seg == 0) { cand = a1; n = n1 < pre ? n1 : pre; }                     /* :90-91 */
            else if (!use_range) { cand = a2; n = 0; }
            else if (n2 != 0) { cand = a2; n = n2 < pre ? n2 : pre; }
            else { int s = n1 < pre1 ? n1 : pre1; cand = a1 + s; n = n1 - s < pre ? n1 - s : pre; }   /* :97-98 */
            int nk = nms_list(bx, cand, n, thresh, kind, trig_mode, postk, kept);     /* :101-110 */
            for (int k = 0; k < nk; k++, tot++) {
                memcpy(out_boxes + ((size_t)b * post + tot) * 7, bx + (size_t)kept[k] * 7, sizeof(float) * 7);
                out_scores[(size_t)b * post + tot] = sc[kept[k]];
            }
        }
        if (out_count) out_count[b] = tot;
    }
    free(order); free(a1); free(a2); free(kept);
}

/* Final detection select (tools/eval_rcnn.py:600-614), per frame: rows with valid != 0 (the caller's
 * norm_score > SCORE_THRESH test), ordered by descending raw score, greedy NMS on their BEV boxes.
 * keep (B, M): indices into the frame's M rows in kept order, -1 padded; num_keep (B). */
PRCNN_EXPORT void prcnn_cpu_nms_batched(const float* boxes3d, const float* scores, const uint8_t* valid, int B, int M,
                                        float thresh, int kind, int trig_mode, int max_keep, int32_t* keep, int32_t* num_keep) {
    if (max_keep <= 0 || max_keep > M) max_keep = M;
    int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (size_t)(M > 0 ? M : 1));
    for (int b = 0; b < B; b++) {
        prcnn_cpu_argsort_desc(scores + (size_t)b * M, M, order);
        int n = 0;
        for (int k = 0; k < M; k++)
            if (!valid || valid[(size_t)b * M + order[k]]) cand[n++] = order[k];
        int32_t* kp = keep + (size_t)b * max_keep;
        int nk = nms_list(boxes3d + (size_t)b * M * 7, cand, n, thresh, kind, trig_mode, max_keep, kp);
        for (int k = nk; k < max_keep; k++) kp[k] = -1;
        num_keep[b] = nk;
    }
    free(order); free(cand);
}

/* Canonical transformation of pooled RoI points, in place on a pooled tensor (B, M, S, W) whose first 3 columns are
 * xyz (lib/net/rcnn_net.py:143-150): xyz -= roi centre (raw roi x,y,z), then kitti_utils.py:45-63
 * rotate_pc_along_y_torch(xyz, roi ry): [x', z'] = [x, z] . R^T with R = [[cos, -sin], [sin, cos]].
 * PINNED against the reference's own Python (tests/golden/make_golden.py: canonical_ref) to 1e-5: torch's cos/sin and
 * bmm are not bit-specified; trig_mode 0 = cosf/sinf, 1 = canonical (double trig rounded once), products and sums
 * individually rounded. */
PRCNN_EXPORT void prcnn_cpu_canonical_transform(float* pooled, const float* rois, int B, int M, int S, int W, int trig_mode) {
    for (size_t r = 0; r < (size_t)B * M; r++) {
        const float* roi = rois + r * 7;
        float ca, sa;
        box_trig(roi[6], trig_mode, &ca, &sa);
        for (int s = 0; s < S; s++) {
            float* p = pooled + (r * S + s) * W;
            float x = p[0] - roi[0], y = p[1] - roi[1], z = p[2] - roi[2];
            float nx = x * ca + z * (-sa);
            float nz = x * sa + z * ca;
            p[0] = nx; p[1] = y; p[2] = nz;
        }
    }
}

/* ======================================================================================================
 * KITTI object evaluation: tools/kitti_object_eval_python/{rotate_iou,eval}.py.
 *   PINNED -- tests/golden/kitti_eval_ref.npz holds the outputs of the reference's own Python (numba decorators
 *   stubbed to identity, the single CUDA launch replaced by a loop over its own device function; see
 *   tests/golden/ref_kitti_eval.py).  The matching / counting logic is reproduced exactly (integer tp/fp/fn); IoU
 *   arithmetic to ~1e-6 (numba's float32/float64 type inference cannot be reproduced without numba).
 * Canonical IoU arithmetic (shared with kitti_eval.hip): fp32, every operation individually rounded, cos/sin of the
 * box angle evaluated in double and rounded once.
 * ====================================================================================================== */
static void kr_corners(const float* r, float* c) {                      /* rotate_iou.py:203-227 rbbox_to_corners */
    float a_cos = (float)cos((double)r[4]), a_sin = (float)sin((double)r[4]);
    float xd = r[2], yd = r[3];
    float cx[4] = {-xd / 2, -xd / 2, xd / 2, xd / 2}, cy[4] = {-yd / 2, yd / 2, yd / 2, -yd / 2};
    for (int i = 0; i < 4; i++) {
        c[2 * i] = a_cos * cx[i] + a_sin * cy[i] + r[0];
        c[2 * i + 1] = -a_sin * cx[i] + a_cos * cy[i] + r[1];
    }
}
static int kr_point_in_quad(float px, float py, const float* c) {       /* :161-177 */
    float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    float ap0 = px - c[0], ap1 = py - c[1];
    float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}
static int kr_seg_intersection(const float* p1, const float* p2, int i, int j, float* t) {    /* :77-115 */
    float A0 = p1[2 * i], A1 = p1[2 * i + 1], B0 = p1[2 * ((i + 1) % 4)], B1 = p1[2 * ((i + 1) % 4) + 1];
    float C0 = p2[2 * j], C1 = p2[2 * j + 1], D0 = p2[2 * ((j + 1) % 4)], D1 = p2[2 * ((j + 1) % 4) + 1];
    float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    int acd = DA1 * CA0 > CA1 * DA0;
    int bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd != bcd) {
        int abc = CA1 * BA0 > BA1 * CA0, abd = DA1 * BA0 > BA1 * DA0;
        if (abc != abd) {
            float DC0 = D0 - C0, DC1 = D1 - C1;
            float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
            float DH = BA1 * DC0 - BA0 * DC1, Dx = ABBA * DC0 - BA0 * CDDC, Dy = ABBA * DC1 - BA1 * CDDC;
            t[0] = Dx / DH; t[1] = Dy / DH;
            return 1;
        }
    }
    return 0;
}
#define KR_MAXPTS 24     /* the reference's local array holds 8 points (int_pts[16]); 24 can never overflow */
static float kr_inter(const float* r1, const float* r2) {               /* :230-245 inter */
    float c1[8], c2[8], ip[2 * KR_MAXPTS], vs[KR_MAXPTS], t[2];
    kr_corners(r1, c1); kr_corners(r2, c2);
    int n = 0;
    for (int i = 0; i < 4; i++) {                                        /* :180-200 quadrilateral_intersection */
        if (kr_point_in_quad(c1[2 * i], c1[2 * i + 1], c2)) { ip[2 * n] = c1[2 * i]; ip[2 * n + 1] = c1[2 * i + 1]; n++; }
        if (kr_point_in_quad(c2[2 * i], c2[2 * i + 1], c1)) { ip[2 * n] = c2[2 * i]; ip[2 * n + 1] = c2[2 * i + 1]; n++; }
    }
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            if (n < KR_MAXPTS && kr_seg_intersection(c1, c2, i, j, t)) { ip[2 * n] = t[0]; ip[2 * n + 1] = t[1]; n++; }
    if (n > 0) {                                                         /* :33-74 sort_vertex_in_convex_polygon */
        float cx = 0.f, cy = 0.f;
        for (int i = 0; i < n; i++) { cx += ip[2 * i]; cy += ip[2 * i + 1]; }
        cx /= (float)n; cy /= (float)n;
        for (int i = 0; i < n; i++) {
            float v0 = ip[2 * i] - cx, v1 = ip[2 * i + 1] - cy;
            float d = sqrtf(v0 * v0 + v1 * v1);
            v0 = v0 / d; v1 = v1 / d;
            if (v1 < 0) v0 = -2 - v0;
            vs[i] = v0;
        }
        for (int i = 1; i < n; i++)
            if (vs[i - 1] > vs[i]) {
                float temp = vs[i], tx = ip[2 * i], ty = ip[2 * i + 1];
                int j = i;
                while (j > 0 && vs[j - 1] > temp) {
                    vs[j] = vs[j - 1]; ip[2 * j] = ip[2 * j - 2]; ip[2 * j + 1] = ip[2 * j - 1];
                    j--;
                }
                vs[j] = temp; ip[2 * j] = tx; ip[2 * j + 1] = ty;
            }
    }
    float area = 0.f;                                                    /* :24-31 area (triangle fan) */
    for (int i = 0; i < n - 2; i++) {
        const float *a = ip, *b = ip + 2 * i + 2, *c = ip + 2 * i + 4;
        area += fabsf(((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0f);
    }
    return area;
}
static float kr_iou_eval(const float* r1, const float* r2, int criterion) {      /* :248-260 devRotateIoUEval */
    float area1 = r1[2] * r1[3], area2 = r2[2] * r2[3], ai = kr_inter(r1, r2);
    if (criterion == -1) return ai / (area1 + area2 - ai);
    if (criterion == 0) return ai / area1;
    if (criterion == 1) return ai / area2;
    return ai;
}
/* rotate_iou.py:287-329 rotate_iou_gpu_eval: boxes (N,5), query (K,5) [cx, cy, dx, dy, angle] -> iou (N,K);
 * the kernel evaluates devRotateIoUEval(query_k, box_n) (:282-284) */
PRCNN_EXPORT void prcnn_cpu_rotate_iou_eval(const float* boxes, int N, const float* query, int K, int criterion, float* out) {
    for (int n = 0; n < N; n++)
        for (int k = 0; k < K; k++) out[(size_t)n * K + k] = kr_iou_eval(query + 5 * k, boxes + 5 * n, criterion);
}

/* Per-frame overlap blocks as eval.py:326-397 calculate_iou_partly produces them for eval_class (which passes
 * (dt_annos, gt_annos): rows = detections, columns = ground truth).  metric 0: image_box_overlap on bbox (., 4)
 * (:87-113, float64); 1: bev_box_overlap on (x, z, l, w, ry) (:116-118, float32 result); 2: d3_box_overlap (:121-152:
 * rotated intersection area x height overlap / union volume, float64 math on a float32 area, stored as float32).
 * dt / gt: (.,4) for metric 0, (.,7) [x, y, z, l, h, w, ry] otherwise; offsets (F+1); out: concatenated row-major blocks. */
PRCNN_EXPORT void prcnn_cpu_kitti_overlaps(int metric, const double* dt, const int32_t* dt_off, const double* gt,
                                           const int32_t* gt_off, const int64_t* ov_off, int F, double* out) {
    for (int f = 0; f < F; f++) {
        int nd = dt_off[f + 1] - dt_off[f], ng = gt_off[f + 1] - gt_off[f];
        double* o = out + ov_off[f];
        for (int n = 0; n < nd; n++)
            for (int k = 0; k < ng; k++) {
                double v = 0.0;
                if (metric == 0) {
                    const double *b = dt + (size_t)(dt_off[f] + n) * 4, *q = gt + (size_t)(gt_off[f] + k) * 4;
                    double qarea = (q[2] - q[0]) * (q[3] - q[1]);
                    double iw = fmin(b[2], q[2]) - fmax(b[0], q[0]);
                    if (iw > 0) {
                        double ih = fmin(b[3], q[3]) - fmax(b[1], q[1]);
                        if (ih > 0) { double ua = (b[2] - b[0]) * (b[3] - b[1]) + qarea - iw * ih; v = iw * ih / ua; }
                    }
                } else {
                    const double *b = dt + (size_t)(dt_off[f] + n) * 7, *q = gt + (size_t)(gt_off[f] + k) * 7;
                    float rb[5] = {(float)b[0], (float)b[2], (float)b[3], (float)b[5], (float)b[6]};
                    float rq[5] = {(float)q[0], (float)q[2], (float)q[3], (float)q[5], (float)q[6]};
                    if (metric == 1) v = (double)kr_iou_eval(rq, rb, -1);
                    else {
                        float rinc = kr_iou_eval(rq, rb, 2);
                        if (rinc > 0) {
                            double iw = fmin(b[1], q[1]) - fmax(b[1] - b[4], q[1] - q[4]);
                            if (iw > 0) {
                                double area1 = b[3] * b[4] * b[5], area2 = q[3] * q[4] * q[5];
                                double inc = iw * (double)rinc;
                                v = (double)(float)(inc / (area1 + area2 - inc));
                            }
                        }
                    }
                }
                o[(size_t)n * ng + k] = v;
            }
    }
}

/* eval.py:155-268 compute_statistics_jit for ONE frame and ONE score threshold.
 * overlaps (nd, ng) row-major; gt_datas (ng,5) [bbox, alpha]; dt_datas (nd,6) [bbox, alpha, score]; dc (ndc,4).
 * returns tp, fp, fn, similarity in res[4]; matched (ng) receives the score of the detection that made gt i a true
 * positive, NaN elsewhere (the `thresholds` list of the reference, in gt order). */
static void kitti_stats_frame(const double* ov, const double* gtd, int ng, const double* dtd, int nd, const int32_t* ign_gt,
                              const int32_t* ign_det, const double* dc, int ndc, int metric, double min_overlap, double thresh,
                              int compute_fp, int compute_aos, double* res, double* matched) {
    unsigned char* assigned = (unsigned char*)calloc((size_t)(nd > 0 ? nd : 1), 1);
    unsigned char* ign_thr = (unsigned char*)calloc((size_t)(nd > 0 ? nd : 1), 1);
    if (compute_fp)
        for (int j = 0; j < nd; j++) ign_thr[j] = dtd[j * 6 + 5] < thresh;
    const double NO_DETECTION = -10000000;
    double tp = 0, fp = 0, fn = 0, sim_sum = 0;
    for (int i = 0; i < ng; i++) {
        if (matched) matched[i] = NAN;
        if (ign_gt[i] == -1) continue;
        int det_idx = -1, assigned_ignored = 0;
        double valid_detection = NO_DETECTION, max_overlap = 0;
        for (int j = 0; j < nd; j++) {
            if (ign_det[j] == -1 || assigned[j] || ign_thr[j]) continue;
            double overlap = ov[(size_t)j * ng + i], score = dtd[j * 6 + 5];
            if (!compute_fp && overlap > min_overlap && score > valid_detection) { det_idx = j; valid_detection = score; }
            else if (compute_fp && overlap > min_overlap && (overlap > max_overlap || assigned_ignored) && ign_det[j] == 0) {
                max_overlap = overlap; det_idx = j; valid_detection = 1; assigned_ignored = 0;
            } else if (compute_fp && overlap > min_overlap && valid_detection == NO_DETECTION && ign_det[j] == 1) {
                det_idx = j; valid_detection = 1; assigned_ignored = 1;
            }
        }
        if (valid_detection == NO_DETECTION && ign_gt[i] == 0) fn += 1;
        else if (valid_detection != NO_DETECTION && (ign_gt[i] == 1 || ign_det[det_idx] == 1)) assigned[det_idx] = 1;
        else if (valid_detection != NO_DETECTION) {
            tp += 1;
            if (matched) matched[i] = dtd[det_idx * 6 + 5];
            if (compute_aos) sim_sum += (1.0 + cos(gtd[i * 5 + 4] - dtd[det_idx * 6 + 4])) / 2.0;
            assigned[det_idx] = 1;
        }
    }
    double similarity = 0;
    if (compute_fp) {
        for (int j = 0; j < nd; j++)
            if (!(assigned[j] || ign_det[j] == -1 || ign_det[j] == 1 || ign_thr[j])) fp += 1;
        int nstuff = 0;
        if (metric == 0)
            for (int i = 0; i < ndc; i++)
                for (int j = 0; j < nd; j++) {
                    if (assigned[j] || ign_det[j] == -1 || ign_det[j] == 1 || ign_thr[j]) continue;
                    const double *b = dtd + j * 6, *q = dc + i * 4;      /* image_box_overlap(dt, dc, criterion 0) */
                    double o = 0, iw = fmin(b[2], q[2]) - fmax(b[0], q[0]);
                    if (iw > 0) {
                        double ih = fmin(b[3], q[3]) - fmax(b[1], q[1]);
                        if (ih > 0) o = iw * ih / ((b[2] - b[0]) * (b[3] - b[1]));
                    }
                    if (o > min_overlap) { assigned[j] = 1; nstuff++; }
                }
        fp -= nstuff;
        if (compute_aos) similarity = (tp > 0 || fp > 0) ? sim_sum : -1;
    }
    res[0] = tp; res[1] = fp; res[2] = fn; res[3] = similarity;
    free(assigned); free(ign_thr);
}
/* all frames x all thresholds: res (F, T, 4); matched (G) or NULL (written for t == 0 only) */
PRCNN_EXPORT void prcnn_cpu_kitti_statistics(const double* overlaps, const int64_t* ov_off, const double* gt_datas,
                                             const int32_t* gt_off, const double* dt_datas, const int32_t* dt_off,
                                             const int32_t* ign_gt, const int32_t* ign_det, const double* dc, const int32_t* dc_off,
                                             int F, int metric, double min_overlap, const double* thresholds, int T, int compute_fp,
                                             int compute_aos, double* res, double* matched) {
    for (int f = 0; f < F; f++)
        for (int t = 0; t < T; t++)
            kitti_stats_frame(overlaps + ov_off[f], gt_datas + (size_t)gt_off[f] * 5, gt_off[f + 1] - gt_off[f],
                              dt_datas + (size_t)dt_off[f] * 6, dt_off[f + 1] - dt_off[f], ign_gt + gt_off[f], ign_det + dt_off[f],
                              dc + (size_t)dc_off[f] * 4, dc_off[f + 1] - dc_off[f], metric, min_overlap, thresholds[t], compute_fp,
                              compute_aos, res + ((size_t)f * T + t) * 4, (matched && t == 0) ? matched + gt_off[f] : NULL);
}

/* ===================================================================================================
 * RPN input builder -- restates, per frame, the inference branch of
 * lib/datasets/kitti_rcnn_dataset.py:246-310 with lib/utils/calibration.py:51-70 and get_valid_flag (:198-219).
 * PINNED (deterministic part) against the reference's own Python run in the build container: tests/golden/ref_scene.py
 * -> tests/golden/scene_ref.npz (rect / image coordinates within fp32 rounding of the BLAS sgemm the reference calls,
 * valid flags identical away from the crop boundaries).  The random draw is RE-SPECIFIED (the reference uses numpy's
 * unseeded global Mersenne-Twister stream): see the contract in pointrcnn_amd/csrc/scene.hip, restated here.
 * =================================================================================================== */
static uint32_t scene_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static uint32_t scene_rand(uint32_t seed, uint32_t stream, uint32_t frame, uint32_t i) {
    return scene_mix(i ^ scene_mix(frame * 0x9E3779B9U + scene_mix(seed + stream * 0x85EBCA6BU)));
}

/* calibration.py:51-70 + kitti_rcnn_dataset.py:198-219 for n points of one frame.
 * calib: M (4x3 row-major = V2C^T . R0^T, calibration.py:57) then P2 (3x4).  scope: 6 doubles or NULL. */
PRCNN_EXPORT void prcnn_cpu_scene_project(const float* raw, int n, const float* calib, int H, int W, const double* scope,
                                          float* rect, float* img, float* depth, int32_t* flag) {
    const float* c = calib;
    const float* P = calib + 12;
    for (int i = 0; i < n; i++) {
        const float x = raw[i * 4], y = raw[i * 4 + 1], z = raw[i * 4 + 2];
        float r[3];
        for (int a = 0; a < 3; a++) r[a] = ((x * c[a] + y * c[3 + a]) + z * c[6 + a]) + c[9 + a];     /* calibration.py:56-57 */
        float h[3];
        for (int a = 0; a < 3; a++) h[a] = ((r[0] * P[a * 4] + r[1] * P[a * 4 + 1]) + r[2] * P[a * 4 + 2]) + P[a * 4 + 3];   /* :66-67 */
        const float u = h[0] / r[2], v = h[1] / r[2];      /* :68 divides by the RECT z */
        const float d = h[2] - P[11];                      /* :69 */
        int ok = (u >= 0.f) && (u < (float)W) && (v >= 0.f) && (v < (float)H) && (d >= 0.f);            /* :207-210 */
        if (scope)                                          /* :212-218, float32 vs double bounds compare in double */
            ok = ok && ((double)r[0] >= scope[0]) && ((double)r[0] <= scope[1]) && ((double)r[1] >= scope[2]) &&
                 ((double)r[1] <= scope[3]) && ((double)r[2] >= scope[4]) && ((double)r[2] <= scope[5]);
        if (rect) { rect[i * 3] = r[0]; rect[i * 3 + 1] = r[1]; rect[i * 3 + 2] = r[2]; }
        if (img) { img[i * 2] = u; img[i * 2 + 1] = v; }
        if (depth) depth[i] = d;
        flag[i] = ok;
    }
}

typedef struct { uint64_t key; } scene_key;
static int scene_cmp(const void* a, const void* b) {
    const uint64_t x = ((const scene_key*)a)->key, y = ((const scene_key*)b)->key;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* kitti_rcnn_dataset.py:285-310 for a batch; the draw per the re-specified contract. */
PRCNN_EXPORT void prcnn_cpu_scene_prepare(const float* raw, const int64_t* off, int B, const float* calib, const int32_t* img_hw,
                                          const double* scope, int npoints, uint32_t seed, float* out_xyz, float* out_int,
                                          int32_t* out_src, int32_t* nvalid, int32_t* status) {
    for (int b = 0; b < B; b++) {
        const float* R = raw + off[b] * 4;
        const int n_raw = (int)(off[b + 1] - off[b]);
        const float* c = calib + b * 24;
        float* rect = (float*)malloc((size_t)(n_raw > 0 ? n_raw : 1) * 3 * sizeof(float));
        int32_t* flag = (int32_t*)malloc((size_t)(n_raw > 0 ? n_raw : 1) * sizeof(int32_t));
        prcnn_cpu_scene_project(R, n_raw, c, img_hw[b * 2], img_hw[b * 2 + 1], scope, rect, NULL, NULL, flag);
        int n = 0, f = 0;
        for (int i = 0; i < n_raw; i++)
            if (flag[i]) { n++; if (!(rect[i * 3 + 2] < 40.0f)) f++; }           /* :287-288 near = depth < 40.0 */
        nvalid[b] = n;
        float* ox = out_xyz + (size_t)b * npoints * 3;
        float* oi = out_int + (size_t)b * npoints;
        int32_t* os = out_src + (size_t)b * npoints;
        if (n == 0) {
            for (int j = 0; j < npoints; j++) { ox[j * 3] = ox[j * 3 + 1] = ox[j * 3 + 2] = 0.f; oi[j] = 0.f; os[j] = -1; }
            status[b] = 2;
            free(rect); free(flag);
            continue;
        }
        int st = 0, k, keep_far, keep_all;
        if (n > npoints) {                                   /* :286-294 */
            keep_all = 0;
            if (f > npoints) { st = 1; keep_far = 0; k = npoints; }
            else { keep_far = 1; k = npoints - f; }
        } else {                                             /* :296-301 */
            keep_all = 1; keep_far = 0;
            k = npoints - n;
            if (k > n) { st = 1; k = n; }
        }
        /* candidates ordered by (30-bit draw key, raw index): the first k are drawn */
        scene_key* cand = (scene_key*)malloc((size_t)n * sizeof(scene_key));
        scene_key* sel = (scene_key*)malloc((size_t)(npoints + 1) * sizeof(scene_key));
        int nc = 0, ns = 0;
        for (int i = 0; i < n_raw; i++) {
            if (!flag[i]) continue;
            const int far = !(rect[i * 3 + 2] < 40.0f);
            if (!(keep_far && far)) cand[nc++].key = ((uint64_t)(scene_rand(seed, 0u, (uint32_t)b, (uint32_t)i) >> 2) << 32) | (uint32_t)i;
            if (keep_all || (keep_far && far)) sel[ns++].key = ((uint64_t)scene_rand(seed, 1u, (uint32_t)b, (uint32_t)i) << 32) | (uint32_t)i;
        }
        qsort(cand, (size_t)nc, sizeof(scene_key), scene_cmp);
        for (int q = 0; q < k && q < nc; q++) {
            const uint32_t i = (uint32_t)cand[q].key;
            sel[ns++].key = ((uint64_t)scene_rand(seed, keep_all ? 2u : 1u, (uint32_t)b, i) << 32) | i;
        }
        qsort(sel, (size_t)ns, sizeof(scene_key), scene_cmp);          /* the shuffle (:293 / :302) */
        for (int j = 0; j < npoints; j++) {
            const uint32_t i = (uint32_t)sel[j < ns ? j : j % ns].key;
            ox[j * 3] = rect[i * 3]; ox[j * 3 + 1] = rect[i * 3 + 1]; ox[j * 3 + 2] = rect[i * 3 + 2];     /* :304 */
            oi[j] = R[i * 4 + 3] - 0.5f;                                /* :305 */
            os[j] = (int32_t)i;
        }
        status[b] = st;
        free(cand); free(sel); free(rect); free(flag);
    }
}

/* ======================================================================================================
 * RCNN-stage training targets: ProposalTargetLayer.sample_rois_for_rcnn + aug_roi_by_noise_torch +
 * random_aug_box3d (lib/rpn/proposal_target_layer.py:75-300), iou3d_utils.boxes_iou3d_gpu (lib/utils/iou3d/iou3d_utils.py:20-53).
 * The reference draws from numpy's and torch's global generators in a data-dependent order; the draw is re-specified with the
 * counter-based generator of the input builder, one number per (purpose, frame, position) -- see csrc/proposal_target.hip for the
 * table.  Everything else (3-D IoU, assignment, the fg / hard-bg / easy-bg candidate sets, the four sampling cases, the
 * accept / retry loop of the noise augmentation) follows the reference line by line; tests/golden/ref_proposal_target.py runs the
 * reference's own methods with its random calls answered from the same table and must get the same boxes.
 * ====================================================================================================== */
static uint32_t pt_mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
static uint32_t pt_rand(uint32_t seed, uint32_t stream, uint32_t frame, uint32_t i) {
    return pt_mix(i ^ pt_mix(frame * 0x9E3779B9U + pt_mix(seed + stream * 0x85EBCA6BU)));
}
static float pt_u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }          /* torch.rand: 24 random bits */
static int pt_below(uint32_t r, int n) { return (int)(((uint64_t)r * (uint64_t)n) >> 32); } /* uniform integer in [0, n) */

/* iou3d_utils.py:20-53 for one pair, every operation an individually rounded fp32 operation */
static float pt_iou3d(const float* a, const float* b, int trig_mode) {
    float abev[5] = { a[0] - a[5] / 2, a[2] - a[4] / 2, a[0] + a[5] / 2, a[2] + a[4] / 2, a[6] };    /* kitti_utils.py:134-147 */
    float bbev[5] = { b[0] - b[5] / 2, b[2] - b[4] / 2, b[0] + b[5] / 2, b[2] + b[4] / 2, b[6] };
    float ov = box_overlap(abev, bbev, trig_mode);
    float amin = a[1] - a[3], bmin = b[1] - b[3];
    float max_of_min = amin > bmin ? amin : bmin, min_of_max = a[1] < b[1] ? a[1] : b[1];
    float h = min_of_max - max_of_min;
    if (!(h > 0.0f)) h = 0.0f;
    float ov3 = ov * h;
    float va = a[3] * a[4];
    va = va * a[5];
    float vb = b[3] * b[4];
    vb = vb * b[5];
    float den = va + vb;
    den = den - ov3;
    if (den < 1e-7f) den = 1e-7f;
    return ov3 / den;
}

PRCNN_EXPORT void prcnn_cpu_boxes_iou3d(const float* a, int Na, const float* b, int Nb, int trig_mode, float* out) {
    for (int i = 0; i < Na; i++)
        for (int j = 0; j < Nb; j++) out[(size_t)i * Nb + j] = pt_iou3d(a + (size_t)i * 7, b + (size_t)j * 7, trig_mode);
}

/* proposal_target_layer.py:240-300 random_aug_box3d, methods 'multiple' (0) and 'single' (1); draws r[0..7] */
static void pt_random_aug(const float* box, int method, const uint32_t* r, float* out) {
    float ps[3], hs[3], ar;
    if (method == 0) {
        static const double rc[5][3] = { {0.2, 0.1, 3.14159265358979323846 / 12}, {0.3, 0.15, 3.14159265358979323846 / 12},
                                         {0.5, 0.15, 3.14159265358979323846 / 9}, {0.8, 0.15, 3.14159265358979323846 / 6},
                                         {1.0, 0.15, 3.14159265358979323846 / 3} };
        const int idx = pt_below(r[0], 5);
        for (int c = 0; c < 3; c++) ps[c] = ((pt_u01(r[1 + c]) - 0.5f) / 0.5f) * (float)rc[idx][0];
        for (int c = 0; c < 3; c++) hs[c] = ((pt_u01(r[4 + c]) - 0.5f) / 0.5f) * (float)rc[idx][1] + 1.0f;
        ar = ((pt_u01(r[7]) - 0.5f) / 0.5f) * (float)rc[idx][2];
    } else {
        for (int c = 0; c < 3; c++) ps[c] = pt_u01(r[1 + c]) - 0.5f;
        for (int c = 0; c < 3; c++) hs[c] = (pt_u01(r[4 + c]) - 0.5f) / (float)(0.5 / 0.15) + 1.0f;
        ar = (pt_u01(r[7]) - 0.5f) / (float)(0.5 / (3.14159265358979323846 / 12));
    }
    for (int c = 0; c < 3; c++) { out[c] = box[c] + ps[c]; out[3 + c] = box[3 + c] * hs[c]; }
    out[6] = box[6] + ar;
}

typedef struct { uint32_t key; int idx; } pt_cand;
static int pt_cand_cmp(const void* x, const void* y) {
    const pt_cand *a = (const pt_cand*)x, *b = (const pt_cand*)y;
    if (a->key != b->key) return a->key < b->key ? -1 : 1;
    return a->idx - b->idx;
}

/* cfgv: reg_fg_thresh, cls_fg_thresh, cls_bg_thresh, cls_bg_thresh_lo, fg_ratio, hard_bg_ratio
 * outputs: rois / gt_of_rois (B,R,7), roi_iou (B,R), src (B,R) input RoI of every slot, max_overlaps / gt_assignment (B,M),
 * counts (B,4): fg candidates, hard-bg candidates, easy-bg candidates, fg slots; status (B): 0 ok, 1 = neither foreground nor
 * background candidates (the reference raises), 2 = no ground-truth box */
PRCNN_EXPORT void prcnn_cpu_proposal_target_sample(const float* roi_boxes3d, const float* gt_boxes3d, int B, int M, int G, int gt_cols,
                                                   int R, const double* cfgv, int aug_times, int aug_method, uint32_t seed, int trig_mode,
                                                   float* rois, float* gt_of_rois, float* roi_iou, int32_t* src, float* max_overlaps,
                                                   int32_t* gt_assignment, int32_t* counts, int32_t* status) {
    /* thresholds meet float32 overlaps; the two ratios stay double (Python arithmetic: int(10 * 0.7) = 7, with 0.7f it would be 6) */
    const float reg_fg = (float)cfgv[0], cls_fg = (float)cfgv[1], cls_bg = (float)cfgv[2], cls_bg_lo = (float)cfgv[3];
    const double fg_ratio = cfgv[4], hard_ratio = cfgv[5];
    const float fg_thresh = reg_fg < cls_fg ? reg_fg : cls_fg;
    const int fg_per_image = (int)nearbyint(fg_ratio * (double)R);                  /* np.round: half to even */
    pt_cand* cand = (pt_cand*)malloc(sizeof(pt_cand) * (size_t)(M > 0 ? M : 1));
    int* fg = (int*)malloc(sizeof(int) * (size_t)(M > 0 ? M : 1) * 3);
    int *hard = fg + M, *easy = fg + 2 * M;
    for (int b = 0; b < B; b++) {
        const float* roi = roi_boxes3d + (size_t)b * M * 7;
        const float* gt = gt_boxes3d + (size_t)b * G * gt_cols;
        float* o_roi = rois + (size_t)b * R * 7;
        float* o_gt = gt_of_rois + (size_t)b * R * 7;
        float* o_iou = roi_iou + (size_t)b * R;
        int32_t* o_src = src + (size_t)b * R;
        memset(o_roi, 0, sizeof(float) * (size_t)R * 7);
        memset(o_gt, 0, sizeof(float) * (size_t)R * 7);
        memset(o_iou, 0, sizeof(float) * (size_t)R);
        for (int t = 0; t < R; t++) o_src[t] = -1;
        int ng = G;                                              /* :96-99: drop the zero rows at the end */
        while (ng > 0) {
            float s = 0.f;
            for (int c = 0; c < gt_cols; c++) s += gt[(size_t)(ng - 1) * gt_cols + c];
            if (s != 0.f) break;
            ng--;
        }
        status[b] = 0;
        counts[b * 4] = counts[b * 4 + 1] = counts[b * 4 + 2] = counts[b * 4 + 3] = 0;
        if (ng == 0) { status[b] = 2; continue; }
        int nfg = 0, nhard = 0, neasy = 0;
        for (int i = 0; i < M; i++) {
            float best = 0.f; int arg = 0;
            for (int j = 0; j < ng; j++) {
                const float v = pt_iou3d(roi + (size_t)i * 7, gt + (size_t)j * gt_cols, trig_mode);
                if (j == 0 || v > best) { best = v; arg = j; }     /* torch.max: first maximum */
            }
            max_overlaps[(size_t)b * M + i] = best; gt_assignment[(size_t)b * M + i] = arg;
            if (best >= fg_thresh) fg[nfg++] = i;
            if (best < cls_bg_lo) easy[neasy++] = i;
            if (best < cls_bg && best >= cls_bg_lo) hard[nhard++] = i;
        }
        counts[b * 4] = nfg; counts[b * 4 + 1] = nhard; counts[b * 4 + 2] = neasy;
        const int nbg = nhard + neasy;
        int n_fg_slots = 0, n_bg_slots = 0;
        if (nfg > 0 && nbg > 0) {
            n_fg_slots = fg_per_image < nfg ? fg_per_image : nfg;
            for (int t = 0; t < nfg; t++) { cand[t].key = pt_rand(seed, 10, (uint32_t)b, (uint32_t)fg[t]); cand[t].idx = fg[t]; }
            qsort(cand, (size_t)nfg, sizeof(pt_cand), pt_cand_cmp);            /* a random permutation's prefix (:123-125) */
            for (int t = 0; t < n_fg_slots; t++) o_src[t] = cand[t].idx;
            n_bg_slots = R - n_fg_slots;
        } else if (nfg > 0) {
            n_fg_slots = R;                                                     /* :131-137: with replacement */
            for (int t = 0; t < R; t++) o_src[t] = fg[pt_below(pt_rand(seed, 11, (uint32_t)b, (uint32_t)t), nfg)];
        } else if (nbg > 0) {
            n_bg_slots = R;
        } else {
            status[b] = 1;
            continue;
        }
        counts[b * 4 + 3] = n_fg_slots;
        if (n_bg_slots > 0) {                                                   /* sample_bg_inds (:190-220) */
            int n_hard = 0;
            if (nhard > 0 && neasy > 0) n_hard = (int)((double)n_bg_slots * hard_ratio);     /* int(bg * HARD_BG_RATIO) */
            else if (nhard > 0) n_hard = n_bg_slots;
            for (int t = 0; t < n_bg_slots; t++) {
                if (t < n_hard) o_src[n_fg_slots + t] = hard[pt_below(pt_rand(seed, 12, (uint32_t)b, (uint32_t)t), nhard)];
                else o_src[n_fg_slots + t] = easy[pt_below(pt_rand(seed, 13, (uint32_t)b, (uint32_t)(t - n_hard)), neasy)];
            }
        }
        for (int t = 0; t < R; t++) {                                           /* aug_roi_by_noise_torch (:222-250) */
            const int i = o_src[t];
            const float* box = roi + (size_t)i * 7;
            const float* g = gt + (size_t)gt_assignment[(size_t)b * M + i] * gt_cols;
            const float iou_src = max_overlaps[(size_t)b * M + i];
            const int times = t < n_fg_slots ? aug_times : (aug_times > 0 ? 1 : 0);
            float aug[7], temp_iou = 0.f;
            int cnt = 0, keep = 1;
            memcpy(aug, box, sizeof(aug));
            while (temp_iou < fg_thresh && cnt < times) {
                uint32_t r[9];
                for (int q = 0; q < 9; q++) r[q] = pt_rand(seed, 20, (uint32_t)b, (uint32_t)((t * 16 + cnt) * 16 + q));
                if (pt_u01(r[8]) < 0.2f) { memcpy(aug, box, sizeof(aug)); keep = 1; }
                else { pt_random_aug(box, aug_method, r, aug); keep = 0; }
                temp_iou = pt_iou3d(aug, g, trig_mode);
                cnt++;
            }
            memcpy(o_roi + (size_t)t * 7, aug, sizeof(aug));
            memcpy(o_gt + (size_t)t * 7, g, sizeof(float) * 7);
            o_iou[t] = (cnt == 0 || keep) ? iou_src : temp_iou;
        }
    }
    free(cand);
    free(fg);
}


/* the restated libm functions (fn 0 sinf, 1 cosf, 2 atan2f(a, b)) and the host libm's own (fn 3, 4, 5), element-wise: tests */
PRCNN_EXPORT void prcnn_cpu_ref_trig(const float* a, const float* b, int n, int fn, float* out) {
    for (int i = 0; i < n; i++) {
        switch (fn) {
            case 0: out[i] = prcnn_ref_sinf(a[i]); break;
            case 1: out[i] = prcnn_ref_cosf(a[i]); break;
            case 2: out[i] = prcnn_ref_atan2f(a[i], b[i]); break;
            case 3: out[i] = sinf(a[i]); break;
            case 4: out[i] = cosf(a[i]); break;
            default: out[i] = atan2f(a[i], b[i]); break;
        }
    }
}
