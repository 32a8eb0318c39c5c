// iou3d_geom.h -- BEV box geometry shared by iou3d.hip (pair matrices, mask NMS) and proposal.hip (batched greedy NMS).
//
// The clipping algorithm follows the reference step for step (lib/utils/iou3d/src/iou3d_kernel.cu:34-221: edge-edge
// intersections, contained corners, centroid, angular sort, shoelace) in the REFERENCE'S arithmetic (oracle trig_mode 2 ==
// trig_mode 0 == the reference's own sources compiled for the host): every product / sum individually rounded, the
// box's cos / sin, the cos(-angle) / sin(-angle) of the containment test and the atan2 of the angular sort evaluated by
// ref_trig.h -- glibc's float routines restated bit for bit (rounds 1-2 used double-rounded trigonometry and a division-only
// surrogate of atan2, which agreed to 5e-7 in IoU but could flip a decision sitting within 3e-6 of its threshold).
#pragma once
#include "common.h"
#include "ref_trig.h"

struct Pt { float x, y; };
struct RBox {
    float x1, y1, x2, y2;   // raw extents
    float cx, cy;           // centre
    float c, s;             // cos(angle), sin(angle)
    float cn, sn;           // cos(-angle), sin(-angle) as the reference evaluates them (iou3d_kernel.cu:56)
    Pt p[5];                // rotated corners, p[4] == p[0]
    float rad;              // half diagonal: every corner lies on this circle around the centre
};

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }

__device__ __forceinline__ Pt rotate_around_center(float cx, float cy, float c, float s, float px, float py) {
    float dx = sub(px, cx), dy = sub(py, cy);                                   // iou3d_kernel.cu:98-102
    Pt r;
    r.x = add(add(mul(dx, c), mul(dy, s)), cx);
    r.y = add(add(mul(-dx, s), mul(dy, c)), cy);
    return r;
}

__device__ void make_rbox(const float* __restrict__ b, RBox& r) {
    r.x1 = b[0]; r.y1 = b[1]; r.x2 = b[2]; r.y2 = b[3];
    r.cx = add(r.x1, r.x2) / 2; r.cy = add(r.y1, r.y2) / 2;
    r.c = prcnn_ref_cosf(b[4]);
    r.s = prcnn_ref_sinf(b[4]);
    r.cn = prcnn_ref_cosf(-b[4]);
    r.sn = prcnn_ref_sinf(-b[4]);
    r.p[0] = rotate_around_center(r.cx, r.cy, r.c, r.s, r.x1, r.y1);
    r.p[1] = rotate_around_center(r.cx, r.cy, r.c, r.s, r.x2, r.y1);
    r.p[2] = rotate_around_center(r.cx, r.cy, r.c, r.s, r.x2, r.y2);
    r.p[3] = rotate_around_center(r.cx, r.cy, r.c, r.s, r.x1, r.y2);
    r.p[4] = r.p[0];
    float hx = sub(r.x2, r.x1) / 2, hy = sub(r.y2, r.y1) / 2;
    r.rad = sqrtf(add(mul(hx, hx), mul(hy, hy)));
}

// Conservative disjointness test: true only when the two rectangles' circumscribed circles are separated by a
// margin far larger than any rounding error or the reference's 1e-5 containment margin.  Then box_overlap() finds no
// intersection point and no contained corner and returns exactly 0 (iou3d_kernel.cu:108-186 with cnt == 0), so the
// full clipping can be skipped without changing a bit.  NaN / inf boxes compare false and take the full path.
__device__ __forceinline__ bool far_apart(const RBox& a, const RBox& b) {
    float dx = sub(a.cx, b.cx), dy = sub(a.cy, b.cy);
    float s = add(mul(add(a.rad, b.rad), 1.0001f), 1e-3f);
    return add(mul(dx, dx), mul(dy, dy)) > mul(s, s);
}

// Round 6: an upper bound of the overlap without the clip.  In A's frame A is the axis-aligned rectangle [-hax, hax] x [-hay, hay] and B
// lies inside ITS bounding rectangle there (centre offset d, half extents hbx |cos| + hby |sin| and hbx |sin| + hby |cos| of the angle
// between the boxes), so the intersection polygon lies inside the intersection of those two rectangles: area <= ox * oy.  The same in
// B's frame, and the overlap cannot exceed either box.  IoU = ov / (Sa + Sb - ov) grows with ov, so IoU <= u / (Sa + Sb - u), u the
// smallest of the bounds.  The decision "iou_bev > thresh" is taken from the bound only when it is FALSE WITH A 10 % MARGIN (and every
// extent is padded by 2 mm): the reference's clip evaluates the true overlap to ~1e-5 (fp32 on coordinates < 100 m; its 1e-5 containment
// margin and an ill-conditioned crossing of nearly parallel edges add slivers of that order), nowhere near 10 %.  Everything within
// the margin takes the clip.  Most pairs of an RPN's proposals around one object overlap by 0.2-0.6 against thresholds of 0.8-0.85:
// they used to cost ~5 k dependent instructions each and now cost ~50.
__device__ __forceinline__ float padded_axis_overlap(float h, float d, float e) {          // |[-h, h] n [d - e, d + e]|, padded
    return fmaxf(fminf(h, d + e) - fmaxf(-h, d - e) + 2e-3f, 0.f);
}
__device__ __forceinline__ bool cannot_exceed(const RBox& a, const RBox& b, float thresh) {
    const float hax = (a.x2 - a.x1) * 0.5f, hay = (a.y2 - a.y1) * 0.5f, hbx = (b.x2 - b.x1) * 0.5f, hby = (b.y2 - b.y1) * 0.5f;
    if (!(hax > 0.f && hay > 0.f && hbx > 0.f && hby > 0.f && thresh > 0.01f)) return false;          // (NaN compares false: full path)
    const float cd = fabsf(a.c * b.c + a.s * b.s), sd = fabsf(b.s * a.c - b.c * a.s);                // |cos|, |sin| of the angle between them
    const float dx = b.cx - a.cx, dy = b.cy - a.cy;
    // a box's local x axis is (c, -s), its y axis (s, c) in the plane (rotate_around_center above)
    const float u1 = padded_axis_overlap(hax, dx * a.c - dy * a.s, hbx * cd + hby * sd) * padded_axis_overlap(hay, dx * a.s + dy * a.c, hbx * sd + hby * cd);
    const float u2 = padded_axis_overlap(hbx, dy * b.s - dx * b.c, hax * cd + hay * sd) * padded_axis_overlap(hby, -dx * b.s - dy * b.c, hax * sd + hay * cd);
    const float sa = 4.f * hax * hay, sb = 4.f * hbx * hby;
    const float u = fminf(fminf(u1, u2), fminf(sa, sb));
    return u < 0.9f * thresh * (sa + sb - u);
}
// the pairs a threshold decision can skip: exact-zero overlap, or an overlap that cannot reach the threshold (thresh >= 0 at the call sites)
__device__ __forceinline__ bool decided_without_clip(const RBox& a, const RBox& b, float thresh) {
    return far_apart(a, b) || cannot_exceed(a, b, thresh);
}

__device__ __forceinline__ float cross3(Pt p1, Pt p2, Pt p0) {                  // iou3d_kernel.cu:38-40
    return sub(mul(sub(p1.x, p0.x), sub(p2.y, p0.y)), mul(sub(p2.x, p0.x), sub(p1.y, p0.y)));
}

__device__ __forceinline__ bool check_rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {  // iou3d_kernel.cu:42-48
    return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
           fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

// iou3d_kernel.cu:50-65
__device__ __forceinline__ bool check_in_box2d(const RBox& box, Pt p) {
    const float MARGIN = 1e-5f;
    float dx = sub(p.x, box.cx), dy = sub(p.y, box.cy);
    float sn = box.sn;
    float rot_x = add(add(mul(dx, box.cn), mul(dy, sn)), box.cx);
    float rot_y = add(add(mul(-dx, sn), mul(dy, box.cn)), box.cy);
    return rot_x > sub(box.x1, MARGIN) && rot_x < add(box.x2, MARGIN) && rot_y > sub(box.y1, MARGIN) &&
           rot_y < add(box.y2, MARGIN);
}

// iou3d_kernel.cu:67-96
__device__ __forceinline__ bool seg_intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt& ans) {
    const float EPS = 1e-8f;
    if (!check_rect_cross(p0, p1, q0, q1)) return false;
    float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
    if (!(mul(s1, s2) > 0 && mul(s3, s4) > 0)) return false;
    float s5 = cross3(q1, p1, p0);
    if (fabsf(sub(s5, s1)) > EPS) {
        float den = sub(s5, s1);
        ans.x = sub(mul(s5, q0.x), mul(s1, q1.x)) / den;
        ans.y = sub(mul(s5, q0.y), mul(s1, q1.y)) / den;
    } else {
        float a0 = sub(p0.y, p1.y), b0 = sub(p1.x, p0.x), c0 = sub(mul(p0.x, p1.y), mul(p1.x, p0.y));
        float a1 = sub(q0.y, q1.y), b1 = sub(q1.x, q0.x), c1 = sub(mul(q0.x, q1.y), mul(q1.x, q0.y));
        float D = sub(mul(a0, b1), mul(a1, b0));
        ans.x = sub(mul(b0, c1), mul(b1, c0)) / D;
        ans.y = sub(mul(a1, c0), mul(a0, c1)) / D;
    }
    return true;
}

// ordering key of the angular sort: atan2(dy, dx) exactly as the reference's host libm evaluates it (iou3d_kernel.cu:104-106)
__device__ __forceinline__ float angle_key(float dx, float dy) { return prcnn_ref_atan2f(dy, dx); }

// Storage of the clip's point list (up to 16 edge intersections + 8 contained corners, and their sort keys).  The reference keeps them in
// thread-local arrays and bubble-sorts them there (iou3d_kernel.cu:110-111,188-196); on gfx950 dynamically indexed private arrays live in
// scratch memory.  ClipLds keeps the same arrays in a lane-private column of an LDS block (element e of lane t at base[e * T + t]:
// conflict-free whatever the lanes' indices) -- same algorithm, same operations in the same order, same bits.  Measured in round 5 and NOT
// used by the NMS kernels: rotated NMS of 6300 boxes 777 vs 696 us, the proposal stage's rotated NMS 1586 (256 threads, LDS) vs 1346 us
// (512 threads, scratch) -- the clip is not waiting for its arrays, it is ~5 k mostly dependent instructions (16 edge tests, correctly
// rounded divisions, an atan2 per point) whatever the number of lanes that run it.
struct ClipPrivate {
    Pt cp[24];
    float k[24];
    __device__ __forceinline__ float& x(int i) { return cp[i].x; }
    __device__ __forceinline__ float& y(int i) { return cp[i].y; }
    __device__ __forceinline__ float& key(int i) { return k[i]; }
};
#define CLIP_LDS_FLOATS 72          // per lane
struct ClipLds {
    float* b;                       // block base + the lane's index
    int T;                          // lanes sharing the block (the stride between a lane's elements)
    __device__ __forceinline__ float& x(int i) { return b[(size_t)(3 * i) * T]; }
    __device__ __forceinline__ float& y(int i) { return b[(size_t)(3 * i + 1) * T]; }
    __device__ __forceinline__ float& key(int i) { return b[(size_t)(3 * i + 2) * T]; }
};

// iou3d_kernel.cu:108-212
template <class S>
__device__ float box_overlap_s(const RBox& A, const RBox& B, S& st) {
    float pcx = 0.f, pcy = 0.f;
    int cnt = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            Pt ans;
            if (seg_intersection(A.p[i + 1], A.p[i], B.p[j + 1], B.p[j], ans)) {
                pcx = add(pcx, ans.x); pcy = add(pcy, ans.y);
                st.x(cnt) = ans.x; st.y(cnt) = ans.y; cnt++;
            }
        }
    for (int k = 0; k < 4; k++) {
        if (check_in_box2d(A, B.p[k])) { pcx = add(pcx, B.p[k].x); pcy = add(pcy, B.p[k].y); st.x(cnt) = B.p[k].x; st.y(cnt) = B.p[k].y; cnt++; }
        if (check_in_box2d(B, A.p[k])) { pcx = add(pcx, A.p[k].x); pcy = add(pcy, A.p[k].y); st.x(cnt) = A.p[k].x; st.y(cnt) = A.p[k].y; cnt++; }
    }
    if (cnt == 0) return 0.0f;
    pcx = pcx / (float)cnt; pcy = pcy / (float)cnt;
    for (int i = 0; i < cnt; i++) st.key(i) = angle_key(sub(st.x(i), pcx), sub(st.y(i), pcy));
    for (int j = 0; j < cnt - 1; j++)                                            // iou3d_kernel.cu:188-196
        for (int i = 0; i < cnt - j - 1; i++) {
            const float k0 = st.key(i), k1 = st.key(i + 1);
            if (k0 > k1) {
                const float x0 = st.x(i), y0 = st.y(i), x1 = st.x(i + 1), y1 = st.y(i + 1);
                st.x(i) = x1; st.y(i) = y1; st.x(i + 1) = x0; st.y(i + 1) = y0;
                st.key(i) = k1; st.key(i + 1) = k0;
            }
        }
    float area = 0.f;
    const float ox = st.x(0), oy = st.y(0);
    for (int k = 0; k < cnt - 1; k++) {                                          // iou3d_kernel.cu:206-211
        float ux = sub(st.x(k), ox), uy = sub(st.y(k), oy);
        float vx = sub(st.x(k + 1), ox), vy = sub(st.y(k + 1), oy);
        area = add(area, sub(mul(ux, vy), mul(uy, vx)));
    }
    return fabsf(area) / 2.0f;
}
__device__ float box_overlap(const RBox& A, const RBox& B) { ClipPrivate st; return box_overlap_s(A, B, st); }

template <class S>
__device__ __forceinline__ float iou_bev_s(const RBox& a, const RBox& b, S& st) {         // iou3d_kernel.cu:214-221
    float sa = mul(sub(a.x2, a.x1), sub(a.y2, a.y1));
    float sb = mul(sub(b.x2, b.x1), sub(b.y2, b.y1));
    float ov = box_overlap_s(a, b, st);
    return ov / fmaxf(sub(add(sa, sb), ov), 1e-8f);
}
__device__ __forceinline__ float iou_bev(const RBox& a, const RBox& b) { ClipPrivate st; return iou_bev_s(a, b, st); }

__device__ __forceinline__ float iou_normal(const float* a, const float* b) {    // iou3d_kernel.cu:295-303
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float width = fmaxf(sub(right, left), 0.f), height = fmaxf(sub(bottom, top), 0.f);
    float interS = mul(width, height);
    float Sa = mul(sub(a[2], a[0]), sub(a[3], a[1]));
    float Sb = mul(sub(b[2], b[0]), sub(b[3], b[1]));
    return interS / fmaxf(sub(add(Sa, Sb), interS), 1e-8f);
}
