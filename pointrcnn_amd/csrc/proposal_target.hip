// proposal_target.hip -- RCNN-stage training targets on the device: RoI sampling + noise augmentation.
//
// Replaces ProposalTargetLayer.sample_rois_for_rcnn / sample_bg_inds / aug_roi_by_noise_torch / random_aug_box3d
// (lib/rpn/proposal_target_layer.py:75-300) and the iou3d_utils.boxes_iou3d_gpu calls inside them
// (lib/utils/iou3d/iou3d_utils.py:20-53).  The reference walks the batch in a Python loop; per frame it launches one
// (M x G) overlap kernel, about thirty small torch kernels with four host synchronisations (torch.nonzero, .numel()), and then,
// for each of the 64 sampled RoIs, a Python `while` loop of up to ten attempts that each build a box on the device, launch a
// 1 x 1 overlap kernel and read the IoU back (`temp_iou < pos_thresh` on the host): ~700 blocking round trips per frame.
// Here a frame is one workgroup and one launch covers the batch; nothing returns to the host.
//
//   phase 1  thread per RoI: 3-D IoU against every ground-truth box (rotated BEV overlap of iou3d_geom.h x height overlap /
//            union volume, each operation individually rounded as the torch expression does), first maximum + its index
//   phase 2  the three candidate lists (foreground >= min(REG_FG, CLS_FG); hard background in [BG_LO, BG); easy < BG_LO) in RoI
//            order, the reference's four cases, the picks
//   phase 3  thread per output slot: the accept / retry loop of aug_roi_by_noise_torch with random_aug_box3d inline
//
// Randomness is re-specified (the reference draws from numpy's and torch's global streams in a data-dependent order) with the
// counter-based generator of scene.hip, one number per (purpose, frame, position):
//   stream 10, index = RoI            fg keys: sampling without replacement = the RoIs with the smallest (key, RoI), in that order
//   stream 11, index = slot           fg sampling with replacement (frames without a background candidate)
//   stream 12 / 13, index = position  hard / easy background picks
//   stream 20, index = ((slot * 16 + attempt) * 16 + q): q = 8 keep-the-original decision (u < 0.2), q = 0 range row,
//                                     q = 1..3 position shift, q = 4..6 size scale, q = 7 rotation
// tests/golden/ref_proposal_target.py answers the reference's own random calls from the same table: its output must be -- and is
// -- reproduced bit for bit by the oracle in reference arithmetic; this kernel equals the oracle in the kernels' arithmetic
// (box trigonometry in double rounded once, division-only vertex order: DESIGN.md section 2).
#include "iou3d_geom.h"

#define PT_THREADS 256
#define PT_MAX_GT 128

__host__ __device__ __forceinline__ unsigned pt_mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ unsigned pt_rand(unsigned seed, unsigned stream, unsigned frame, unsigned i) {
    return pt_mix(i ^ pt_mix(frame * 0x9E3779B9U + pt_mix(seed + stream * 0x85EBCA6BU)));
}
__device__ __forceinline__ float pt_u01(unsigned r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ int pt_below(unsigned r, int n) { return (int)(((unsigned long long)r * (unsigned long long)n) >> 32); }

struct Box3 { float v[7]; };

__device__ __forceinline__ void bev_of(const float* b, float* o) {                  // kitti_utils.py:134-147
    o[0] = sub(b[0], b[5] / 2); o[1] = sub(b[2], b[4] / 2); o[2] = add(b[0], b[5] / 2); o[3] = add(b[2], b[4] / 2); o[4] = b[6];
}

// iou3d_utils.py:20-53 for one pair; B: the ground-truth box with its RBox already built
__device__ float iou3d_pair(const float* a, const float* b, const RBox& rb) {
    float abev[5];
    bev_of(a, abev);
    RBox ra;
    make_rbox(abev, ra);
    const float ov = far_apart(ra, rb) ? 0.0f : box_overlap(ra, rb);
    const float amin = sub(a[1], a[3]), bmin = sub(b[1], b[3]);
    const float max_of_min = amin > bmin ? amin : bmin, min_of_max = a[1] < b[1] ? a[1] : b[1];
    float h = sub(min_of_max, max_of_min);
    if (!(h > 0.0f)) h = 0.0f;
    const float ov3 = mul(ov, h);
    const float va = mul(mul(a[3], a[4]), a[5]), vb = mul(mul(b[3], b[4]), b[5]);
    float den = sub(add(va, vb), ov3);
    if (den < 1e-7f) den = 1e-7f;
    return ov3 / den;
}

struct PtParams {
    const float* roi;        // (B, M, 7)
    const float* gt;         // (B, G, gt_cols)
    int B, M, G, gt_cols, R;
    float reg_fg, cls_fg, cls_bg, cls_bg_lo;
    double hard_ratio;
    int fg_per_image, aug_times, aug_method;
    unsigned seed;
    float* rois; float* gt_of_rois; float* roi_iou; int32_t* src;      // (B,R,7) (B,R,7) (B,R) (B,R)
    float* max_overlaps; int32_t* gt_assignment;                      // (B,M)
    int32_t* counts; int32_t* status;                                 // (B,4) (B)
};

// :96-99: the ground truth of a frame without its all-zero rows at the end
__device__ __forceinline__ int pt_live_gt(const float* gt, int G, int gt_cols) {
    int ng = G;
    while (ng > 0) {
        float s = 0.f;
        for (int c = 0; c < gt_cols; c++) s = add(s, gt[(size_t)(ng - 1) * gt_cols + c]);
        if (s != 0.f) break;
        ng--;
    }
    return ng;
}

// Phase 1 on the whole chip: max IoU and its ground-truth box for every RoI (iou3d_utils.boxes_iou3d_gpu + torch.max over dim 1,
// :100-101).  The sampler kernel below is one workgroup per frame -- its branching is per frame -- and 512 RoIs x 12 boxes of
// rotated-polygon clipping on one workgroup was 400 of its 650 us.  Here a wave takes 4 RoIs, 16 lanes per RoI over the ground truth;
// lanes combine to the FIRST maximum (torch.max's rule: ties -> lowest index; a NaN only counts in column 0, as in the sequential
// `j == 0 || v > best` scan of the sampler's own reference).
__global__ __launch_bounds__(64) void pt_overlaps_kernel(const PtParams P) {
    const int b = blockIdx.y, lane = threadIdx.x;
    const int i = blockIdx.x * 4 + (lane >> 4), sub = lane & 15;
    const float* gt = P.gt + (size_t)b * P.G * P.gt_cols;
    const int ng = pt_live_gt(gt, P.G, P.gt_cols);
    if (ng == 0) return;
    const bool live = i < P.M;
    float a[7];
    for (int c = 0; c < 7; c++) a[c] = P.roi[((size_t)b * P.M + (live ? i : 0)) * 7 + c];
    float best = -INFINITY, v0 = 0.f;
    int arg = 0x7fffffff;
    for (int j = sub; j < ng; j += 16) {
        float g[7], bev[5];
        RBox rb;
        for (int c = 0; c < 7; c++) g[c] = gt[(size_t)j * P.gt_cols + c];
        bev_of(g, bev);
        make_rbox(bev, rb);
        const float v = iou3d_pair(a, g, rb);
        if (j == 0) v0 = v;
        if (v > best || (v == best && j < arg)) { best = v; arg = j; }
    }
    for (int o = 8; o > 0; o >>= 1) {                   // within the RoI's 16 lanes
        const float ob = __shfl_xor(best, o);
        const int oa = __shfl_xor(arg, o);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    v0 = __shfl(v0, lane & ~15);
    if (v0 != v0 || arg == 0x7fffffff) { best = v0; arg = 0; }      // NaN in column 0 / nothing but NaNs: what the sequential scan keeps
    if (live && sub == 0) {
        P.max_overlaps[(size_t)b * P.M + i] = best;
        P.gt_assignment[(size_t)b * P.M + i] = arg;
    }
}

__global__ __launch_bounds__(PT_THREADS) void proposal_target_kernel(const PtParams P) {
    extern __shared__ int lds_i[];                    // fg[M], hard[M], easy[M] candidate lists, then keys[M]
    __shared__ RBox s_gt[PT_MAX_GT];
    __shared__ float s_gtraw[PT_MAX_GT][7];
    __shared__ int s_ng, s_n[3], s_slots[2], s_hard_slots;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* roi = P.roi + (size_t)b * P.M * 7;
    const float* gt = P.gt + (size_t)b * P.G * P.gt_cols;
    int* fg = lds_i;
    int* hard = fg + P.M;
    int* easy = hard + P.M;
    unsigned* keys = reinterpret_cast<unsigned*>(easy + P.M);
    float* o_roi = P.rois + (size_t)b * P.R * 7;
    float* o_gt = P.gt_of_rois + (size_t)b * P.R * 7;
    float* o_iou = P.roi_iou + (size_t)b * P.R;
    int32_t* o_src = P.src + (size_t)b * P.R;
    const float fg_thresh = fminf(P.reg_fg, P.cls_fg);

    if (tid == 0) {                                   // :96-99: drop the all-zero rows at the end of the ground truth
        const int ng = pt_live_gt(gt, P.G, P.gt_cols);
        s_ng = ng;
        P.status[b] = ng == 0 ? 2 : 0;
        for (int q = 0; q < 4; q++) P.counts[b * 4 + q] = 0;
    }
    for (int t = tid; t < P.R; t += PT_THREADS) {
        o_src[t] = -1; o_iou[t] = 0.f;
        for (int c = 0; c < 7; c++) { o_roi[(size_t)t * 7 + c] = 0.f; o_gt[(size_t)t * 7 + c] = 0.f; }
    }
    __syncthreads();
    const int ng = s_ng;
    if (ng == 0) return;
    for (int j = tid; j < ng; j += PT_THREADS) {
        float bev[5];
        for (int c = 0; c < 7; c++) s_gtraw[j][c] = gt[(size_t)j * P.gt_cols + c];
        bev_of(s_gtraw[j], bev);
        make_rbox(bev, s_gt[j]);
    }
    __syncthreads();
    // phase 1 (max_overlaps / gt_assignment): pt_overlaps_kernel, launched before this one
    for (int i = tid; i < P.M; i += PT_THREADS) keys[i] = pt_rand(P.seed, 10, (unsigned)b, (unsigned)i);
    __syncthreads();
    // phase 2: candidate lists in RoI order (one thread: M <= a few thousand), then the reference's four cases
    if (tid == 0) {
        int nfg = 0, nhard = 0, neasy = 0;
        for (int i = 0; i < P.M; i++) {
            const float mo = P.max_overlaps[(size_t)b * P.M + i];
            if (mo >= fg_thresh) fg[nfg++] = i;
            if (mo < P.cls_bg_lo) easy[neasy++] = i;
            if (mo < P.cls_bg && mo >= P.cls_bg_lo) hard[nhard++] = i;
        }
        s_n[0] = nfg; s_n[1] = nhard; s_n[2] = neasy;
        P.counts[b * 4] = nfg; P.counts[b * 4 + 1] = nhard; P.counts[b * 4 + 2] = neasy;
        const int nbg = nhard + neasy;
        int fs = 0, bs = 0;
        if (nfg > 0 && nbg > 0) { fs = min(P.fg_per_image, nfg); bs = P.R - fs; }
        else if (nfg > 0) fs = P.R;
        else if (nbg > 0) bs = P.R;
        else P.status[b] = 1;
        s_slots[0] = fs; s_slots[1] = bs;
        int nh = 0;
        if (nhard > 0 && neasy > 0) nh = (int)((double)bs * P.hard_ratio);
        else if (nhard > 0) nh = bs;
        s_hard_slots = nh;
        P.counts[b * 4 + 3] = fs;
    }
    __syncthreads();
    const int nfg = s_n[0], nhard = s_n[1], neasy = s_n[2], fs = s_slots[0], bs = s_slots[1], nh = s_hard_slots;
    if (fs + bs == 0) return;
    if (nfg > 0 && nhard + neasy > 0) {
        // without replacement: candidate t goes to slot rank(t) = number of candidates with a smaller (key, RoI)
        for (int t = tid; t < nfg; t += PT_THREADS) {
            const unsigned k = keys[fg[t]];
            int rank = 0;
            for (int u = 0; u < nfg; u++) {
                const unsigned ku = keys[fg[u]];
                rank += (ku < k || (ku == k && fg[u] < fg[t])) ? 1 : 0;
            }
            if (rank < fs) o_src[rank] = fg[t];
        }
    } else if (nfg > 0) {
        for (int t = tid; t < P.R; t += PT_THREADS) o_src[t] = fg[pt_below(pt_rand(P.seed, 11, (unsigned)b, (unsigned)t), nfg)];
    }
    for (int t = tid; t < bs; t += PT_THREADS) {
        if (t < nh) o_src[fs + t] = hard[pt_below(pt_rand(P.seed, 12, (unsigned)b, (unsigned)t), nhard)];
        else o_src[fs + t] = easy[pt_below(pt_rand(P.seed, 13, (unsigned)b, (unsigned)(t - nh)), neasy)];
    }
    __syncthreads();
    // phase 3: aug_roi_by_noise_torch
    for (int t = tid; t < P.R; t += PT_THREADS) {
        const int i = o_src[t];
        const int ga = P.gt_assignment[(size_t)b * P.M + i];
        const float iou_src = P.max_overlaps[(size_t)b * P.M + i];
        float box[7], aug[7];
        for (int c = 0; c < 7; c++) { box[c] = roi[(size_t)i * 7 + c]; aug[c] = box[c]; }
        const int times = t < fs ? P.aug_times : (P.aug_times > 0 ? 1 : 0);
        float temp_iou = 0.f;
        int cnt = 0;
        bool keep = true;
        while (temp_iou < fg_thresh && cnt < times) {
            const unsigned base = (unsigned)((t * 16 + cnt) * 16);
            if (pt_u01(pt_rand(P.seed, 20, (unsigned)b, base + 8)) < 0.2f) {
                for (int c = 0; c < 7; c++) aug[c] = box[c];
                keep = true;
            } else {
                float ps[3], hs[3], ar;
                if (P.aug_method == 0) {                     // 'multiple' (:262-277)
                    const double rc[5][3] = { {0.2, 0.1, 3.14159265358979323846 / 12}, {0.3, 0.15, 3.14159265358979323846 / 12},
                                              {0.5, 0.15, 3.14159265358979323846 / 9}, {0.8, 0.15, 3.14159265358979323846 / 6},
                                              {1.0, 0.15, 3.14159265358979323846 / 3} };
                    const int idx = pt_below(pt_rand(P.seed, 20, (unsigned)b, base), 5);
                    for (int c = 0; c < 3; c++) ps[c] = mul(sub(pt_u01(pt_rand(P.seed, 20, (unsigned)b, base + 1 + c)), 0.5f) / 0.5f, (float)rc[idx][0]);
                    for (int c = 0; c < 3; c++) hs[c] = add(mul(sub(pt_u01(pt_rand(P.seed, 20, (unsigned)b, base + 4 + c)), 0.5f) / 0.5f, (float)rc[idx][1]), 1.0f);
                    ar = mul(sub(pt_u01(pt_rand(P.seed, 20, (unsigned)b, base + 7)), 0.5f) / 0.5f, (float)rc[idx][2]);
                } else {                                     // 'single' (:254-260)
                    for (int c = 0; c < 3; c++) ps[c] = sub(pt_u01(pt_rand(P.seed, 20, (unsigned)b, base + 1 + c)), 0.5f);
                    for (int c = 0; c < 3; c++) hs[c] = add(sub(pt_u01(pt_rand(P.seed, 20, (unsigned)b, base + 4 + c)), 0.5f) / (float)(0.5 / 0.15), 1.0f);
                    ar = sub(pt_u01(pt_rand(P.seed, 20, (unsigned)b, base + 7)), 0.5f) / (float)(0.5 / (3.14159265358979323846 / 12));
                }
                for (int c = 0; c < 3; c++) { aug[c] = add(box[c], ps[c]); aug[3 + c] = mul(box[3 + c], hs[c]); }
                aug[6] = add(box[6], ar);
                keep = false;
            }
            temp_iou = iou3d_pair(aug, s_gtraw[ga], s_gt[ga]);
            cnt++;
        }
        for (int c = 0; c < 7; c++) { o_roi[(size_t)t * 7 + c] = aug[c]; o_gt[(size_t)t * 7 + c] = s_gtraw[ga][c]; }
        o_iou[t] = (cnt == 0 || keep) ? iou_src : temp_iou;
    }
}

PRCNN_API int prcnn_proposal_target_sample(const float* roi_boxes3d, const float* gt_boxes3d, int B, int M, int G, int gt_cols,
                                           int roi_per_image, const double* cfg6, int aug_times, int aug_method, uint32_t seed,
                                           float* rois, float* gt_of_rois, float* roi_iou, int32_t* src, float* max_overlaps,
                                           int32_t* gt_assignment, int32_t* counts, int32_t* status, prcnn_stream_t stream) {
    PRCNN_REQUIRE(B >= 0 && M > 0 && G > 0 && G <= PT_MAX_GT && gt_cols >= 7 && roi_per_image > 0,
                  "prcnn_proposal_target_sample: bad shape B=%d M=%d G=%d (<= %d) gt_cols=%d R=%d", B, M, G, PT_MAX_GT, gt_cols, roi_per_image);
    PRCNN_REQUIRE(M <= 8192, "prcnn_proposal_target_sample: at most 8192 RoIs per frame (got %d)", M);
    PRCNN_REQUIRE(aug_times >= 0 && aug_times <= 16 && (aug_method == 0 || aug_method == 1),
                  "prcnn_proposal_target_sample: aug_times in 0..16, aug_method 0 ('multiple') or 1 ('single')");
    if (B == 0) return PRCNN_OK;
    PRCNN_REQUIRE(roi_boxes3d && gt_boxes3d && cfg6 && rois && gt_of_rois && roi_iou && src && max_overlaps && gt_assignment && counts && status,
                  "prcnn_proposal_target_sample: null pointer");
    PtParams P;
    P.roi = roi_boxes3d; P.gt = gt_boxes3d; P.B = B; P.M = M; P.G = G; P.gt_cols = gt_cols; P.R = roi_per_image;
    // thresholds meet float32 overlaps (torch casts the Python scalar to the tensor's dtype); the two ratios stay DOUBLE: the reference
    // computes np.round(FG_RATIO * ROI_PER_IMAGE) and int(bg_rois_per_this_image * HARD_BG_RATIO) in Python doubles (0.7f * 10 -> 6, 0.7 * 10 -> 7)
    P.reg_fg = (float)cfg6[0]; P.cls_fg = (float)cfg6[1]; P.cls_bg = (float)cfg6[2]; P.cls_bg_lo = (float)cfg6[3]; P.hard_ratio = cfg6[5];
    P.fg_per_image = (int)nearbyint(cfg6[4] * (double)roi_per_image);                 // np.round
    P.aug_times = aug_times; P.aug_method = aug_method; P.seed = seed;
    P.rois = rois; P.gt_of_rois = gt_of_rois; P.roi_iou = roi_iou; P.src = src; P.max_overlaps = max_overlaps;
    P.gt_assignment = gt_assignment; P.counts = counts; P.status = status;
    hipLaunchKernelGGL(pt_overlaps_kernel, dim3(prcnn_divup(M, 4), B), dim3(64), 0, (hipStream_t)stream, P);
    // keys / fg / hard / easy lists: 16 bytes per RoI of dynamic LDS (128 KB at the 8192-RoI cap) next to ~12 KB static
    const size_t lds = (size_t)M * 4 * sizeof(int);
    static PrcnnLdsLimit attr;
    PRCNN_REQUIRE(lds <= 48 * 1024 || attr.raise((const void*)proposal_target_kernel, 8192 * 4 * (int)sizeof(int)),
                  "prcnn_proposal_target_sample: cannot raise the dynamic LDS limit for M=%d", M);
    hipLaunchKernelGGL(proposal_target_kernel, dim3(B), dim3(PT_THREADS), lds, (hipStream_t)stream, P);
    PRCNN_LAUNCH_CHECK("prcnn_proposal_target_sample");
    return PRCNN_OK;
}

// iou3d_utils.boxes_iou3d_gpu as one kernel: out (Na, Nb)
__global__ __launch_bounds__(256) void boxes_iou3d_kernel(const float* __restrict__ a, int Na, const float* __restrict__ b, int Nb,
                                                          float* __restrict__ out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)Na * Nb) return;
    const int i = (int)(e / Nb), j = (int)(e - (long)i * Nb);
    float bb[7], bev[5];
    for (int c = 0; c < 7; c++) bb[c] = b[(size_t)j * 7 + c];
    bev_of(bb, bev);
    RBox rb;
    make_rbox(bev, rb);
    float aa[7];
    for (int c = 0; c < 7; c++) aa[c] = a[(size_t)i * 7 + c];
    out[e] = iou3d_pair(aa, bb, rb);
}

PRCNN_API int prcnn_boxes_iou3d(const float* a, int Na, const float* b, int Nb, float* out, prcnn_stream_t stream) {
    PRCNN_REQUIRE(Na >= 0 && Nb >= 0, "prcnn_boxes_iou3d: bad shape");
    if (Na == 0 || Nb == 0) return PRCNN_OK;
    PRCNN_REQUIRE(a && b && out, "prcnn_boxes_iou3d: null pointer");
    hipLaunchKernelGGL(boxes_iou3d_kernel, dim3(prcnn_divup((long)Na * Nb, 256)), dim3(256), 0, (hipStream_t)stream, a, Na, b, Nb, out);
    PRCNN_LAUNCH_CHECK("prcnn_boxes_iou3d");
    return PRCNN_OK;
}
