/*
 * prcnn_pointops.h -- C ABI of libprcnn_pointops.so, the MI355X (gfx950) implementation of
 * PointRCNN's point-ops hot path (PointNet++ set-abstraction / feature-propagation operators,
 * roipool3d, iou3d).
 *
 * This is the drop-in boundary: what the reference binds through its three pybind modules
 * (`pointnet2_cuda` [un-vendored submodule, .gitmodules:1-4], `iou3d_cuda`
 * lib/utils/iou3d/src/iou3d.cpp:174-179, `roipool3d_cuda` lib/utils/roipool3d/src/roipool3d.cpp:198-203)
 * is exactly this set of entry points.  INTEGRATION.md shows the ctypes binding
 * (pointrcnn_amd/_cabi.py is that binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in `_host` or the FUNCTION is named
 *     prcnn_host_* (host pointers only, no HIP call inside: safe in forked dataloader workers);
 *   - tensors are dense row-major fp32 / int32 / int64 exactly as the reference op surface lays
 *     them out; the caller allocates every output (reference: iou3d_utils.py:14,32,68;
 *     roipool3d_utils.py:21-23); callees keep no reference to any buffer after return;
 *   - `stream` is a hipStream_t (NULL = the null stream).  All work is enqueued on it and the call
 *     returns without synchronising (the reference launches on the legacy default stream and
 *     nms_* blocks on a D2H copy: iou3d.cpp:93-94 -- the sync lives in the Python shim only);
 *   - return value: 0 on success, <0 on error (never exit(): contrast iou3d.cpp:13-21);
 *     prcnn_last_error() returns a thread-local message for the last failing call;
 *   - every entry point is re-entrant and may be called concurrently from several host threads, each on its own
 *     device (the reference's nn.DataParallel convention).  The only process-wide state is a per-(kernel, device)
 *     "dynamic-LDS limit already raised" bit, updated atomically (csrc/common.h PrcnnLdsLimit).
 *
 * Arithmetic contract (shared bit-for-bit with oracle/prcnn_oracle.c):
 *   - squared distances are ((dx*dx + dy*dy) + dz*dz) with individually rounded fp32 operations (no FMA; the library is built
 *     with -ffp-contract=off);
 *   - box trigonometry of roipool3d / pts_in_boxes3d / labels / GT-aug / overlap / IoU / NMS / proposal NMS / RoI sampling is the
 *     REFERENCE'S HOST ARITHMETIC (oracle trig_mode 2): glibc 2.35's float sinf / cosf / atan2f restated operation by operation
 *     (csrc/ref_trig.h; prcnn_ref_trig exposes it), because the reference calls cos / sin / atan2 on floats
 *     (iou3d_kernel.cu:56,104-106,135-136; roipool3d.cpp:89); polygon vertices are ordered by that atan2f, as the reference's
 *     bubble sort orders them (iou3d_kernel.cu:188-196).  Results are bit-identical to the reference's own sources compiled for
 *     the host (oracle/_ref, oracle trig_mode 0), near-threshold pairs and on-the-face points included
 *     (tests/test_gpu_parity_residuals.py);
 *   - decode_bbox_target and the canonical transform of roipool3d keep cos / sin evaluated in double and rounded once to fp32
 *     (oracle trig_mode 1): their reference is torch's vectorised cos / sin, pinned to 2.4e-7, not glibc's scalar routines;
 *   - MLP layers: fp32 accumulation on the matrix pipe, either fp32 MFMA or the exact three-way bf16 split with six products per
 *     fp32 product (fp32-grade: |err| <= ~3e-7 * sum_k |x_k||w_k| per output element, measured against float64, for both).
 */
#ifndef PRCNN_POINTOPS_H
#define PRCNN_POINTOPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* prcnn_stream_t; /* hipStream_t */

#define PRCNN_OK 0
#define PRCNN_EINVAL (-1)       /* bad argument (shape, null pointer, unsupported size) */
#define PRCNN_EHIP (-2)         /* HIP runtime / launch failure */
#define PRCNN_EUNSUPPORTED (-3) /* valid request this build has no kernel for */

int prcnn_abi_version(void);   /* 9: + prcnn_fps_mode, prcnn_ball_query_arith, prcnn_three_nn_arith (comparison mode: the squared distance as nvcc contracts the
                                 * upstream expression); prcnn_nms_workspace_bytes grew by one flag byte per 64 x 64 tile; 8: + prcnn_mlp_group_split;
                                 * 7: split-bf16 chain entry points take the fp32 pack images too (fp32 recomputation of rows with non-finite values); 6: + prcnn_fps_status, training-mode SharedMLP (prcnn_train_*), prcnn_boxes_iou3d, prcnn_proposal_target_sample, prcnn_ref_trig (box trigonometry = the reference's host libm, bit for bit); 5: + prcnn_gt_aug_edit;
                                 * 4: + prcnn_host_* (host twins of the reference's *_cpu entry points), prcnn_build_id,
                                 * prcnn_fps_order (upstream tie order), prcnn_rpn_labels,
                                 * prcnn_ball_query2_grid takes xyz (dense-frame scan fallback);
                                 * 3: + padding-free grouping (rows_dev / groups_dev, prcnn_group_compact, ...), RoI duplicate
                                 * elimination (seg_cnt / seg_rows, distinct, valid_n), prcnn_scene_prepare */
const char* prcnn_last_error(void);
/* hex digest of the kernel sources + compile flags this library was built from (the Python binding compares it with the
 * sources it sits next to and refuses / rebuilds a stale library instead of silently loading it) */
const char* prcnn_build_id(void);

/* ---------------------------------------------------------------------------------------------
 * PointNet++ operators.  Replace pointnet2_cuda.* [UPSTREAM sshaoshuai/Pointnet2.PyTorch, not in
 * tree]; reference call sites: lib/net/pointnet2_msg.py:27-34,44,61,66-68, lib/net/rcnn_net.py:33-41.
 * ------------------------------------------------------------------------------------------- */

/* furthest_point_sampling_wrapper(B,N,npoint,xyz,temp,idx): xyz (B,N,3) -> idx (B,npoint) i32.
 * Start index 0, ties -> lowest point index.  `tmp` (B,N) 4-byte scratch (upstream's `temp` argument):
 *   N > 16384           required: HBM-resident running min-distances;
 *   2048 < N <= 16384   optional: when given, selects the spatially pruned kernel (Morton pre-sort into `tmp`, exact
 *                       bounding-box skip; bit-identical results); NULL = the register-resident kernel;
 *   N <= 2048           ignored.
 * Domain: FINITE coordinates.  The kernels order distances through integer compares of their bit patterns and bare
 * v_min/v_max (the file is built with -ffinite-math-only); a cloud holding NaN / Inf has no farthest point under any rule
 * and the indices returned for it are unspecified (always within [0, N)).  Callers with unvalidated input filter first
 * (prcnn_scene_prepare drops non-finite raw points). */
int prcnn_fps(const float* xyz, int B, int N, int npoint, float* tmp, int32_t* idx, prcnn_stream_t stream);

/* N > 16384 runs one frame on several cooperating workgroups whose wait for each other is bounded (a hang would take the
 * GPU down); a workgroup that gave up fills the rest of its frame's output with -1 and marks a host-visible word.
 * prcnn_fps_status() returns PRCNN_EHIP (and clears the mark) if that happened since the last check, PRCNN_OK otherwise;
 * it does not synchronise -- call it after the stream has.  prcnn_fps itself checks the mark before every such launch. */
int prcnn_fps_status(void);

/* Same, with a selectable rule for ties among equal running min-distances (they only occur on clouds with duplicate
 * points or exact lattices):
 *   PRCNN_FPS_ORDER_CANONICAL  lowest point index (== prcnn_fps; the contract every other test pins);
 *   PRCNN_FPS_ORDER_UPSTREAM   the order the upstream CUDA kernel's thread layout produces: argmin (k mod T, k) with
 *                              T = min(1024, largest power of two <= N) -- for index-by-index
 *                              comparison against an upstream build; `tmp` (B,N) is required; not a fast path. */
#define PRCNN_FPS_ORDER_CANONICAL 0
#define PRCNN_FPS_ORDER_UPSTREAM 1
int prcnn_fps_order(const float* xyz, int B, int N, int npoint, int order, float* tmp, int32_t* idx, prcnn_stream_t stream);

/* COMPARISON MODE of the three index operators: selectable squared-distance arithmetic (and, for FPS, tie order).
 *   PRCNN_ARITH_CANONICAL  three individually rounded products summed left to right -- the contract of prcnn_fps / prcnn_ball_query* /
 *                          prcnn_three_nn* and of every test that pins an index;
 *   PRCNN_ARITH_UPSTREAM   fma(dz,dz, fma(dy,dy, dx*dx)): what nvcc's default -fmad=true makes of the upstream kernels'
 *                          (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1)  (sshaoshuai/Pointnet2.PyTorch, not in the reference tree:
 *                          .gitmodules:1-4) -- for index-by-index comparison against an upstream build, and for measuring how often
 *                          the two arithmetics disagree (tools/arith_disagreement.py).
 * Plain kernels (one workgroup per frame / one thread per query), bit-identical to the oracle's prcnn_cpu_fps_mode /
 * prcnn_cpu_ball_query_arith / prcnn_cpu_three_nn_arith in both arithmetics; not a fast path, used by no module.
 * prcnn_fps_mode: `tmp` (B,N) f32 is required.  prcnn_three_nn_arith returns SQUARED distances and no weights. */
#define PRCNN_ARITH_CANONICAL 0
#define PRCNN_ARITH_UPSTREAM 1
int prcnn_fps_mode(const float* xyz, int B, int N, int npoint, int order, int arith, float* tmp, int32_t* idx, prcnn_stream_t stream);
int prcnn_ball_query_arith(const float* xyz, const float* new_xyz, int B, int N, int M, float radius, int nsample, int arith,
                           int32_t* idx, prcnn_stream_t stream);
int prcnn_three_nn_arith(const float* unknown, const float* known, int B, int n, int m, int arith, float* dist2, int32_t* idx,
                         prcnn_stream_t stream);

/* gather_points_wrapper(B,C,N,npoint,feat,idx,out): out[b,c,m] = feat[b,c,idx[b,m]] */
int prcnn_gather(const float* feat, const int32_t* idx, int B, int C, int N, int M, float* out, prcnn_stream_t stream);
/* gather_points_grad_wrapper: grad_feat (B,C,N) += scatter(grad_out (B,C,M)); grad_feat pre-zeroed by caller */
int prcnn_gather_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M, float* grad_feat,
                      prcnn_stream_t stream);

/* channels-last twin of gather: out[b,m,0:C] = in_cl[b, idx[b,m], 0:C]; in_cl rows have stride ld_in floats.
 * (new_xyz = xyz[fps_idx] without the two transposes upstream needs around gather_operation.) */
int prcnn_gather_rows(const float* in_cl, int ld_in, const int32_t* idx, int B, int N, int M, int C, float* out,
                      prcnn_stream_t stream);

/* ball_query_wrapper(B,N,M,radius,nsample,new_xyz,xyz,idx): idx (B,M,nsample) i32; first <=nsample
 * points with d2 < radius^2 in ascending index order, padded with the first hit; no hit -> zeros. */
int prcnn_ball_query(const float* xyz, const float* new_xyz, int B, int N, int M, float radius, int nsample,
                     int32_t* idx, prcnn_stream_t stream);
/* Two radii in ONE scan of xyz (the MSG level's two groupers share new_xyz): same results as two
 * prcnn_ball_query calls. */
int prcnn_ball_query2(const float* xyz, const float* new_xyz, int B, int N, int M, float radius_a, int nsample_a,
                      int32_t* idx_a, float radius_b, int nsample_b, int32_t* idx_b, prcnn_stream_t stream);

/* group_points_wrapper(B,C,N,npoint,nsample,feat,idx,out): out[b,c,m,s] = feat[b,c,idx[b,m,s]] */
int prcnn_group(const float* feat, const int32_t* idx, int B, int C, int N, int M, int nsample, float* out,
                prcnn_stream_t stream);
int prcnn_group_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M, int nsample,
                     float* grad_feat, prcnn_stream_t stream);

/* three_nn_wrapper(B,n,m,unknown,known,dist2,idx): 3 nearest known points per unknown point, SQUARED
 * distances (the Python wrapper takes the sqrt, as upstream does), strict-< insertion (ties keep the
 * earlier index).  `weight` (B,n,3), optional (NULL to skip): the FP module's normalised
 * inverse-distance weights w_k = r_k / (r_0+r_1+r_2), r_k = 1/(sqrt(dist2_k)+1e-8). */
int prcnn_three_nn(const float* unknown, const float* known, int B, int n, int m, float* dist2, int32_t* idx,
                   float* weight, prcnn_stream_t stream);

/* three_interpolate_wrapper(B,C,m,n,feat,idx,weight,out): out[b,c,i] = (w0*f[i0] + w1*f[i1]) + w2*f[i2] */
int prcnn_three_interp(const float* feat, const int32_t* idx, const float* weight, int B, int C, int m, int n,
                       float* out, prcnn_stream_t stream);
/* three_interpolate_grad_wrapper: grad_feat (B,C,m) += scatter(grad_out (B,C,n) * weight); grad_feat pre-zeroed by the caller.
 * workspace: (B*m*C) floats of scratch or NULL.  With a workspace the scatter runs through a channels-last accumulator (the 64
 * lanes of a wave add to 64 consecutive channels of one known point: one or two cache lines per atomic instruction instead of
 * 64) and is transposed into grad_feat afterwards; NULL selects the direct kernel. */
int prcnn_three_interp_grad(const float* grad_out, const int32_t* idx, const float* weight, int B, int C, int n,
                            int m, float* grad_feat, float* workspace, prcnn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused per-point MLP layers (new; replaces the SharedMLP = Conv2d1x1+BN+ReLU (+max_pool2d) chain
 * that upstream runs through cuDNN on a materialised (B,C+3,npoint,nsample) tensor; specs at
 * lib/net/pointnet2_msg.py:20-45, lib/net/rpn.py:20-46, lib/net/rcnn_net.py:23-41).
 *
 * All activations are CHANNELS-LAST rows: a row is one point (or one (centroid,sample) pair) and
 * holds its channels contiguously.  One call = one layer:
 *      out[row, col_off + n] = act( sum_k A[row,k] * W[n,k] + bias[n] ),   n < Nout
 * computed with fp32-input MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
 * BatchNorm (eval) is folded into W/bias by the caller.  `wpack` is W re-laid for the MFMA operand
 * order by prcnn_pack_weight.  If pool_ns > 0 the epilogue max-reduces every pool_ns consecutive
 * rows (max_pool2d(kernel=[1,nsample])) and writes rows/pool_ns rows; pool_ns must be 16, 32 or 64.
 * `ld_*` are row strides in floats.
 * ------------------------------------------------------------------------------------------- */

/* floats needed for the packed image of a (Nout,K) weight */
size_t prcnn_wpack_floats(int Nout, int K);
/* w: (Nout,K) row-major (torch conv weight).  k_rot: the first k_rot input channels of W are moved
 * to the END of the packed K order (grouped rows are laid out [features(C), dxyz(3)] while torch's
 * QueryAndGroup order is [dxyz(3), features(C)]: pass k_rot=3 for those layers, 0 otherwise). */
int prcnn_pack_weight(const float* w, int Nout, int K, int k_rot, float* wpack, prcnn_stream_t stream);

/* A = in (rows, K), row stride ld_in.
 * rows_dev (device i32, may be NULL) / rows_unit: DEVICE-side row count -- the kernel processes
 * min(rows, *rows_dev * rows_unit) rows and the launch (sized for `rows`) exits early past that; used with the
 * compact group lists of prcnn_group_compact, whose lengths are only known on the device. */
int prcnn_mlp_rows(const float* in, int ld_in, int64_t rows, int K, const float* wpack, const float* bias, int Nout,
                   int relu, float* out, int ld_out, int col_off, int pool_ns, const int32_t* rows_dev, int rows_unit,
                   const int32_t* seg_cnt, int seg_rows, prcnn_stream_t stream);
/* seg_cnt (device i32, may be NULL) / seg_rows: SEGMENT-PREFIX LIVE ROWS -- rows come in segments of seg_rows rows
 * (a multiple of 128 dividing rows) and only the first seg_cnt[s] rows of segment s are ever read by anyone: roipool3d
 * pads an RoI holding fewer points than it samples with copies of its first rows (prcnn_roipool3d_canonical's
 * `distinct` output).  128-row tiles lying entirely in the dead tail of their segment are skipped; their output rows are
 * left unwritten. */

/* A row (b,m,s) = [ feat_cl[b, idx[b,m,s], 0:C],  xyz[b, idx[b,m,s]] - new_xyz[b,m] ]  (K = C+3;
 * C may be 0 with feat_cl NULL).  new_xyz NULL => GroupAll semantics (no centroid subtraction).
 * rows = B*M*nsample.
 * HOISTED FIRST LAYER (act_wx, act_bias non-NULL; both NULL = the classic form above): grouping is a linear gather
 * and the first conv of a SharedMLP is linear, so W.[feat[idx]; dxyz] = (W_f.feat)[idx] + W_x.dxyz.  The caller
 * computes Z = W_f.feat once per SOURCE point (N rows instead of M*nsample) with prcnn_mlp_rows and passes it as
 * feat_cl (C = width of that first layer, C % 4 == 0); act_wx is (C,3) row-major = the dxyz columns of the first
 * layer's weight, act_bias (C, padded to x4) its bias.  The A row is then relu(Z[idx] + act_wx.dxyz + act_bias),
 * K = C, and wpack/bias describe the SECOND layer.  Same result up to fp32 reassociation. */
int prcnn_mlp_group(const float* xyz, const float* new_xyz, const int32_t* idx, const float* feat_cl, int ld_feat,
                    int B, int N, int M, int nsample, int C, const float* act_wx, const float* act_bias,
                    const float* wpack, const float* bias, int Nout, int relu, float* out, int ld_out, int col_off,
                    int pool_ns, const int32_t* groups_dev, prcnn_stream_t stream);   /* groups_dev: device count of groups
                    (<= B*M) actually present, or NULL -- see prcnn_mlp_rows / prcnn_group_compact */

/* A row (b,i) = [ sum_j w3[b,i,j] * known_cl[b, idx3[b,i,j], 0:C2],  skip_cl[b,i,0:C1] ]  (K = C2+C1;
 * C1 may be 0 with skip_cl NULL).  rows = B*n.
 * HOISTED FIRST LAYER (act_bias non-NULL, requires C1 == 0): interpolation is linear, so W.interp(x) = interp(W.x).
 * The caller computes Y = W.known once per KNOWN point (m rows instead of n) and passes it as known_cl (C2 = width
 * of that layer); the A row is relu(interp(Y) + act_bias) and wpack/bias describe the second layer. */
int prcnn_mlp_interp(const float* known_cl, int ld_known, const int32_t* idx3, const float* w3, const float* skip_cl,
                     int ld_skip, int B, int n, int m, int C2, int C1, const float* act_bias, const float* wpack,
                     const float* bias, int Nout, int relu, float* out, int ld_out, int col_off,
                     prcnn_stream_t stream);

/* Hoisted FP first layer WITH skip features: out[row,n] = act( sum_k in[row,k]*W_b[n,k] + bias[n]
 *                                                             + sum_j w3[row,j] * y_cl[b*m + idx3[row,j], n] )
 * where `in` are the skip features (K = C1), W_b the skip columns of the first layer's weight and
 * y_cl = W_a.known (B*m rows, ld_y floats apart) the interpolated part computed per known point.  rows = B*n. */
int prcnn_mlp_rows_addinterp(const float* in, int ld_in, int K, const float* wpack, const float* bias, int Nout,
                             int relu, const float* y_cl, int ld_y, const int32_t* idx3, const float* w3, int B, int n,
                             int m, float* out, int ld_out, int col_off, prcnn_stream_t stream);

/* Split-bf16 VARIANT of the two plain-row layer calls above (never the default arithmetic; the caller opts in per call).
 * Every fp32 operand is cut exactly into three bf16 pieces (x = x0 + x1 + x2) and each fp32 product is rebuilt from `terms`
 * bf16 MFMA products with fp32 accumulation: terms = 6 keeps everything above 2^-24 |x||w| (fp32-grade results), terms = 3
 * everything above 2^-16 |x||w|.  wsplit: prcnn_wsplit_bytes(Nout, K) bytes written by prcnn_pack_weight_split from the
 * (Nout, K) row-major fp32 weight.  The split kernel needs K % 32 == 0 and 16-byte aligned input rows; any other shape runs
 * the fp32 kernel on `wpack`, exactly as prcnn_mlp_rows / prcnn_mlp_rows_addinterp would.  Non-finite values: a wave that finds
 * a non-finite accumulator (an infinite or NaN input value, an overflowing product) recomputes its rows with fp32 MFMAs on
 * `wpack` -- those outputs are the fp32 kernels' own (inf stays inf, NaN only where IEEE arithmetic gives NaN).
 * pool_ns / rows_dev / rows_unit / seg_cnt / seg_rows: as for prcnn_mlp_rows.  A supported shape runs the split kernel whatever
 * its row count (the arithmetic of a layer does not depend on the batch size).
 * prcnn_pack_weight_split: chain = 0 writes the image of the two layer calls, chain = 1 the image of prcnn_mlp_chain_rows_split
 * (same size).  prcnn_mlp_chain_rows_split: the two-layer plain-row chain (prcnn_mlp_chain_rows with nlayers = 2) for K = 128,
 * nout = {128, 1} or {128, 65..128} -- the RPN heads; wchain / bias / nout / relu are HOST arrays of length 2, wchain[l] the
 * chain = 1 image of layer l (wchain[1] unused when nout[1] == 1: that output is a dot product on wpack[1]); wpack: HOST array
 * of the two layers' fp32 prcnn_pack_weight images (the single-channel output and the fp32 recomputation of rows with
 * non-finite values read them).  Any other shape: PRCNN_EUNSUPPORTED, issue prcnn_mlp_chain_rows.
 * (The reference computes these layers as fp32 cuDNN convolutions, [U] pytorch_utils.py SharedMLP) */
size_t prcnn_wsplit_bytes(int Nout, int K);
int prcnn_pack_weight_split(const float* w, int Nout, int K, int chain, void* wsplit, prcnn_stream_t stream);
int prcnn_mlp_rows_split(const float* in, int ld_in, int64_t rows, int K, const float* wpack, const void* wsplit, int terms,
                         const float* bias, int Nout, int relu, float* out, int ld_out, int col_off, int pool_ns,
                         const int32_t* rows_dev, int rows_unit, const int32_t* seg_cnt, int seg_rows, prcnn_stream_t stream);
int prcnn_mlp_chain_rows_split(const float* in, int ld_in, int64_t rows, int K, const void* const* wchain, const float* const* wpack,
                               const float* const* bias, const int* nout, const int* relu, int terms, float* out, int ld_out,
                               int col_off, prcnn_stream_t stream);
/* hoisted FP0 on the split chain kernel: rows relu(interp(known_cl) + act_bias), C2 = 128 and no skip features, through ONE
 * 128 -> 128 layer (prcnn_mlp_chain_interp with nlayers = 1, C1 = 0, act_bias given); wchain: the chain = 1 image, wpack: the
 * layer's fp32 prcnn_pack_weight image (rows with non-finite values).  Other shapes: PRCNN_EUNSUPPORTED. */
int prcnn_mlp_chain_interp_split(const float* known_cl, int ld_known, const int32_t* idx3, const float* w3, int B, int n, int m,
                                 int C2, const float* act_bias, const void* wchain, const float* wpack, const float* bias, int Nout,
                                 int relu, int terms, float* out, int ld_out, int col_off, prcnn_stream_t stream);
int prcnn_mlp_rows_addinterp_split(const float* in, int ld_in, int K, const float* wpack, const void* wsplit, int terms,
                                   const float* bias, int Nout, int relu, const float* y_cl, int ld_y, const int32_t* idx3,
                                   const float* w3, int B, int n, int m, float* out, int ld_out, int col_off,
                                   prcnn_stream_t stream);
/* prcnn_mlp_group's HOISTED form (act_wx / act_bias given) on the split-bf16 layer kernel: the gathered row relu(Z[idx] + act_wx.dxyz +
 * act_bias) is activated on its way into the split.  wsplit: the layer's prcnn_pack_weight_split image (chain = 0), terms 3 / 6; wpack: its
 * fp32 image (rows holding inf / NaN are redone on the fp32 pipe).  Needs C % 32 == 0 and 16-byte aligned feature rows; any other shape
 * (or act_wx NULL) runs the fp32 layer kernel exactly as prcnn_mlp_group does.  Replaces the same reference code as prcnn_mlp_group. */
int prcnn_mlp_group_split(const float* xyz, const float* new_xyz, const int32_t* idx, const float* feat_cl, int ld_feat, int B, int N,
                          int M, int nsample, int C, const float* act_wx, const float* act_bias, const float* wpack,
                          const void* wsplit, int terms, const float* bias, int Nout, int relu, float* out, int ld_out, int col_off,
                          int pool_ns, const int32_t* groups_dev, prcnn_stream_t stream);

/* Register-resident layer CHAIN: up to 3 consecutive layers (a whole SharedMLP) in ONE kernel; one wave owns 32
 * rows and carries them through every layer inside the register file (the MFMA accumulator layout of layer l is
 * the B-operand layout of layer l+1), so intermediate activations touch neither LDS nor HBM and a gathered /
 * interpolated input row is built once.  Same arithmetic as the per-layer calls (fp32 MFMA, fp32 accumulate).
 * wpack[l], bias[l], nout[l], relu[l] are HOST arrays of length nlayers; every bias[l] (device, may be NULL) must
 * hold 32*ceil(nout[l]/32) floats (zero padded).  Limits: nout[l] <= 128; pool_ns in {0,16,32} (group variant:
 * pool_ns == nsample or 0).  Only a fixed set of width combinations is instantiated: ask
 * prcnn_mlp_chain_supported first, or handle PRCNN_EUNSUPPORTED by issuing the per-layer calls.
 * mode: 0 = rows, 1 = group, 2 = interp.  act_wx / act_bias: hoisted first layer, as for prcnn_mlp_group /
 * prcnn_mlp_interp (the chain then starts at the second layer).
 * One more shape (round 2): mode 1, TWO layers wider than 128 (nout[l] <= 384 / 512), un-pooled, nsample 1, hoisted first
 * layer with C <= 256 -- the flat row lists of the wide set-abstraction levels -- runs as a 32-row two-layer stack through
 * LDS (same bits as the per-layer calls); prcnn_mlp_chain_supported answers 1 for it from the widths alone, the call itself
 * returns PRCNN_EUNSUPPORTED if the other conditions are not met. */
int prcnn_mlp_chain_supported(int mode, int nlayers, const int* nout, int pool_ns);
int prcnn_mlp_chain_rows(const float* in, int ld_in, int64_t rows, int K, int nlayers, const float* const* wpack,
                         const float* const* bias, const int* nout, const int* relu, float* out, int ld_out,
                         int col_off, int pool_ns, const int32_t* seg_cnt, int seg_rows, prcnn_stream_t stream);
                         /* seg_cnt / seg_rows: as for prcnn_mlp_rows */
int prcnn_mlp_chain_group(const float* xyz, const float* new_xyz, const int32_t* idx, const float* feat_cl,
                          int ld_feat, int B, int N, int M, int nsample, int C, const float* act_wx,
                          const float* act_bias, int nlayers,
                          const float* const* wpack, const float* const* bias, const int* nout, const int* relu,
                          float* out, int ld_out, int col_off, int pool_ns, const int32_t* groups_dev, prcnn_stream_t stream);
int prcnn_mlp_chain_interp(const float* known_cl, int ld_known, const int32_t* idx3, const float* w3,
                           const float* skip_cl, int ld_skip, int B, int n, int m, int C2, int C1,
                           const float* act_bias, int nlayers, const float* const* wpack, const float* const* bias, const int* nout, const int* relu,
                           float* out, int ld_out, int col_off, prcnn_stream_t stream);

/* out[r, col_off + c] = max over ns consecutive rows of in (generic nsample fallback for pooling) */
int prcnn_maxpool_rows(const float* in, int ld_in, int64_t rows_out, int ns, int C, float* out, int ld_out,
                       int col_off, prcnn_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * roipool3d.  Replaces roipool3d_cuda.forward (roipool3d.cpp:48-79 -> roipool3d_kernel.cu:209-237).
 * xyz (B,N,3), boxes3d (B,M,7) [x,y(bottom),z,h,w,l,ry] ALREADY enlarged (roipool3d_utils.py:19),
 * feat (B,N,C) -> pooled (B,M,S,3+C), empty (B,M) i32.  One pass, no B*N*M temporary, no allocation.
 * Every output element is written (empty boxes get zero rows), so `pooled` need not be pre-zeroed.
 * ------------------------------------------------------------------------------------------- */
int prcnn_roipool3d(const float* xyz, const float* boxes3d, const float* feat, int B, int N, int M, int C, int S,
                    float* pooled, int32_t* empty, prcnn_stream_t stream);
/* The same operator with a caller-provided scratch buffer (>= prcnn_roipool3d_work_bytes(B, N) device bytes; 0 = this N has no
 * binned form): the frame's points are first bucketed into x-z bins, and each RoI tests only the points of the bins its footprint
 * touches instead of all N (identical selection: same membership test, ascending index order).  work == NULL or too small:
 * the linear scan of prcnn_roipool3d.  Worth it from a handful of RoIs per frame upwards. */
size_t prcnn_roipool3d_work_bytes(int B, int N);
int prcnn_roipool3d_ws(const float* xyz, const float* boxes3d, const float* feat, int B, int N, int M, int C, int S, float* pooled,
                       int32_t* empty, void* work, size_t work_bytes, prcnn_stream_t stream);

/* point-in-box flags on the device: flags (M,N) i32 (device twin of roipool3d.cpp:97-125) */
int prcnn_pts_in_boxes3d(const float* pts, const float* boxes3d, int N, int M, int32_t* flags, prcnn_stream_t stream);

/* RPN training labels for a whole batch on the device (KittiRCNNDataset.generate_rpn_training_labels,
 * lib/datasets/kitti_rcnn_dataset.py:365-394): pts (B,N,3), gt_boxes3d (B,G,7) [x,y(bottom),z,h,w,l,ry], num_gt (B) i32 or
 * NULL (all G rows valid) -> cls_label (B,N) i32 in {-1 ignore, 0 background, 1 foreground}, reg_label (B,N,7)
 * [dx, dy, dz, h, w, l, ry] (zero rows for non-foreground points).  extra_width = 0.2 in the reference.  Boxes are applied in
 * order, later boxes overwrite earlier ones exactly as the reference's loop does.  G <= 128.  The in-box test is the analytic
 * one of roipool3d WITHOUT its 10 m centre-distance gate (the reference's hull test has none: a box with a half extent over
 * 10 m keeps all its points); it can differ from the hull test only for points within rounding distance of a face. */
int prcnn_rpn_labels(const float* pts, const float* gt_boxes3d, const int32_t* num_gt, int B, int N, int G, float extra_width,
                     int32_t* cls_label, float* reg_label, prcnn_stream_t stream);

/* GT-augmentation scene edit for a batch of scenes on the device -- the point work of
 * KittiRCNNDataset.apply_gt_aug_to_one_scene (lib/datasets/kitti_rcnn_dataset.py:484-489 one pts_in_boxes3d_cpu scan of the
 * scene per accepted object, :501-507 boolean-mask copy + concatenate).  pts (B,N,3), intensity (B,N) or NULL, boxes3d (B,K,7)
 * = the accepted objects' boxes [x,y(bottom),z,h,w,l,ry] (K <= 64), tested with h + extra_h (2.0 in the reference);
 * new_pts (B,P,3) / new_intensity (B,P) = the pasted objects' points, already concatenated in paste order.  num_pts / num_boxes /
 * num_new: (B) i32 live counts or NULL (all rows live).  Output: out_pts (B,N+P,3), out_intensity (B,N+P) or NULL: the scene
 * points outside every box IN THEIR ORIGINAL ORDER, then the new points; rows >= out_count[b] are zero.  out_count (B) i32;
 * removed (B,N) i32 or NULL: 1 where a scene point was dropped (the complement of the reference's src_pts_flag).
 * In-box test == prcnn_pts_in_boxes3d.  The sampling loop (database lookup, road plane, collision test) stays with the caller. */
int prcnn_gt_aug_edit(const float* pts, const float* intensity, const int32_t* num_pts, const float* boxes3d,
                      const int32_t* num_boxes, float extra_h, const float* new_pts, const float* new_intensity,
                      const int32_t* num_new, int B, int N, int K, int P, float* out_pts, float* out_intensity,
                      int32_t* out_count, int32_t* removed, prcnn_stream_t stream);

/* HOST twins of the reference's two CPU entry points (roipool3d.cpp:97-125 pts_in_boxes3d_cpu, :127-195 roipool3d_cpu),
 * which its dataloader calls inside forked worker processes (kitti_rcnn_dataset.py:487,582,625,843,970).  Host pointers,
 * no HIP call, bit-identical to the reference's CPU arithmetic (double half-extent compares, inclusive bounds).
 *   pts (N,3), boxes3d (M,7) -> flags (M,N) int64;
 *   pts (N,3), boxes3d (M,7), feat (N,C) -> pooled_pts (M,S,3), pooled_feat (M,S,C), empty (M) int64; rows of an empty
 *   box are left untouched (the caller zero-initialises them, roipool3d_utils.py:77-79). */
int prcnn_host_pts_in_boxes3d(const float* pts, const float* boxes3d, int64_t N, int64_t M, int64_t* flags);
int prcnn_host_roipool3d(const float* pts, const float* boxes3d, const float* feat, int64_t N, int64_t M, int64_t C, int64_t S,
                         float* pooled_pts, float* pooled_feat, int64_t* empty);

/* ---------------------------------------------------------------------------------------------
 * iou3d.  Replace iou3d_cuda.{boxes_overlap_bev_gpu,boxes_iou_bev_gpu,nms_gpu,nms_normal_gpu}
 * (iou3d.cpp:31,52,73,123).  BEV boxes are (N,5) [x1,y1,x2,y2,ry].
 * ------------------------------------------------------------------------------------------- */
int prcnn_boxes_overlap_bev(const float* boxes_a, int Na, const float* boxes_b, int Nb, float* out,
                            prcnn_stream_t stream);
int prcnn_boxes_iou_bev(const float* boxes_a, int Na, const float* boxes_b, int Nb, float* out, prcnn_stream_t stream);

#define PRCNN_NMS_ROTATED 0
#define PRCNN_NMS_NORMAL 1
/* workspace bytes prcnn_nms needs for N boxes (the suppression mask in whole blocks of 64 rows, 64 ceil(N/64) x ceil(N/64) words, + one flag byte per 64 x 64 tile of it) */
size_t prcnn_nms_workspace_bytes(int N);
/* Greedy NMS over boxes ALREADY sorted by descending score (iou3d_utils.py:64-66): box i suppresses
 * j>i iff iou(i,j) > thresh.  Entirely on the device: keep (N) i64 receives the kept positions in
 * ascending order, num_keep (1) i32 their count.  No host synchronisation.
 * max_keep: 0 = full sweep (reference semantics); > 0 = stop once that many boxes are kept -- the leading
 * max_keep entries are identical to the full result (what lib/rpn/proposal_layer.py:112 consumes). */
int prcnn_nms(const float* boxes, int N, float thresh, int kind, int max_keep, int64_t* keep, int32_t* num_keep,
              void* workspace, size_t workspace_bytes, prcnn_stream_t stream);

/* ======================================================================================================
 * Proposal stage: batched, sync-free replacements for the per-frame Python loop
 * between the RPN heads and roipool3d.  Oracle twins: prcnn_cpu_decode_bbox_target / _proposal_layer / _nms_batched.
 * ====================================================================================================== */

/* lib/utils/bbox_transform.py:24-121  decode_bbox_target(roi_box3d, pred_reg, loc_scope, loc_bin_size, num_head_bin,
 * anchor_size, get_xz_fine, get_y_by_bin, loc_y_scope, loc_y_bin_size, get_ry_fine) -- same argument meaning.
 *   roi (N, roi_cols) device: roi_cols 3 = xyz points (RPN stage, proposal_layer.py:23), 7 = RoI boxes (eval_rcnn.py:509,
 *   rotated back by roi ry).  pred_reg (N, C) device; C must equal the channel count the flags imply (the reference
 *   asserts, :106) else PRCNN_EINVAL.  anchor_size_host: 3 floats h,w,l in HOST memory (cfg.CLS_MEAN_SIZE[0]).
 *   Scalars are doubles because the reference computes them in Python floats before they meet a tensor.
 *   y_to_bottom != 0 also applies proposal_layer.py:32 (y += h/2).  out (N, 7) device [x,y,z,h,w,l,ry]. */
int prcnn_decode_bbox_target(const float* roi, int roi_cols, const float* pred_reg, long N, int C, double loc_scope,
                             double loc_bin_size, int num_head_bin, const float* anchor_size_host, int get_xz_fine,
                             int get_y_by_bin, double loc_y_scope, double loc_y_bin_size, int get_ry_fine, int y_to_bottom,
                             float* out, prcnn_stream_t stream);

/* workspace for prcnn_proposal_layer: pre_max = max(pre1, pre2), post_max = max(post1, post2) */
size_t prcnn_proposal_workspace_bytes(int B, int N, int pre_max, int post_max);
/* lib/rpn/proposal_layer.py:35-141 on decoded boxes3d (B, N, 7) and raw scores (B, N), whole batch, no host sync:
 * per frame order rows by descending score (NaN first, ties by ascending row -- torch.sort leaves tie order open),
 *   use_range != 0 (distance_based_proposal :58-117): area 1 = rows with r0 < z <= r1, area 2 = r1 < z <= r2; keep the
 *       first pre1 / pre2 of each; an empty area 2 borrows ranks [pre1, pre1+pre2) of area 1 (:91-98);
 *   use_range == 0 (score_based_proposal :119-141): one list, first pre1 rows (pre2 = post2 = 0);
 * greedy NMS (nms_kind, nms_thresh) on the BEV boxes (kitti_utils.py:134-147) of each list, first post1 / post2 survivors,
 * concatenated into out_boxes (B, post1+post2, 7) / out_scores (B, post1+post2), zero padded (:38-39).
 * out_count (B) i32, optional: rows filled per frame.  An empty area contributes nothing (the reference asserts, :90).
 * Frames of up to 16384 rows are sorted entirely in LDS; larger frames run the same bitonic network in 16384-key chunks with
 * the long strides through the workspace in HBM. */
int prcnn_proposal_layer(const float* scores, const float* boxes3d, int B, int N, int use_range, float r0, float r1, float r2,
                         int pre1, int pre2, int post1, int post2, float nms_thresh, int nms_kind, float* out_boxes,
                         float* out_scores, int32_t* out_count, void* workspace, size_t workspace_bytes, prcnn_stream_t stream);

size_t prcnn_nms_batched_workspace_bytes(int B, int M);
/* Final detection select (tools/eval_rcnn.py:600-614) for the whole batch: per frame, rows with valid[b,i] != 0
 * (valid NULL = all), ordered by descending score, greedy NMS on their BEV boxes; unlike iou3d_cuda.nms_gpu the boxes need
 * NOT be pre-sorted.  keep (B, max_keep) i32: row indices in kept order, -1 padded; num_keep (B) i32.
 * max_keep 0 = M.  PRCNN_EUNSUPPORTED when max_keep boxes do not fit the LDS kept list (~1900 rotated): use prcnn_nms. */
int prcnn_nms_batched(const float* boxes3d, const float* scores, const uint8_t* valid, int B, int M, float thresh, int kind,
                      int max_keep, int32_t* keep, int32_t* num_keep, void* workspace, size_t workspace_bytes,
                      prcnn_stream_t stream);

/* Fused RCNN input builder = lib/net/rcnn_net.py:127-154 in one pass:
 *   cat([extra0, extra1, feat]) -> roipool3d (first <= S in-box points per RoI, wrap-duplicated, boxes = pool_boxes3d,
 *   already enlarged by the caller) -> pooled xyz -= roi centre (:146-147) -> rotate_pc_along_y_torch(pooled xyz, roi ry)
 *   (kitti_utils.py:45-63), without the (B,N,2+C) and (B,M,S,5+C) tensors and without the per-frame Python loop.
 * rois (B,M,7) define the canonical frame; NULL keeps scene coordinates.  extra0 / extra1: per-point scalar channels
 * (B,N) (seg_mask, normalised depth) or NULL.  feat: (B,N,C) rows with stride ld_feat.
 * out_pts : B*M*S rows of stride ld_pts, columns [x',y',z',extra0,extra1];
 * out_feat: B*M*S rows of stride ld_out, C columns written at this pointer (point it INTO a wider buffer to place the
 *           features where the next layer wants them).  empty (B,M) i32.  An empty RoI yields zeros run through the
 *           same transform (i.e. -centre rotated), exactly what the reference's in-place ops produce.
 * distinct (B,M) i32, may be NULL: when given it receives the number of DISTINCT rows of every RoI -- min(points in the
 *           box, S), 1 for an empty RoI (all its rows are equal); rows distinct .. S-1 are wrap-copies of rows
 *           0 .. distinct-1 (roipool3d.cpp:171-177).  out_pts is still written in full (FPS and ball_query of the next
 *           level see the padded cloud), but the FEATURE rows of the copies are NOT written: hand `distinct` to the
 *           consumers as seg_cnt (prcnn_mlp_rows / prcnn_mlp_chain_rows) and valid_n (prcnn_group_compact). */
int prcnn_roipool3d_canonical(const float* xyz, const float* pool_boxes3d, const float* rois, const float* extra0,
                              const float* extra1, const float* feat, int ld_feat, int B, int N, int M, int C, int S,
                              float* out_pts, int ld_pts, float* out_feat, int ld_out, int32_t* empty, int32_t* distinct,
                              prcnn_stream_t stream);
/* ... with the scratch buffer of prcnn_roipool3d_ws (binned point selection) */
int prcnn_roipool3d_canonical_ws(const float* xyz, const float* pool_boxes3d, const float* rois, const float* extra0,
                                 const float* extra1, const float* feat, int ld_feat, int B, int N, int M, int C, int S,
                                 float* out_pts, int ld_pts, float* out_feat, int ld_out, int32_t* empty, int32_t* distinct,
                                 void* work, size_t work_bytes, prcnn_stream_t stream);

/* ======================================================================================================
 * Grid-accelerated neighbour search: the same results as prcnn_ball_query / prcnn_ball_query2 / prcnn_three_nn
 * (bit for bit), computed over a per-frame x-z grid (64 or 128 cells per axis) instead of a scan of all N points.
 * ====================================================================================================== */
/* bytes of the grid buffer for B frames of N points */
size_t prcnn_grid_bytes(int B, int N);
/* bin xyz (B,N,3) per frame: bounds -> cell size max(min_cell, extent/cells_per_axis) -> counting sort into cell-contiguous
 * float4 (x,y,z,index).  grid: 16-byte aligned device buffer of prcnn_grid_bytes(B,N).  min_cell: the largest search
 * radius that will be used (keeps a ball within 3x3 cells); 0 for three_nn.  cells_per_axis: 64 or 128 (finer cells keep
 * crowded near-sensor regions cheap for the ball query; ~1 point per cell suits the three_nn ring search). */
int prcnn_grid_build(const float* xyz, int B, int N, float min_cell, int cells_per_axis, void* grid, size_t grid_bytes,
                     prcnn_stream_t stream);
/* == prcnn_ball_query2 on the binned points (N = points per frame the grid was built with); nsample_b = 0: one radius.
 * xyz (B,N,3), the points the grid was built from, or NULL.  With xyz given, DENSE frames -- prcnn_grid_build estimates the
 * candidates a ball visits from the cell occupancy; above N/32 the grid kernel's per-candidate cost exceeds a scan's -- are
 * answered by the index-order scan of prcnn_ball_query2 instead (launched behind the grid kernel; each kernel exits at once on
 * the other's frames, so the launch sequence is static).  Results are identical either way. */
int prcnn_ball_query2_grid(const void* grid, const float* xyz, const float* new_xyz, int B, int N, int M, float radius_a,
                           int nsample_a, int32_t* idx_a, float radius_b, int nsample_b, int32_t* idx_b, prcnn_stream_t stream);
/* == prcnn_three_nn with `known` given as a grid of m points per frame */
int prcnn_three_nn_grid(const void* grid, const float* unknown, int B, int n, int m, float* dist2, int32_t* idx, float* weight,
                        prcnn_stream_t stream);

/* ======================================================================================================
 * KITTI object evaluation kernels (host side: pointrcnn_amd/kitti_eval.py).
 * Oracle twins: prcnn_cpu_rotate_iou_eval / prcnn_cpu_kitti_overlaps / prcnn_cpu_kitti_statistics.
 * ====================================================================================================== */
/* tools/kitti_object_eval_python/rotate_iou.py:287-329 rotate_iou_gpu_eval(boxes, query_boxes, criterion):
 * boxes (N,5), query (K,5) rows [cx, cy, dx, dy, angle] -> out (N,K).  criterion -1: IoU, 0: inter / area(query),
 * 1: inter / area(box), 2: intersection area (the numba kernel evaluates devRotateIoUEval(query_k, box_n), :282-284). */
int prcnn_rotate_iou_eval(const float* boxes, int N, const float* query, int K, int criterion, float* out, prcnn_stream_t stream);
/* Per-frame overlap blocks of eval.py:326-397 calculate_iou_partly for F frames in one launch: rows = first box set
 * ("dt"), columns = second ("gt").  metric 0: 2-D IoU of image boxes (.,4); 1: rotated BEV IoU, 2: 3-D IoU of camera
 * boxes (.,7) [x, y, z, l, h, w, ry] (float64 in, like the annotation arrays).  dt_off / gt_off (F+1) i32 row offsets,
 * ov_off (F+1) i64 offsets of the row-major blocks inside out (float64). */
int prcnn_kitti_overlaps(int metric, const double* dt, const int32_t* dt_off, const double* gt, const int32_t* gt_off,
                         const int64_t* ov_off, int F, double* out, prcnn_stream_t stream);
/* eval.py:155-324 compute_statistics_jit for every (frame, threshold): res (F, T, 4) = tp, fp, fn, similarity
 * (similarity -1 as in the reference when undefined).  gt_datas (G,5) [bbox, alpha], dt_datas (D,6) [bbox, alpha, score],
 * ign_gt (G) / ign_det (D) from clean_data, dc (DC,4) DontCare boxes, offsets (F+1).  compute_fp = 0 is the
 * reference's first pass (thresh ignored): matched (G) f64, optional, receives the score of the detection that made gt i
 * a true positive, NaN otherwise.  max_det_per_frame: caller-known bound (<= 1024, PRCNN_EUNSUPPORTED above). */
int prcnn_kitti_statistics(const double* overlaps, const int64_t* ov_off, const double* gt_datas, const int32_t* gt_off,
                           const double* dt_datas, const int32_t* dt_off, const int32_t* ign_gt, const int32_t* ign_det,
                           const double* dc, const int32_t* dc_off, int F, int max_det_per_frame, int metric, double min_overlap,
                           const double* thresholds, int T, int compute_fp, int compute_aos, double* res, double* matched,
                           prcnn_stream_t stream);

/* ======================================================================================================
 * Padding-free grouping (dedup.hip).  ball_query pads short groups by repeating their first hit; the rows of a group
 * beyond its cnt real hits are copies and cannot change the max-pooled result.  prcnn_group_compact splits the G = B*M
 * groups of one ball query, on the device, into
 *   sparse groups (cnt <= sparse_max): their real rows are appended to ONE flat row list -- ridx (G*sparse_max) global
 *       point index b*N + p, rnx (G*sparse_max, 3) the row's group centroid -- and the group is recorded as
 *       slist (G) group id, soff (G) first flat row, scnt (G) = cnt;
 *   dense groups (cnt > sparse_max): idxn (G, nsample) global indices of ALL nsample rows (padding included),
 *       nxn (G,3), listn (G) group id  (may be NULL when sparse_max == nsample: every group is then sparse).
 * counts (3 device i32): [0] flat rows, [1] dense groups, [2] sparse groups; list order is arbitrary.  All outputs are
 * sized for the worst case.  Run the MLP on each list as one frame (B = 1, N = B*N): the flat rows with M = G*sparse_max,
 * nsample = 1, no pooling, groups_dev = &counts[0], then prcnn_segmax_scatter; the dense list with M = G, pooling,
 * groups_dev = &counts[1], then prcnn_scatter_rows.
 * ====================================================================================================== */
int prcnn_group_compact(const int32_t* idx, const float* new_xyz, int B, int N, int M, int nsample, int sparse_max,
                        const int32_t* valid_n, int32_t* ridx, float* rnx, int32_t* slist, int32_t* soff, int32_t* scnt,
                        int32_t* idxn, float* nxn, int32_t* listn, int32_t* counts, prcnn_stream_t stream);
/* valid_n (B) i32, may be NULL: only points 0 .. valid_n[b]-1 of frame b are distinct, the rest of the frame are wrap-copies
 * of them in order (prcnn_roipool3d_canonical's `distinct`): ball_query lists ascending indices, so a hit >= valid_n[b] is a
 * copy of an earlier hit of the same group -- it ends the group's real rows like padding does, and in the dense list it is
 * replaced by the group's first hit, so that no consumer ever reads a copy's (unwritten) feature row. */
/* dst[list[j], col_off + c] = max over r < cnt[j] of src[off[j] + r, c]  for j < *count, c < C */
int prcnn_segmax_scatter(const float* src, int ld_src, const int32_t* list, const int32_t* off, const int32_t* cnt,
                         const int32_t* count, int max_groups, int C, float* dst, int ld_dst, int col_off, prcnn_stream_t stream);
/* dst[list[r], col_off : col_off + C] = src[r, 0:C] for r < *count (count: device i32; max_rows sizes the launch) */
int prcnn_scatter_rows(const float* src, int ld_src, const int32_t* list, const int32_t* count, int max_rows, int C, float* dst,
                       int ld_dst, int col_off, prcnn_stream_t stream);

/* ======================================================================================================
 * RPN input builder (scene.hip) -- SURVEY 8(f) rank 4.  Replaces the inference branch of
 * lib/datasets/kitti_rcnn_dataset.py:246-310 (get_rpn_sample): calibration.py:51-70 lidar_to_rect + rect_to_img,
 * get_valid_flag (kitti_rcnn_dataset.py:198-219), mask compaction and the npoints sampling of :285-306, for a whole batch.
 *   raw      (total_points, 4) f32 device: the frames' velodyne scans [x y z intensity] back to back (kitti_dataset.py:40-43)
 *   offsets  (B+1) i64 device: first raw point of every frame; max_points_per_frame: host-known bound on the frame sizes
 *   calib    (B, 24) f32 device: M = V2C^T . R0^T (4x3 row-major, calibration.py:57) then P2 (3x4 row-major)
 *   img_hw   (B, 2) i32 device: image height, width (kitti_dataset.py:34-39)
 *   scope    6 doubles on the HOST [x0 x1 y0 y1 z0 z1] = cfg.PC_AREA_SCOPE, or NULL (cfg.PC_REDUCE_BY_RANGE false)
 *   out_xyz (B, npoints, 3), out_intensity (B, npoints) = intensity - 0.5, out_src (B, npoints) raw index of every row,
 *   nvalid (B) number of valid points, status (B): 0 ok; 1 = the reference raises for this frame (more than npoints
 *   far points, or fewer than npoints/2 valid ones) -- rows are still produced; 2 = no valid point (rows zero, src -1).
 * Every point at rect depth >= 40 m is kept, the rest drawn without replacement, the result shuffled; short frames are
 * topped up with a draw without replacement from themselves.  The draw uses a counter-based generator keyed by
 * (seed, frame, raw index) -- see scene.hip -- so results are reproducible and independent of launch geometry
 * (the reference uses numpy's unseeded global stream).  npoints <= 16384.
 * ====================================================================================================== */
size_t prcnn_scene_workspace_bytes(int64_t total_points, int B);
int prcnn_scene_prepare(const float* raw, const int64_t* offsets, int B, int64_t total_points, int max_points_per_frame,
                        const float* calib, const int32_t* img_hw, const double* scope, int npoints, uint32_t seed,
                        float* out_xyz, float* out_intensity, int32_t* out_src, int32_t* nvalid, int32_t* status,
                        void* workspace, size_t workspace_bytes, prcnn_stream_t stream);

/* ======================================================================================================
 * Training-mode SharedMLP (csrc/mlp_train.h) -- BASELINE config 4, `train_rcnn.py --train_mode rpn`.
 * Replaces, for one SharedMLP layer, nn.Conv2d(1x1, bias=False) -> nn.BatchNorm2d (batch statistics, running-stat update) ->
 * ReLU and its autograd (cuDNN / ATen in the reference: lib/net/pointnet2_msg.py:20-45 via the upstream pytorch_utils.SharedMLP,
 * tools/train_rcnn.py:198-199), with F.max_pool2d(kernel=[1, nsample]) after the last layer.  All tensors are channels-last rows.
 * The layer's ONLY saved activation is its pre-normalisation output y; consumers apply relu(y * scale + shift) on load.
 *
 * prcnn_train_src_t says how the rows entering a layer's convolution are produced (mode = 0 plain rows `in`, optionally with
 * the previous layer's normalisation `pro_scale / pro_shift` (K entries zero-padded to a multiple of 32 floats) applied;
 * 1 = grouped rows [feat[idx] | xyz[idx] - new_xyz] (the weight is packed with k_rot = 3), rows = B * M * ns, K = C + 3;
 * 2 = interpolated rows [sum_j w3 known[idx3] | skip], rows = B * n, K = C2 + C1).
 * A layer's constant table cst holds 6 rows of ld_c floats: scale = gamma * invstd, shift = beta - mean * scale, mean, invstd
 * (forward), mean(dyhat), mean(dyhat * xhat) (backward).  One call runs a whole stack; the host side is C++ (no per-layer Python).
 * ====================================================================================================== */
typedef struct prcnn_train_src {
    int mode;
    int64_t rows;
    int K;
    const float* in; int ld_in; const float* pro_scale; const float* pro_shift;
    const float* xyz; const float* new_xyz; const int32_t* idx; const float* feat; int ld_feat; int B, N, M, ns, C;
    const float* known; const int32_t* idx3; const float* w3; const float* skip; int ld_known, ld_skip, n, m, C2, C1;
    /* padding-free rows (grouped source only; all NULL / 0 otherwise): the rows are the DISTINCT rows of the groups as built by
     * prcnn_train_group_rows -- B = 1, N = frames * points, M = rows = the worst case frames * npoint * nsample, ns = 1, idx = ridx,
     * new_xyz = rnx; mult (rows) row multiplicities, rows_dev the live row count on the device, norm_rows = frames * npoint *
     * nsample (what the batch statistics are normalised by), seg_off / seg_cnt (groups) first row / row count of every group,
     * row_grp (rows) the group of every row.  The pooled output has `groups` rows; arg holds the arg-max position inside the group. */
    const float* mult; const int32_t* rows_dev; int64_t norm_rows; const int32_t* seg_off; const int32_t* seg_cnt; const int32_t* row_grp;
    int groups;
} prcnn_train_src_t;

/* One layer of a stack.  The caller owns every buffer (nothing is allocated inside):
 *   W (Nout, K) conv weight in torch layout (K = the source's K for layer 0, the previous layer's Nout after that); gamma, beta,
 *   running_mean, running_var (Nout) and eps, momentum of its BatchNorm;
 *   y (rows, Nout)  the layer's saved pre-normalisation output (written by forward, read by backward);
 *   cst (6, ld_c)   constant table, ZERO-INITIALISED by the caller before forward; ld_c a multiple of 128, >= Nout;
 *   wpack           prcnn_wpack_floats(Nout, K) floats: the forward weight image (written by forward);
 *   wpack_t         prcnn_wpack_floats(Kin, Nout) floats: the dgrad weight image, Kin = K (K - 3 for a grouped layer 0, whose
 *                   xyz columns carry no gradient); NULL when the layer's input takes no gradient (layer 0 only);
 *   dW (Nout, K) in torch's channel order, dgamma, dbeta (Nout): written by backward.
 * gamma == NULL: a layer WITHOUT normalisation -- Conv (bias in `beta`, or NULL) -> ReLU, the form of the RCNN stage
 * (cfg.RCNN.USE_BN = False, lib/net/rcnn_net.py:24-60): no statistics, dbeta receives the bias gradient, dgamma / running_* unused. */
typedef struct prcnn_train_layer {
    int Nout;
    const float* W; const float* gamma; const float* beta;
    float eps, momentum;
    float* running_mean; float* running_var;
    float* y;
    float* cst; int ld_c;
    float* wpack; float* wpack_t;
    float* dW; float* dgamma; float* dbeta;
} prcnn_train_layer_t;

/* scratch both passes need (partials of the statistics, wgrad partial tiles, the inter-layer gradient ping-pong); 256-byte aligned */
size_t prcnn_train_stack_work_bytes(int64_t rows, const prcnn_train_layer_t* layers, int nl, int K0, int backward);
/* forward of the whole stack: per layer pack weights -> y = A . W^T on the MFMA pipe with per-64-row (mean, M2) partials -> batch
 * statistics in double, fixed order -> cst rows 0..3 and the running statistics ((1 - momentum) * old + momentum * new, unbiased
 * variance); then out[g, col_off + n] = max over the pool_ns rows of group g of relu(y * scale + shift), arg (groups, Nout) u8 = the
 * FIRST maximal row (torch.max / max_pool2d route the gradient there); pool_ns <= 1: plain normalise + ReLU, arg unused.
 * a_dump (rows, ld_dump): copy of the assembled first-layer rows of a gathered source (backward's wgrad reads it); NULL for plain. */
int prcnn_train_stack_fwd(const prcnn_train_src_t* src, const prcnn_train_layer_t* layers, int nl, int pool_ns, float* a_dump, int ld_dump,
                          float* out, int ld_out, int col_off, uint8_t* arg, void* work, size_t work_bytes, prcnn_stream_t stream);
/* backward of the whole stack from gout = d loss / d out ((rows / pool_ns, N_last) rows): per layer BatchNorm reductions (dgamma,
 * dbeta), wgrad (dW), dgrad into the previous layer.  gin (rows, ld_gin), may be NULL: gradient w.r.t. the rows that entered layer 0
 * (grouped source: the C feature columns only) -- scatter it with prcnn_group_rows_grad / prcnn_interp_rows_grad. */
int prcnn_train_stack_bwd(const prcnn_train_src_t* src, const prcnn_train_layer_t* layers, int nl, int pool_ns, const float* a_dump,
                          int ld_dump, const float* gout, int ld_gout, const uint8_t* arg, float* gin, int ld_gin, void* work,
                          size_t work_bytes, prcnn_stream_t stream);
/* Padding-free rows for a grouped stack (exact: ball_query pads a group with copies of its first hit; copies are identical rows).
 * prcnn_train_group_rows builds the distinct-row list deterministically (count -> exclusive scan -> fill, group order);
 * prcnn_flat_rows_grad scatters the first layer's row gradients back: dfeat[ridx[r], 0:C] += G[r, 0:C] for the live rows. */
int prcnn_train_group_rows(const int32_t* idx, const float* new_xyz, int B, int N, int M, int ns, int32_t* cnt, int32_t* off,
                           int32_t* rows_dev, int32_t* ridx, float* rnx, float* mult, int32_t* row_grp, prcnn_stream_t stream);
int prcnn_flat_rows_grad(const float* G, int ldG, const int32_t* ridx, const int32_t* rows_dev, int64_t max_rows, int C, float* dfeat,
                         int ld_d, prcnn_stream_t stream);
/* ... as a gather (no atomics, summed in ascending row order): frame b's rows are those of its M groups (seg_off from
 * prcnn_train_group_rows); dfeat is WRITTEN.  Scratch: >= prcnn_flat_rows_grad_work_bytes(B, N, max_rows) device bytes (0: no gather
 * form for this shape); work == NULL / too small: dfeat is cleared and prcnn_flat_rows_grad runs. */
size_t prcnn_flat_rows_grad_work_bytes(int B, int N, int64_t max_rows);
int prcnn_flat_rows_grad_ws(const float* G, int ldG, const int32_t* ridx, const int32_t* rows_dev, int64_t max_rows, int C, float* dfeat,
                            int ld_d, const int32_t* seg_off, int B, int N, int M, void* work, size_t work_bytes, prcnn_stream_t stream);
/* backward of the gathers on channels-last rows: dfeat (B, N, ld_d) += scatter of G ((B, M, ns) rows, C channels) through idx;
 * dknown (B, m, ld_d) += w3-weighted scatter of G ((B, n) rows) through idx3.  Outputs pre-zeroed by the caller. */
int prcnn_group_rows_grad(const float* G, int ldG, const int32_t* idx, int B, int M, int ns, int C, int N, float* dfeat, int ld_d,
                          prcnn_stream_t stream);
int prcnn_interp_rows_grad(const float* G, int ldG, const int32_t* idx3, const float* w3, int B, int n, int m, int C, float* dknown,
                           int ld_d, prcnn_stream_t stream);
/* The same as a gather (no atomics, summed in ascending row order: the result does not depend on scheduling) with a scratch buffer of
 * >= prcnn_interp_rows_grad_work_bytes(B, n, m) device bytes (0: this shape has no gather form).  dknown is WRITTEN, not accumulated
 * into, and need not be zeroed.  work == NULL / too small / rows not 16-byte aligned: dknown is cleared and prcnn_interp_rows_grad runs. */
size_t prcnn_interp_rows_grad_work_bytes(int B, int n, int m);
int prcnn_interp_rows_grad_ws(const float* G, int ldG, const int32_t* idx3, const float* w3, int B, int n, int m, int C, float* dknown,
                              int ld_d, void* work, size_t work_bytes, prcnn_stream_t stream);

/* ======================================================================================================
 * RCNN-stage training targets (csrc/proposal_target.hip) -- `train_rcnn.py --train_mode rcnn`.
 * prcnn_boxes_iou3d: iou3d_utils.boxes_iou3d_gpu (lib/utils/iou3d/iou3d_utils.py:20-53) as one kernel: a (Na,7), b (Nb,7)
 * [x, y(bottom), z, h, w, l, ry] -> out (Na, Nb).
 * prcnn_proposal_target_sample: ProposalTargetLayer.sample_rois_for_rcnn with sample_bg_inds, aug_roi_by_noise_torch and
 * random_aug_box3d (lib/rpn/proposal_target_layer.py:75-300) for the whole batch in one launch, no host round trip:
 *   roi_boxes3d (B, M, 7), gt_boxes3d (B, G, gt_cols >= 7) with all-zero rows padding the end (G <= 128);
 *   cfg6 (HOST pointer, six DOUBLES -- the reference's sampler does its slot arithmetic in Python doubles): RCNN.REG_FG_THRESH, CLS_FG_THRESH, CLS_BG_THRESH, CLS_BG_THRESH_LO, FG_RATIO, HARD_BG_RATIO;
 *   aug_times = RCNN.ROI_FG_AUG_TIMES; aug_method 0 = 'multiple', 1 = 'single' (RCNN.REG_AUG_METHOD);
 *   -> rois, gt_of_rois (B, R, 7), roi_iou (B, R), src (B, R) the input RoI behind every slot, max_overlaps / gt_assignment
 *   (B, M), counts (B, 4) = fg / hard-bg / easy-bg candidates and fg slots, status (B): 0 ok, 1 = neither foreground nor
 *   background candidates (the reference raises NotImplementedError), 2 = no ground-truth box.
 * The random draw is a counter-based table keyed by (seed, purpose, frame, position) -- see the kernel file; the reference's own
 * methods, answered from the same table, return the same boxes (tests/golden/ref_proposal_target.py).
 * ====================================================================================================== */
int prcnn_boxes_iou3d(const float* a, int Na, const float* b, int Nb, float* out, prcnn_stream_t stream);
/* The box trigonometry of roipool3d / iou3d / NMS / labels / RoI sampling is the reference's HOST libm arithmetic, restated bit for
 * bit (csrc/ref_trig.h: glibc 2.35 sinf / cosf / atan2f).  Element-wise evaluation for inspection / tests: fn 0 sinf(a), 1 cosf(a),
 * 2 atan2f(a, b). */
int prcnn_ref_trig(const float* a, const float* b, int n, int fn, float* out, prcnn_stream_t stream);
int prcnn_proposal_target_sample(const float* roi_boxes3d, const float* gt_boxes3d, int B, int M, int G, int gt_cols, int roi_per_image,
                                 const double* cfg6, int aug_times, int aug_method, uint32_t seed, float* rois, float* gt_of_rois,
                                 float* roi_iou, int32_t* src, float* max_overlaps, int32_t* gt_assignment, int32_t* counts,
                                 int32_t* status, prcnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PRCNN_POINTOPS_H */
