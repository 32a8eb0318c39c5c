"""CPU oracle for the PointRCNN point-ops hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package.  The product (``pointrcnn_amd``) never does: it fails loudly when its HIP library is missing.

``oracle.cpu``  -- numpy front-end of ``libprcnn_oracle.so`` (``prcnn_oracle.c``: plain-C restatement of
                   every operator on the path, each citing the reference file:line it follows).
``oracle.ref``  -- numpy front-end of ``_ref/libprcnn_ref.so`` (the reference's OWN iou3d / roipool3d
                   sources compiled for the host, see ``build_ref.py``); ``None`` when not built.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int32)
_L = ctypes.POINTER(ctypes.c_int64)
_U = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    """Compile the C restatement (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "libprcnn_oracle.so")
    src = os.path.join(_HERE, "prcnn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "libprcnn_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libprcnn_ref.so")
    from . import build_ref
    if build_ref.have_reference() and (force or not os.path.exists(ref_so)):
        build_ref.build(verbose=False)
    return so


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


# the arithmetic the HIP kernels of roipool3d / iou3d / NMS / labels / RoI sampling implement: 2 = glibc's float sinf / cosf /
# atan2f restated bit for bit (csrc/ref_trig.h) == the reference's host arithmetic (mode 0); 1 = the round-1/2 contract (double
# rounded once, division-only vertex order), still what decode_bbox_target / the canonical transform use
KERNEL_TRIG = 2


class _Cpu:
    """numpy wrappers around prcnn_cpu_* (shapes follow the reference op surface)."""

    def __init__(self):
        build()
        self.lib = ctypes.CDLL(os.path.join(_HERE, "libprcnn_oracle.so"))
        self.lib.prcnn_cpu_nms.restype = ctypes.c_int
        self.lib.prcnn_cpu_decode_bbox_target.restype = ctypes.c_int

    # ---- PointNet++ ops ----
    def fps(self, xyz, npoint):
        xyz = _f32(xyz)
        B, N, _ = xyz.shape
        idx = np.zeros((B, npoint), np.int32)
        self.lib.prcnn_cpu_fps(_p(xyz, _F), B, N, npoint, None, _p(idx, _I))
        return idx

    def fps_upstream(self, xyz, npoint):
        """FPS with the upstream CUDA kernel's tie order: argmin (k mod T, k) among equal maxima"""
        xyz = _f32(xyz)
        B, N, _ = xyz.shape
        idx = np.zeros((B, npoint), np.int32)
        self.lib.prcnn_cpu_fps_upstream(_p(xyz, _F), B, N, npoint, _p(idx, _I))
        return idx

    # ---- comparison mode: tie order x distance arithmetic ("upstream" = nvcc's contraction of a*a + b*b + c*c, prcnn_oracle.c) ----
    def fps_mode(self, xyz, npoint, order=0, arith=0):
        xyz = _f32(xyz)
        B, N, _ = xyz.shape
        idx = np.zeros((B, npoint), np.int32)
        self.lib.prcnn_cpu_fps_mode(_p(xyz, _F), B, N, npoint, int(order), int(arith), _p(idx, _I))
        return idx

    def ball_query_arith(self, radius, nsample, xyz, new_xyz, arith):
        xyz, new_xyz = _f32(xyz), _f32(new_xyz)
        B, N, _ = xyz.shape
        M = new_xyz.shape[1]
        idx = np.zeros((B, M, nsample), np.int32)
        self.lib.prcnn_cpu_ball_query_arith(_p(xyz, _F), _p(new_xyz, _F), B, N, M, ctypes.c_float(radius), nsample, int(arith), _p(idx, _I))
        return idx

    def three_nn_arith(self, unknown, known, arith):
        unknown, known = _f32(unknown), _f32(known)
        B, n, _ = unknown.shape
        m = known.shape[1]
        d2 = np.zeros((B, n, 3), np.float32)
        idx = np.zeros((B, n, 3), np.int32)
        self.lib.prcnn_cpu_three_nn_arith(_p(unknown, _F), _p(known, _F), B, n, m, int(arith), _p(d2, _F), _p(idx, _I))
        return d2, idx

    def gt_aug_edit(self, pts, intensity, boxes3d, new_pts, new_intensity, num_pts=None, num_boxes=None, num_new=None,
                    extra_h=2.0, trig_mode=KERNEL_TRIG):
        """-> out_pts (B,N+P,3), out_intensity (B,N+P), count (B) i32, removed (B,N) i32"""
        pts, boxes3d, new_pts = _f32(pts), _f32(boxes3d), _f32(new_pts)
        intensity, new_intensity = _f32(intensity), _f32(new_intensity)
        B, N, _ = pts.shape
        K, P = boxes3d.shape[1], new_pts.shape[1]
        cnt = [None if c is None else np.ascontiguousarray(c, np.int32) for c in (num_pts, num_boxes, num_new)]
        out_pts = np.zeros((B, N + P, 3), np.float32)
        out_int = np.zeros((B, N + P), np.float32)
        count = np.zeros((B,), np.int32)
        removed = np.zeros((B, N), np.int32)
        self.lib.prcnn_cpu_gt_aug_edit(_p(pts, _F), _p(intensity, _F), None if cnt[0] is None else _p(cnt[0], _I), _p(boxes3d, _F),
                                       None if cnt[1] is None else _p(cnt[1], _I), ctypes.c_float(extra_h), _p(new_pts, _F),
                                       _p(new_intensity, _F), None if cnt[2] is None else _p(cnt[2], _I), B, N, K, P, trig_mode,
                                       _p(out_pts, _F), _p(out_int, _F), _p(count, _I), _p(removed, _I))
        return out_pts, out_int, count, removed

    def rpn_labels(self, pts, gt_boxes3d, num_gt=None, extra_width=0.2, trig_mode=KERNEL_TRIG):
        pts, gt = _f32(pts), _f32(gt_boxes3d)
        B, N, _ = pts.shape
        G = gt.shape[1]
        cls, reg = np.zeros((B, N), np.int32), np.zeros((B, N, 7), np.float32)
        ng = None if num_gt is None else _i32(num_gt)
        self.lib.prcnn_cpu_rpn_labels(_p(pts, _F), _p(gt, _F), None if ng is None else _p(ng, _I), B, N, G,
                                      ctypes.c_float(extra_width), trig_mode, _p(cls, _I), _p(reg, _F))
        return cls, reg

    def gather(self, feat, idx):
        feat, idx = _f32(feat), _i32(idx)
        B, C, N = feat.shape
        M = idx.shape[1]
        out = np.zeros((B, C, M), np.float32)
        self.lib.prcnn_cpu_gather(_p(feat, _F), _p(idx, _I), B, C, N, M, _p(out, _F))
        return out

    def gather_grad(self, grad_out, idx, N):
        grad_out, idx = _f32(grad_out), _i32(idx)
        B, C, M = grad_out.shape
        out = np.zeros((B, C, N), np.float32)
        self.lib.prcnn_cpu_gather_grad(_p(grad_out, _F), _p(idx, _I), B, C, N, M, _p(out, _F))
        return out

    def ball_query(self, radius, nsample, xyz, new_xyz):
        xyz, new_xyz = _f32(xyz), _f32(new_xyz)
        B, N, _ = xyz.shape
        M = new_xyz.shape[1]
        idx = np.zeros((B, M, nsample), np.int32)
        self.lib.prcnn_cpu_ball_query(_p(xyz, _F), _p(new_xyz, _F), B, N, M, ctypes.c_float(radius), nsample,
                                      _p(idx, _I))
        return idx

    def group(self, feat, idx):
        feat, idx = _f32(feat), _i32(idx)
        B, C, N = feat.shape
        _, M, ns = idx.shape
        out = np.zeros((B, C, M, ns), np.float32)
        self.lib.prcnn_cpu_group(_p(feat, _F), _p(idx, _I), B, C, N, M, ns, _p(out, _F))
        return out

    def group_grad(self, grad_out, idx, N):
        grad_out, idx = _f32(grad_out), _i32(idx)
        B, C, M, ns = grad_out.shape
        out = np.zeros((B, C, N), np.float32)
        self.lib.prcnn_cpu_group_grad(_p(grad_out, _F), _p(idx, _I), B, C, N, M, ns, _p(out, _F))
        return out

    def three_nn(self, unknown, known):
        """returns (dist2 (B,n,3), idx (B,n,3)); the op surface returns sqrt(dist2)."""
        unknown, known = _f32(unknown), _f32(known)
        B, n, _ = unknown.shape
        m = known.shape[1]
        d2 = np.zeros((B, n, 3), np.float32)
        idx = np.zeros((B, n, 3), np.int32)
        self.lib.prcnn_cpu_three_nn(_p(unknown, _F), _p(known, _F), B, n, m, _p(d2, _F), _p(idx, _I))
        return d2, idx

    def three_weights(self, dist2):
        dist2 = _f32(dist2)
        w = np.zeros_like(dist2)
        self.lib.prcnn_cpu_three_weights(_p(dist2, _F), dist2.size // 3, _p(w, _F))
        return w

    def three_interp(self, feat, idx, w):
        feat, idx, w = _f32(feat), _i32(idx), _f32(w)
        B, C, m = feat.shape
        n = idx.shape[1]
        out = np.zeros((B, C, n), np.float32)
        self.lib.prcnn_cpu_three_interp(_p(feat, _F), _p(idx, _I), _p(w, _F), B, C, m, n, _p(out, _F))
        return out

    def three_interp_grad(self, grad_out, idx, w, m):
        grad_out, idx, w = _f32(grad_out), _i32(idx), _f32(w)
        B, C, n = grad_out.shape
        out = np.zeros((B, C, m), np.float32)
        self.lib.prcnn_cpu_three_interp_grad(_p(grad_out, _F), _p(idx, _I), _p(w, _F), B, C, n, m, _p(out, _F))
        return out

    def linear_rows(self, a, w, bias=None, relu=False):
        """a (R,K), w (Nout,K) [torch conv weight layout], bias (Nout) -> (R,Nout); double accumulation."""
        a, w = _f32(a), _f32(w)
        R, K = a.shape
        Nout = w.shape[0]
        out = np.zeros((R, Nout), np.float32)
        b = _f32(bias) if bias is not None else None
        self.lib.prcnn_cpu_linear_rows(_p(a, _F), _p(w, _F), _p(b, _F) if b is not None else None, R, K, Nout,
                                       int(relu), _p(out, _F))
        return out

    # ---- roipool3d ----
    def pts_in_boxes3d(self, pts, boxes3d, trig_mode=KERNEL_TRIG):
        pts, boxes3d = _f32(pts), _f32(boxes3d)
        N, M = pts.shape[0], boxes3d.shape[0]
        flags = np.zeros((M, N), np.int64)
        self.lib.prcnn_cpu_pts_in_boxes3d(_p(pts, _F), _p(boxes3d, _F), N, M, trig_mode, _p(flags, _L))
        return flags

    def roipool3d(self, xyz, boxes3d, feat, S, trig_mode=KERNEL_TRIG):
        """boxes already enlarged.  -> pooled (B,M,S,3+C) f32, empty (B,M) i32"""
        xyz, boxes3d, feat = _f32(xyz), _f32(boxes3d), _f32(feat)
        B, N, _ = xyz.shape
        M, C = boxes3d.shape[1], feat.shape[2]
        out = np.zeros((B, M, S, 3 + C), np.float32)
        empty = np.zeros((B, M), np.int32)
        self.lib.prcnn_cpu_roipool3d(_p(xyz, _F), _p(boxes3d, _F), _p(feat, _F), B, N, M, C, S, trig_mode,
                                     _p(out, _F), _p(empty, _I))
        return out, empty

    def canonical_transform(self, pooled, rois, trig_mode=1):
        """rcnn_net.py:143-150 on a pooled (B,M,S,W) tensor (returns a transformed copy)"""
        out = _f32(pooled).copy()
        rois = _f32(rois)
        B, M, S, W = out.shape
        self.lib.prcnn_cpu_canonical_transform(_p(out, _F), _p(rois, _F), B, M, S, W, trig_mode)
        return out

    # ---- iou3d ----
    def boxes_overlap_bev(self, a, b, trig_mode=KERNEL_TRIG):
        a, b = _f32(a), _f32(b)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        self.lib.prcnn_cpu_boxes_overlap_bev(_p(a, _F), a.shape[0], _p(b, _F), b.shape[0], trig_mode, _p(out, _F))
        return out

    def boxes_iou_bev(self, a, b, trig_mode=KERNEL_TRIG):
        a, b = _f32(a), _f32(b)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        self.lib.prcnn_cpu_boxes_iou_bev(_p(a, _F), a.shape[0], _p(b, _F), b.shape[0], trig_mode, _p(out, _F))
        return out

    def nms(self, boxes_sorted, thresh, kind="rotated", trig_mode=KERNEL_TRIG):
        """boxes already sorted by descending score -> kept positions (int64)"""
        boxes = _f32(boxes_sorted)
        keep = np.zeros((boxes.shape[0],), np.int64)
        n = self.lib.prcnn_cpu_nms(_p(boxes, _F), boxes.shape[0], ctypes.c_float(thresh),
                                   0 if kind == "rotated" else 1, trig_mode, _p(keep, _L))
        return keep[:n].copy()

    def nms_mask(self, boxes_sorted, thresh, kind="rotated", trig_mode=KERNEL_TRIG):
        boxes = _f32(boxes_sorted)
        N = boxes.shape[0]
        mask = np.zeros((N, (N + 63) // 64), np.uint64)
        self.lib.prcnn_cpu_nms_mask(_p(boxes, _F), N, ctypes.c_float(thresh), 0 if kind == "rotated" else 1,
                                    trig_mode, _p(mask, _U))
        return mask

    # ---- proposal stage ----
    def decode_bbox_target(self, roi, reg, loc_scope, loc_bin_size, num_head_bin, anchor_size, get_xz_fine=True,
                           get_y_by_bin=False, loc_y_scope=0.5, loc_y_bin_size=0.25, get_ry_fine=False,
                           y_to_bottom=False, trig_mode=1):
        roi, reg, anchor = _f32(roi), _f32(reg), _f32(anchor_size)
        out = np.zeros((reg.shape[0], 7), np.float32)
        rc = self.lib.prcnn_cpu_decode_bbox_target(
            _p(roi, _F), roi.shape[1], _p(reg, _F), reg.shape[0], reg.shape[1], ctypes.c_double(loc_scope),
            ctypes.c_double(loc_bin_size), int(num_head_bin), _p(anchor, _F), int(get_xz_fine), int(get_y_by_bin),
            ctypes.c_double(loc_y_scope), ctypes.c_double(loc_y_bin_size), int(get_ry_fine), int(y_to_bottom), trig_mode,
            _p(out, _F))
        if rc:
            raise ValueError("decode_bbox_target: channel layout does not match C=%d" % reg.shape[1])
        return out

    def argsort_desc(self, scores):
        scores = _f32(scores)
        order = np.zeros(scores.shape, np.int32)
        self.lib.prcnn_cpu_argsort_desc(_p(scores, _F), scores.shape[0], _p(order, _I))
        return order

    def proposal_layer(self, scores, boxes3d, pre, post, thresh, kind="normal", ranges=(0.0, 40.0, 80.0), trig_mode=KERNEL_TRIG):
        """pre / post = (n_area1, n_area2); ranges=None -> score_based_proposal"""
        scores, boxes3d = _f32(scores), _f32(boxes3d)
        B, N = scores.shape
        tot = post[0] + post[1]
        ob, osc, cnt = np.zeros((B, tot, 7), np.float32), np.zeros((B, tot), np.float32), np.zeros((B,), np.int32)
        r = ranges if ranges is not None else (0.0, 0.0, 0.0)
        self.lib.prcnn_cpu_proposal_layer(_p(scores, _F), _p(boxes3d, _F), B, N, int(ranges is not None),
                                          ctypes.c_float(r[0]), ctypes.c_float(r[1]), ctypes.c_float(r[2]), pre[0], pre[1],
                                          post[0], post[1], ctypes.c_float(thresh), 0 if kind == "rotated" else 1, trig_mode,
                                          _p(ob, _F), _p(osc, _F), _p(cnt, _I))
        return ob, osc, cnt

    def nms_batched(self, boxes3d, scores, valid, thresh, kind="rotated", max_keep=0, trig_mode=KERNEL_TRIG):
        boxes3d, scores = _f32(boxes3d), _f32(scores)
        B, M = scores.shape
        mk = M if max_keep <= 0 or max_keep > M else max_keep
        keep, num = np.zeros((B, mk), np.int32), np.zeros((B,), np.int32)
        v = None if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
        self.lib.prcnn_cpu_nms_batched(_p(boxes3d, _F), _p(scores, _F),
                                       None if v is None else v.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), B, M,
                                       ctypes.c_float(thresh), 0 if kind == "rotated" else 1, trig_mode, mk,
                                       _p(keep, _I), _p(num, _I))
        return keep, num


    def ref_trig(self, fn, a, b=None):
        """fn in {"sinf", "cosf", "atan2f"} (csrc/ref_trig.h) or {"libm_sinf", "libm_cosf", "libm_atan2f"} (the host's libm)"""
        a = _f32(a).reshape(-1)
        b = a if b is None else _f32(b).reshape(-1)
        out = np.zeros(a.shape, np.float32)
        self.lib.prcnn_cpu_ref_trig(_p(a, _F), _p(b, _F), a.shape[0], ["sinf", "cosf", "atan2f", "libm_sinf", "libm_cosf", "libm_atan2f"].index(fn),
                                    _p(out, _F))
        return out

    def boxes_iou3d(self, a, b, trig_mode=KERNEL_TRIG):
        """iou3d_utils.boxes_iou3d_gpu: a (N,7), b (M,7) -> (N,M)"""
        a, b = _f32(a), _f32(b)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        self.lib.prcnn_cpu_boxes_iou3d(_p(a, _F), a.shape[0], _p(b, _F), b.shape[0], trig_mode, _p(out, _F))
        return out

    def proposal_target_sample(self, roi_boxes3d, gt_boxes3d, roi_per_image=64, cfgv=(0.55, 0.6, 0.45, 0.05, 0.5, 0.8), aug_times=10,
                               aug_method="multiple", seed=0, trig_mode=KERNEL_TRIG):
        """ProposalTargetLayer.sample_rois_for_rcnn with the counter-based draw -> dict of arrays"""
        roi, gt = _f32(roi_boxes3d), _f32(gt_boxes3d)
        B, M, _ = roi.shape
        G, gc = gt.shape[1], gt.shape[2]
        R = roi_per_image
        o = {"rois": np.zeros((B, R, 7), np.float32), "gt_of_rois": np.zeros((B, R, 7), np.float32), "roi_iou": np.zeros((B, R), np.float32),
             "src": np.zeros((B, R), np.int32), "max_overlaps": np.zeros((B, M), np.float32), "gt_assignment": np.zeros((B, M), np.int32),
             "counts": np.zeros((B, 4), np.int32), "status": np.zeros((B,), np.int32)}
        c = np.ascontiguousarray(cfgv, np.float64)
        self.lib.prcnn_cpu_proposal_target_sample(_p(roi, _F), _p(gt, _F), B, M, G, gc, R, _p(c, ctypes.POINTER(ctypes.c_double)), int(aug_times),
                                                  {"multiple": 0, "single": 1}[aug_method], ctypes.c_uint32(seed), trig_mode,
                                                  _p(o["rois"], _F), _p(o["gt_of_rois"], _F), _p(o["roi_iou"], _F), _p(o["src"], _I),
                                                  _p(o["max_overlaps"], _F), _p(o["gt_assignment"], _I), _p(o["counts"], _I),
                                                  _p(o["status"], _I))
        return o


class _Ref:
    """numpy wrappers around the reference's own functions compiled for the host (oracle/_ref)."""

    def __init__(self, so):
        self.lib = ctypes.CDLL(so)

    def boxes_overlap_bev(self, a, b):
        a, b = _f32(a), _f32(b)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        self.lib.ref_boxes_overlap_bev(_p(a, _F), a.shape[0], _p(b, _F), b.shape[0], _p(out, _F))
        return out

    def boxes_iou_bev(self, a, b):
        a, b = _f32(a), _f32(b)
        out = np.zeros((a.shape[0], b.shape[0]), np.float32)
        self.lib.ref_boxes_iou_bev(_p(a, _F), a.shape[0], _p(b, _F), b.shape[0], _p(out, _F))
        return out

    def nms(self, boxes_sorted, thresh, kind="rotated"):
        boxes = _f32(boxes_sorted)
        keep = np.zeros((boxes.shape[0],), np.int64)
        fn = self.lib.ref_nms if kind == "rotated" else self.lib.ref_nms_normal
        n = fn(_p(boxes, _F), boxes.shape[0], ctypes.c_float(thresh), _p(keep, _L))
        return keep[:n].copy()

    def roipool3d_gpu(self, xyz, boxes3d, feat, S, slow=False):
        xyz, boxes3d, feat = _f32(xyz), _f32(boxes3d), _f32(feat)
        B, N, _ = xyz.shape
        M, C = boxes3d.shape[1], feat.shape[2]
        out = np.zeros((B, M, S, 3 + C), np.float32)
        empty = np.zeros((B, M), np.int32)
        self.lib.ref_roipool3d_gpu(_p(xyz, _F), _p(boxes3d, _F), _p(feat, _F), B, N, M, C, S, _p(out, _F),
                                   _p(empty, _I), int(slow))
        return out, empty

    def pts_in_boxes3d_cpu(self, pts, boxes3d):
        pts, boxes3d = _f32(pts), _f32(boxes3d)
        N, M = pts.shape[0], boxes3d.shape[0]
        flags = np.zeros((M, N), np.int64)
        self.lib.ref_pts_in_boxes3d_cpu(_p(flags, _L), _p(pts, _F), _p(boxes3d, _F), N, M)
        return flags

    def roipool3d_cpu(self, pts, boxes3d, feat, S):
        pts, boxes3d, feat = _f32(pts), _f32(boxes3d), _f32(feat)
        N, M, C = pts.shape[0], boxes3d.shape[0], feat.shape[1]
        pp = np.zeros((M, S, 3), np.float32)
        pf = np.zeros((M, S, C), np.float32)
        empty = np.zeros((M,), np.int64)
        self.lib.ref_roipool3d_cpu(_p(pts, _F), _p(boxes3d, _F), _p(feat, _F), N, M, C, S, _p(pp, _F), _p(pf, _F),
                                   _p(empty, _L))
        return pp, pf, empty


def scene_project(raw, calib24, H, W, scope):
    """oracle of calibration.py:51-70 + get_valid_flag for one frame -> rect (n,3), img (n,2), depth (n), flag (n) bool"""
    lib = cpu().lib
    raw, calib24 = _f32(raw), _f32(calib24)
    n = raw.shape[0]
    rect, img, depth, flag = np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
    sc = None if scope is None else (ctypes.c_double * 6)(*[float(v) for v in scope])
    lib.prcnn_cpu_scene_project(_p(raw, _F), n, _p(calib24, _F), int(H), int(W), sc, _p(rect, _F), _p(img, _F), _p(depth, _F), _p(flag, _I))
    return rect, img, depth, flag.astype(bool)


def scene_prepare(raw, offsets, calib, img_hw, scope, npoints, seed):
    """oracle of prcnn_scene_prepare -> xyz (B,npoints,3), intensity (B,npoints), src (B,npoints), nvalid (B), status (B)"""
    lib = cpu().lib
    raw, calib, img_hw = _f32(raw), _f32(calib), _i32(img_hw)
    offsets = np.ascontiguousarray(offsets, np.int64)
    B = len(offsets) - 1
    xyz, inten = np.zeros((B, npoints, 3), np.float32), np.zeros((B, npoints), np.float32)
    src, nvalid, status = np.zeros((B, npoints), np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
    sc = None if scope is None else (ctypes.c_double * 6)(*[float(v) for v in scope])
    lib.prcnn_cpu_scene_prepare(_p(raw, _F), _p(offsets, _L), B, _p(calib, _F), _p(img_hw, _I), sc, int(npoints),
                                ctypes.c_uint32(int(seed) & 0xFFFFFFFF), _p(xyz, _F), _p(inten, _F), _p(src, _I), _p(nvalid, _I), _p(status, _I))
    return xyz, inten, src, nvalid, status


class KittiBackend:
    """compute backend for pointrcnn_amd.kitti_eval running on the CPU oracle (tests only)"""

    def __init__(self):
        self.lib = cpu().lib

    def overlaps(self, metric, dt, dt_off, gt, gt_off, ov_off):
        D = ctypes.POINTER(ctypes.c_double)
        dt, gt = np.ascontiguousarray(dt, np.float64), np.ascontiguousarray(gt, np.float64)
        dt_off, gt_off = _i32(dt_off), _i32(gt_off)
        ov_off = np.ascontiguousarray(ov_off, np.int64)
        out = np.zeros(int(ov_off[-1]), np.float64)
        self.lib.prcnn_cpu_kitti_overlaps(int(metric), _p(dt, D), _p(dt_off, _I), _p(gt, D), _p(gt_off, _I), _p(ov_off, _L),
                                          len(dt_off) - 1, _p(out, D))
        return out

    def statistics(self, overlaps, ov_off, gt_datas, gt_off, dt_datas, dt_off, ign_gt, ign_det, dc, dc_off, metric, min_overlap,
                   thresholds, compute_fp, compute_aos):
        D = ctypes.POINTER(ctypes.c_double)
        f64 = lambda a: np.ascontiguousarray(a, np.float64)      # noqa: E731
        overlaps, gt_datas, dt_datas, dc, thresholds = f64(overlaps), f64(gt_datas), f64(dt_datas), f64(dc), f64(thresholds)
        gt_off, dt_off, dc_off, ign_gt, ign_det = _i32(gt_off), _i32(dt_off), _i32(dc_off), _i32(ign_gt), _i32(ign_det)
        ov_off = np.ascontiguousarray(ov_off, np.int64)
        F, T = len(gt_off) - 1, len(thresholds)
        res = np.zeros((F, T, 4), np.float64)
        matched = np.full(int(gt_off[-1]), np.nan, np.float64)
        self.lib.prcnn_cpu_kitti_statistics(_p(overlaps, D), _p(ov_off, _L), _p(gt_datas, D), _p(gt_off, _I), _p(dt_datas, D),
                                            _p(dt_off, _I), _p(ign_gt, _I), _p(ign_det, _I), _p(dc, D), _p(dc_off, _I), F, int(metric),
                                            ctypes.c_double(min_overlap), _p(thresholds, D), T, int(compute_fp), int(compute_aos),
                                            _p(res, D), _p(matched, D))
        return res, matched

    def rotate_iou_eval(self, boxes, query, criterion=-1):
        boxes, query = _f32(boxes), _f32(query)
        out = np.zeros((boxes.shape[0], query.shape[0]), np.float32)
        self.lib.prcnn_cpu_rotate_iou_eval(_p(boxes, _F), boxes.shape[0], _p(query, _F), query.shape[0], int(criterion), _p(out, _F))
        return out


_cpu = None
_ref = None


def cpu():
    global _cpu
    if _cpu is None:
        _cpu = _Cpu()
    return _cpu


def ref():
    """reference-compiled functions, or None when oracle/_ref has not been built."""
    global _ref
    if _ref is None:
        build()
        so = os.path.join(_HERE, "_ref", "libprcnn_ref.so")
        if not os.path.exists(so):
            return None
        _ref = _Ref(so)
    return _ref
