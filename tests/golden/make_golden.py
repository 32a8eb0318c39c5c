#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

    iou3d_ref.npz, roipool3d_ref.npz -- outputs of the REFERENCE'S OWN native code (lib/utils/iou3d/src/*,
        lib/utils/roipool3d/src/* compiled for the host by oracle/build_ref.py -> oracle/_ref).  Needs
        /root/reference; the fixtures travel to the GPU box, the reference does not.
    proposal_ref.npz -- outputs of the REFERENCE'S OWN Python proposal stage (lib/utils/bbox_transform.py
        decode_bbox_target, lib/rpn/proposal_layer.py ProposalLayer) imported on CPU by ref_proposal.py, with the NMS
        extension calls routed to oracle/_ref.  Inputs are regenerated from seeds (tests/util.py); their CRCs are stored.
    canonical_ref.npz -- rcnn_net.py:143-150 (centre subtract + kitti_utils.rotate_pc_along_y_torch) run by the reference's
        own Python on the reference's pooled tensor.
    kitti_eval_ref.npz -- outputs of the reference's own tools/kitti_object_eval_python evaluator (result strings, AP
        dictionaries, precision / recall arrays, per-frame overlaps, rotate_iou_gpu_eval matrices) on seeded synthetic
        annotations; numba is stubbed to identity (ref_kitti_eval.py), the code that runs is the reference's.
    pointnet2_oracle.npz -- outputs of the CPU oracle (oracle/prcnn_oracle.c) for the PointNet++ ops, whose
        reference source is an empty git submodule (parity unpinned: these pin the ORACLE's behaviour across
        refactors, not the reference's).

Seeds are fixed; re-running reproduces the files byte for byte.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from util import enlarge, kitti_cloud, rand_bev, rand_boxes3d, rpn_like_scene, unit_cloud  # noqa: E402


def proposal_cases():
    """name -> kwargs shared by make_golden.py and the tests (inputs are regenerated from these seeds)"""
    return {
        "test_normal": dict(B=2, N=16384, seed=1, mode="TEST", nms="normal", z_max=70.4, distance=True),
        "test_rotate": dict(B=2, N=16384, seed=1, mode="TEST", nms="rotate", z_max=70.4, distance=True),
        "train_normal": dict(B=1, N=16384, seed=2, mode="TRAIN", nms="normal", z_max=70.4, distance=True),
        "near_only": dict(B=1, N=16384, seed=3, mode="TEST", nms="normal", z_max=38.0, distance=True),
        "sparse": dict(B=2, N=2000, seed=4, mode="TEST", nms="rotate", z_max=70.4, distance=True),
        "score_based": dict(B=1, N=4096, seed=5, mode="TEST", nms="rotate", z_max=70.4, distance=False),
    }


def crc(*arrays):
    import zlib
    c = 0
    for a in arrays:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return np.uint32(c)


def make_proposal_golden():
    import torch
    import ref_proposal
    cfg, decode, make_pl, anchor = ref_proposal.load()
    out = {}
    for name, c in proposal_cases().items():
        xyz, sc, reg = rpn_like_scene(c["B"], c["N"], seed=c["seed"], z_max=c["z_max"])
        cfg.RPN.NMS_TYPE = c["nms"]
        cfg.TEST.RPN_DISTANCE_BASED_PROPOSE = c["distance"]
        rois, rs = make_pl(c["mode"])(torch.from_numpy(sc), torch.from_numpy(reg), torch.from_numpy(xyz))
        out[name + "_rois"], out[name + "_scores"], out[name + "_crc"] = rois.numpy(), rs.numpy(), crc(xyz, sc, reg)
        print(name, "filled", (rois.abs().sum(-1) > 0).sum(1).tolist())
    cfg.TEST.RPN_DISTANCE_BASED_PROPOSE = True
    # decode_bbox_target alone, the RPN layout (roi = xyz) and the RCNN layouts (roi = boxes, eval_rcnn.py:509-517)
    xyz, sc, reg = rpn_like_scene(1, 4096, seed=6)
    t = torch.from_numpy
    out["dec_rpn"] = decode(t(xyz[0]), t(reg[0]), anchor_size=anchor(), loc_scope=3.0, loc_bin_size=0.5, num_head_bin=12,
                            get_xz_fine=True, get_y_by_bin=False, get_ry_fine=False).numpy()
    out["dec_rpn_coarse"] = decode(t(xyz[0]), t(np.ascontiguousarray(reg[0][:, 24:])), anchor_size=anchor(), loc_scope=3.0,
                                   loc_bin_size=0.5, num_head_bin=12, get_xz_fine=False).numpy()
    r = np.random.default_rng(7)
    rois7 = rand_boxes3d(xyz[0], 512, seed=8)
    reg46 = r.normal(0, 1, (512, 46)).astype(np.float32)
    reg53 = r.normal(0, 1, (512, 53)).astype(np.float32)
    out["dec_rois7"], out["dec_reg46"], out["dec_reg53"] = rois7, reg46, reg53
    out["dec_rcnn"] = decode(t(rois7.copy()), t(reg46), anchor_size=anchor(), loc_scope=1.5, loc_bin_size=0.5, num_head_bin=9,
                             get_xz_fine=True, get_y_by_bin=False, loc_y_scope=0.5, loc_y_bin_size=0.25, get_ry_fine=True).numpy()
    out["dec_rcnn_ybin"] = decode(t(rois7.copy()), t(reg53), anchor_size=anchor(), loc_scope=1.5, loc_bin_size=0.5,
                                  num_head_bin=9, get_xz_fine=True, get_y_by_bin=True, loc_y_scope=0.5, loc_y_bin_size=0.25,
                                  get_ry_fine=True).numpy()
    out["dec_crc"] = crc(xyz, reg)
    np.savez_compressed(os.path.join(HERE, "proposal_ref.npz"), **out)


def kitti_eval_inputs():
    """shared by make_golden.py and the tests"""
    from util import kitti_annos
    gt = kitti_annos(120, seed=1)
    dt = kitti_annos(120, seed=2, with_score=True, gt=gt)
    return gt, dt


def annos_crc(annos):
    return crc(*[np.asarray(a[k], np.float64) for a in annos for k in ("bbox", "dimensions", "location", "rotation_y", "alpha", "score")])


def make_kitti_eval_golden():
    """outputs of the reference's own tools/kitti_object_eval_python (numba stubbed, see ref_kitti_eval.py)"""
    import ref_kitti_eval
    ev, kc = ref_kitti_eval.load()
    gt, dt = kitti_eval_inputs()
    out = {"crc": np.array([annos_crc(gt), annos_crc(dt)])}
    res, d = ev.get_official_eval_result(gt, dt, [0, 1, 2])
    out["official_str"] = np.array(res)
    out["official_keys"] = np.array(sorted(d))
    out["official_vals"] = np.array([d[k] for k in sorted(d)])
    out["official_car_str"] = np.array(ev.get_official_eval_result(gt, dt, 0)[0])
    _lin = np.linspace
    np.linspace = lambda a, b, n=50, **k: _lin(a, b, int(n), **k)        # eval.py:593 passes num as a float
    out["coco_str"] = np.array(ev.get_coco_eval_result(gt, dt, [0, 1]))
    np.linspace = _lin
    mo = np.stack([np.array([[0.7, 0.5, 0.5]] * 3), np.array([[0.5, 0.25, 0.25]] * 3)], 0)
    for metric in (0, 1, 2):
        blocks = ev.calculate_iou_partly(dt, gt, metric, 50)[0]
        out["ov%d" % metric] = np.concatenate([b.reshape(-1) for b in blocks]) if blocks else np.zeros(0)
        r = ev.eval_class(gt, dt, [0, 1, 2], [0, 1, 2], metric, mo, compute_aos=(metric == 0))
        for k in ("recall", "precision", "orientation"):
            out["m%d_%s" % (metric, k)] = r[k]
    rr = np.random.default_rng(5)
    boxes = np.concatenate([rr.uniform(-4, 4, (40, 2)), rr.uniform(1, 4, (40, 2)), rr.uniform(-3.2, 3.2, (40, 1))], 1).astype(np.float32)
    query = np.concatenate([rr.uniform(-4, 4, (30, 2)), rr.uniform(1, 4, (30, 2)), rr.uniform(-3.2, 3.2, (30, 1))], 1).astype(np.float32)
    query[:5] = boxes[:5]                                                 # identical boxes: degenerate clipping
    query[5, :4], query[5, 4] = boxes[5, :4], boxes[5, 4] + np.float32(np.pi / 2)
    out["riou_boxes"], out["riou_query"] = boxes, query
    for c in (-1, 0, 1, 2):
        out["riou_c%d" % c] = ev.rotate_iou_gpu_eval(boxes, query, c)
    np.savez_compressed(os.path.join(HERE, "kitti_eval_ref.npz"), **out)


def make_canonical_golden(ref):
    """rcnn_net.py:143-150 run with the reference's own kitti_utils.rotate_pc_along_y_torch on the reference's pooled tensor"""
    import torch
    import ref_proposal
    ref_proposal.load()
    import lib.utils.kitti_utils as kitti_utils
    g = np.load(os.path.join(HERE, "roipool3d_ref.npz"))
    xyz, boxes, feat, S = g["xyz"], g["boxes"], g["feat"], int(g["S"])
    rois = boxes.copy()
    rois[..., 3:6] -= 2.0                                # the un-enlarged RoIs (enlarge_box3d(rois, 1.0) == boxes)
    rois[..., 1] -= 1.0
    pooled, empty = ref.roipool3d_gpu(xyz, boxes, feat, S)
    pf = torch.from_numpy(pooled.copy())
    batch_rois = torch.from_numpy(rois)
    pf[:, :, :, 0:3] -= batch_rois[:, :, 0:3].unsqueeze(dim=2)
    for k in range(pf.shape[0]):
        pf[k, :, :, 0:3] = kitti_utils.rotate_pc_along_y_torch(pf[k, :, :, 0:3], batch_rois[k, :, 6])
    np.savez_compressed(os.path.join(HERE, "canonical_ref.npz"), rois=rois, pooled_canonical=pf.numpy(), empty=empty)


def main():
    cpu, ref = oracle.cpu(), oracle.ref()
    if ref is None:
        raise SystemExit("oracle/_ref is not built (needs /root/reference)")
    # ---- iou3d: reference code
    a, b = rand_bev(120, 5.0, seed=1), rand_bev(90, 5.0, seed=2)
    nms_boxes = rand_bev(700, 7.0, seed=3)
    out = dict(a=a, b=b, overlap=ref.boxes_overlap_bev(a, b), iou=ref.boxes_iou_bev(a, b), nms_boxes=nms_boxes)
    for kind in ("rotated", "normal"):
        for thr in (0.1, 0.5, 0.8):
            out["keep_%s_%s" % (kind, thr)] = ref.nms(nms_boxes, thr, kind)
    np.savez_compressed(os.path.join(HERE, "iou3d_ref.npz"), **out)
    # ---- roipool3d: reference CPU code
    N, M, C, S = 3000, 24, 8, 64
    xyz = kitti_cloud(1, N, seed=42)
    boxes = enlarge(rand_boxes3d(xyz[0], M, seed=7), 1.0)
    boxes[-1, 0] += 300.0            # an empty box
    feat = np.random.default_rng(9).normal(size=(N, C)).astype(np.float32)
    pp, pf, pe = ref.roipool3d_cpu(xyz[0], boxes, feat, S)
    flags = ref.pts_in_boxes3d_cpu(xyz[0], boxes)
    np.savez_compressed(os.path.join(HERE, "roipool3d_ref.npz"), xyz=xyz, boxes=boxes[None], feat=feat[None], S=S,
                        pooled_pts=pp, pooled_feat=pf, empty=pe, flags=flags.astype(np.uint8))
    # ---- PointNet++ ops: oracle (unpinned upstream)
    pts = unit_cloud(2, 1024, seed=11)
    fidx = cpu.fps(pts, 128)
    new_xyz = np.stack([pts[b][fidx[b]] for b in range(2)])
    bq = cpu.ball_query(0.2, 16, pts, new_xyz)
    d2, i3 = cpu.three_nn(pts[:, :300], new_xyz)
    np.savez_compressed(os.path.join(HERE, "pointnet2_oracle.npz"), xyz=pts, fps_idx=fidx, ball_idx=bq,
                        nn_dist2=d2, nn_idx=i3, nn_w=cpu.three_weights(d2))
    make_proposal_golden()
    make_canonical_golden(ref)
    make_kitti_eval_golden()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
