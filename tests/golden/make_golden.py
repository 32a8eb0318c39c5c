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

    net_ref.npz -- outputs of the reference's own unchanged lib/net PointRCNN(mode='TEST') forward (RPN -> proposal layer ->
        roipool3d -> RCNN) run on the drop-in op surface on CPU (ref_net.py: operators routed to the oracle, graph code = the
        reference's Python), parameters a pure function of their state-dict names.
    train_ref.npz -- the reference's own loss code (lib/net/train_functions.py + lib/utils/loss_utils.py) on seeded network
        outputs, and one whole RPN training forward/backward of the reference's PointRCNN(mode='TRAIN') on the drop-in surface.

    labels_ref.npz -- the reference's own generate_rpn_training_labels (numpy + scipy Delaunay) on seeded dense scenes.

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


NET_CASE = dict(B=2, N=16384, seed0=100, wseed=5)
TRAIN_CASE = dict(B=2, N=8192, seed0=300, wseed=9, G=6)


def train_batch(c=TRAIN_CASE):
    """seeded synthetic RPN training batch: clouds + a few car-sized GT boxes centred on cloud points -> labels via the oracle
    point-in-box test (shared by make_golden.py and the tests)"""
    import torch
    from pointrcnn_amd import rpn as mrpn
    pts = mrpn.synthetic_clouds(c["B"], c["N"], seed0=c["seed0"]).numpy()
    cpu = oracle.cpu()
    gt = np.stack([rand_boxes3d(pts[b], c["G"], seed=c["seed0"] + 50 + b, jitter=0.1) for b in range(c["B"])])
    gt[..., 3:6] *= 2.5                                   # oversized boxes so that the sparse synthetic cloud yields foreground points
    cls = np.zeros((c["B"], c["N"]), np.int64)
    reg = np.zeros((c["B"], c["N"], 7), np.float32)
    for b in range(c["B"]):
        inside = cpu.pts_in_boxes3d(pts[b], gt[b]) > 0
        near = cpu.pts_in_boxes3d(pts[b], enlarge(gt[b], 0.2)) > 0
        for k in range(c["G"]):
            fg = inside[k]
            cls[b][fg] = 1
            cls[b][near[k] ^ fg] = -1
            ctr = gt[b, k, 0:3].copy()
            ctr[1] -= gt[b, k, 3] / 2
            reg[b, fg, 0:3] = ctr - pts[b][fg]
            reg[b, fg, 3:7] = gt[b, k, 3:7]
    return pts, gt, cls, reg


def make_net_golden():
    """outputs of the reference's own unchanged PointRCNN(mode='TEST') (lib/net/*.py) on the drop-in op surface, CPU"""
    import torch
    import ref_net
    from pointrcnn_amd import rpn as mrpn
    c = NET_CASE
    model = ref_net.build_reference_model("TEST", seed=c["wseed"])
    pts = mrpn.synthetic_clouds(c["B"], c["N"], seed0=c["seed0"])
    out = ref_net.run_reference(model, pts)
    sd = model.state_dict()
    g = {"keys": np.array(sorted(sd)), "shapes": np.array([str(tuple(sd[k].shape)) for k in sorted(sd)]), "crc_in": crc(pts.numpy())}
    g["rpn_cls"] = out["rpn_cls"].numpy()[:, :, 0]
    g["rpn_reg_s8"] = out["rpn_reg"].numpy()[:, ::8]
    g["rpn_reg_abs_sum"] = np.abs(out["rpn_reg"].numpy().astype(np.float64)).sum((1, 2))
    g["feat_s16"] = out["backbone_features"].numpy()[:, :, ::16]
    for k in ("rois", "roi_scores_raw", "seg_result", "rcnn_cls", "rcnn_reg"):
        g[k] = out[k].numpy()
    np.savez_compressed(os.path.join(HERE, "net_ref.npz"), **g)
    print("net_ref: rois filled", (np.abs(g["rois"]).sum(-1) > 0).sum(1).tolist())


def make_train_golden():
    """(a) the reference's own loss code (train_functions.model_fn + loss_utils) on seeded network outputs: loss, tb_dict,
    gradients; (b) one RPN training forward/backward of the reference's own PointRCNN(mode='TRAIN') + model_fn on the drop-in
    op surface (CPU, oracle-backed operators): loss and per-parameter gradient norms + samples"""
    import torch
    import ref_net
    import cpu_ops
    g = {}
    for name, (seed, loss_cls) in {"focal": (1, "SigmoidFocalLoss"), "dice": (2, "DiceLoss"), "bce": (3, "BinaryCrossEntropy"),
                                   "nofg": (4, "SigmoidFocalLoss")}.items():
        case = ref_net.train_loss_case(seed, nfg=0 if name == "nofg" else 300)
        if name == "nofg":
            case[2][:] = 0
        loss, tb, gc, gr = ref_net.reference_rpn_loss(*case, loss_cls=loss_cls)
        g[name + "_loss"], g[name + "_gcls"], g[name + "_greg_s4"] = np.float64(loss), gc, gr[:, ::4]
        g[name + "_tb_keys"], g[name + "_tb_vals"] = np.array(sorted(tb)), np.array([float(tb[k]) for k in sorted(tb)])
        g[name + "_crc"] = crc(*case)
    # (b) whole training step
    c = TRAIN_CASE
    pts, gt, cls, reg = train_batch(c)
    ns = ref_net.load()
    model = ref_net.build_reference_model("TRAIN", seed=c["wseed"], rpn_only=True)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()                                  # CPU and GPU dropout streams differ: the golden step runs without dropout
    data = {"pts_rect": pts, "pts_features": np.zeros((c["B"], c["N"], 1), np.float32), "pts_input": pts, "gt_boxes3d": gt,
            "rpn_cls_label": cls, "rpn_reg_label": reg}
    with cpu_ops.oracle_ops(), cpu_ops.cuda_is_cpu():
        ret = ns.train_functions.model_joint_fn_decorator()(model, data)
        ret.loss.backward()
    names = [n for n, p in sorted(model.named_parameters()) if p.grad is not None]
    params = dict(model.named_parameters())
    g["step_loss"] = np.float64(ret.loss.item())
    g["step_fg"] = np.int64((cls > 0).sum())
    g["step_names"] = np.array(names)
    g["step_gnorm"] = np.array([float(params[n].grad.double().norm()) for n in names])
    g["step_gsample"] = np.stack([np.resize(params[n].grad.reshape(-1)[:8].numpy(), 8) for n in names])
    g["step_crc"] = crc(pts, gt, cls, reg)
    np.savez_compressed(os.path.join(HERE, "train_ref.npz"), **g)
    print("train_ref: step loss", g["step_loss"], "fg", int(g["step_fg"]), "params with grad", len(names))


def label_scene(seed):
    """dense synthetic scene for the RPN label fixtures: 16384 points in a 16 x 4 x 20 m volume, 10 car-sized GT boxes centred
    near cloud points (some overlap each other)"""
    pts = np.random.default_rng(seed).uniform([-8, -1, 5], [8, 3, 25], (16384, 3)).astype(np.float32)
    return pts, rand_boxes3d(pts, 10, seed=seed + 50, jitter=0.3)


def make_labels_golden():
    """the reference's own KittiRCNNDataset.generate_rpn_training_labels (kitti_rcnn_dataset.py:365-394, scipy Delaunay hull
    test) on seeded scenes"""
    import ref_net
    ref_net.load()
    import lib.datasets.kitti_rcnn_dataset as d
    g = {}
    for seed in (0, 1, 2):
        pts, gt = label_scene(seed)
        cls, reg = d.KittiRCNNDataset.generate_rpn_training_labels(pts, gt)
        g["cls%d" % seed], g["reg%d" % seed], g["crc%d" % seed] = cls.astype(np.int8), reg, crc(pts, gt)
    np.savez_compressed(os.path.join(HERE, "labels_ref.npz"), **g)


def main():
    if "--net" in sys.argv:                       # only the network-level fixtures
        make_net_golden()
        make_train_golden()
        make_labels_golden()
        return
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
    make_net_golden()
    make_train_golden()
    make_labels_golden()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
