"""bench_reference.py -- frames/s of the reference's OWN, UNCHANGED caller on the drop-in (`bench.py --workload reference`).

`north_star`: "... exposed through the same pointnet2_lib / roipool3d / iou3d Python op surface so lib/net and tools/eval_rcnn.py
call it unchanged".  bench.py's headline is measured on the mirror (pointrcnn_amd/rpn.py under pipeline.py's hipGraphs); this file
measures what a user gets who changes NOTHING: the reference's `lib/net/point_rcnn.py` `PointRCNN(num_classes=2, use_xyz=True,
mode='TEST')` (tools/eval_rcnn.py:882), RPN stage, imported as it is from the reference tree with `sys.path` set the way
tools/_init_path.py:1-4 sets it, on `pointrcnn_amd/dropin`, driven exactly as the RPN evaluation loop of tools/eval_rcnn.py drives it
(:153-166, `--test`): one batch in flight, eager launches on torch's current stream, per batch

    inputs = torch.from_numpy(pts_input).cuda(non_blocking=True).float()          eval_rcnn.py:153        "h2d"
    ret_dict = model({'pts_input': inputs})                                       :157  lib/net/rpn.py:68-82   "backbone" + "heads"
    rpn_scores = torch.sigmoid(rpn_cls[:, :, 0]); seg_result = (rpn_scores > thr).long()     :161-163       "seg"
    rois, roi_scores_raw = model.rpn.proposal_layer(rpn_scores_raw, rpn_reg, backbone_xyz)   :166  lib/rpn/proposal_layer.py:35-119
    rois / scores / seg -> host (what `--save_result` or the joint loop :559ff reads back per batch)              "d2h"

The reference tree is NOT part of this repository and does not exist on the GPU box: it is read from $PRCNN_REFERENCE /
/root/reference when present, else from the archive `oracle/stage_reference.py` stages under oracle/_ref/ (git-ignored built
artefact, the same one tests/test_gpu_reference_unchanged.py unpacks).  Here the reference is the CALLER being timed, not a checker:
nothing of it enters the product, and no oracle code (CPU restatement) runs in this path -- every operator underneath is a C-ABI call
into libprcnn_pointops.so.  Packages of the reference's environment that this image lacks (easydict, tensorboardX, ...) come from
tests/compat/ (stand-ins for third-party packages, not reference code).

Two figures, both frames/s on one GPU, bs32 x 16 384 points, inputs in pinned host memory (the loop's H2D copy is inside):
  model_only  -- h2d + model(input_data) + synchronize: SURVEY 8(d)(i)'s "RPN inference end-to-end (H2D + backbone + heads)";
  eval_loop   -- the whole loop body above, proposal layer and read-back included.
plus the per-stage milliseconds of one batch with a synchronize between stages.
"""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
_STATE = {}


def _reference_tree():
    """-> directory holding the reference's lib/ and tools/cfgs/ (see the module docstring); None when there is none"""
    if "tree" in _STATE:
        return _STATE["tree"]
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    from oracle import stage_reference          # only locate(): unpacks the staged archive, builds / runs nothing
    tmp = tempfile.mkdtemp(prefix="prcnn_reference_")
    _STATE["tree"] = stage_reference.locate(tmp)
    return _STATE["tree"]


def available():
    return _reference_tree() is not None


def load(cfg_file="tools/cfgs/default.yaml"):
    """import the reference's lib.config / lib.net.point_rcnn with sys.path as tools/_init_path.py:1-4 sets it; -> (cfg, PointRCNN)"""
    if "ns" in _STATE:
        return _STATE["ns"]
    import pointrcnn_amd
    pointrcnn_amd.install()
    tree = _reference_tree()
    if tree is None:
        raise RuntimeError("no reference tree: neither $PRCNN_REFERENCE / /root/reference nor oracle/_ref/reference_py.tar.gz exists")
    for p in (os.path.join(ROOT, "tests", "compat"), os.path.join(tree, "lib", "net"), os.path.join(tree, "lib", "datasets"), tree):
        if p not in sys.path:
            sys.path.insert(0, p)
    import yaml
    _load = yaml.load
    yaml.load = lambda f, Loader=yaml.SafeLoader: _load(f, Loader=Loader)      # lib/config.py:187 predates PyYAML 6
    sys.dont_write_bytecode = True
    try:
        from lib.config import cfg, cfg_from_file
        cfg_from_file(os.path.join(tree, cfg_file))
    finally:
        yaml.load = _load
    from lib.net.point_rcnn import PointRCNN
    _STATE["ns"] = (cfg, PointRCNN)
    return _STATE["ns"]


def build_model(dev, mirror_rpn=None, nms_type="normal"):
    """the reference's PointRCNN, RPN stage only (tools/eval_rcnn.py --eval_mode rpn: cfg.RPN.ENABLED, not cfg.RCNN.ENABLED :864-866),
    parameters and BatchNorm statistics copied from `mirror_rpn` (pointrcnn_amd.rpn.RPN: same state-dict keys) when given"""
    cfg, PointRCNN = load()
    cfg.RPN.ENABLED, cfg.RCNN.ENABLED, cfg.RPN.FIXED = True, False, False
    cfg.RPN.NMS_TYPE = nms_type
    with torch.cuda.device(dev):
        model = PointRCNN(num_classes=2, use_xyz=True, mode="TEST")
    if mirror_rpn is not None:
        missing = model.rpn.load_state_dict(mirror_rpn.state_dict(), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    model.cuda()                                                      # eval_rcnn.py:883
    model.eval()                                                      # eval_one_epoch_rpn :101
    return cfg, model


def _batch(cfg, model, pts_input, want_host=True):
    """one iteration of the evaluation loop's body, eval_rcnn.py:153-166 (+ the read-back); `pts_input` is the dataloader's numpy batch"""
    inputs = torch.from_numpy(pts_input).cuda(non_blocking=True).float()
    input_data = {"pts_input": inputs}
    ret_dict = model(input_data)
    rpn_cls, rpn_reg = ret_dict["rpn_cls"], ret_dict["rpn_reg"]
    backbone_xyz = ret_dict["backbone_xyz"]
    rpn_scores_raw = rpn_cls[:, :, 0]
    rpn_scores = torch.sigmoid(rpn_scores_raw)
    seg_result = (rpn_scores > cfg.RPN.SCORE_THRESH).long()
    rois, roi_scores_raw = model.rpn.proposal_layer(rpn_scores_raw, rpn_reg, backbone_xyz)
    if want_host:
        return rois.cpu().numpy(), roi_scores_raw.cpu().numpy(), seg_result.cpu().numpy()
    return rois, roi_scores_raw, seg_result


def measure(dev, clouds_cpu, mirror_rpn=None, steps=12, warmup=3, nms_type="normal", mirror_out=None):
    """-> dict for the bench line.  clouds_cpu: (B, N, 3) float32 CPU tensor (the batch bench.py's slot 0 holds)."""
    cfg, model = build_model(dev, mirror_rpn, nms_type)
    B, N = clouds_cpu.shape[:2]
    host = [clouds_cpu.numpy().copy() for _ in range(2)]              # the dataloader hands over numpy batches (collate_batch)
    out = {"caller": "reference lib/net/point_rcnn.py PointRCNN(mode='TEST'), RPN stage, unchanged; loop body of tools/eval_rcnn.py:153-166 "
                     "(--test), one batch in flight, eager, torch's current stream",
           "batch": B, "npoints": N, "nms_type": nms_type, "steps": steps}
    with torch.no_grad():                                             # eval_rcnn.py:897
        for k in range(warmup):
            res = _batch(cfg, model, host[k % 2])
        torch.cuda.synchronize()
        if mirror_out is not None:
            # same parameters, same clouds: the reference route and the mirror must agree (backbone: same modules, same kernels)
            chk = model({"pts_input": torch.from_numpy(host[0]).to(dev)})
            out["backbone_bit_identical_to_mirror"] = bool(torch.equal(chk["backbone_features"], mirror_out["backbone_features"]))
            d = (chk["rpn_reg"] - mirror_out["rpn_reg"]).abs().max().item()
            out["rpn_reg_max_abs_diff_vs_mirror"] = d
        # (1) H2D + model(input_data): the headline's definition of RPN inference end to end
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            inputs = torch.from_numpy(host[k % 2]).cuda(non_blocking=True).float()
            ret = model({"pts_input": inputs})
            torch.cuda.current_stream().synchronize()              # the caller reads the result: one synchronisation per batch
        e_model = time.perf_counter() - t0
        # (2) the whole loop body
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            res = _batch(cfg, model, host[k % 2])
        torch.cuda.synchronize()
        e_loop = time.perf_counter() - t0
        # (3) per-stage milliseconds (a synchronize between stages; median of `steps` batches)
        stages = {k: [] for k in ("h2d", "backbone", "heads", "seg", "proposal_layer", "d2h")}
        rpn = model.rpn
        for k in range(steps):
            sync = torch.cuda.synchronize
            sync(); t = time.perf_counter()
            inputs = torch.from_numpy(host[k % 2]).cuda(non_blocking=True).float()
            sync(); t1 = time.perf_counter(); stages["h2d"].append(t1 - t); t = t1
            backbone_xyz, backbone_features = rpn.backbone_net(inputs)                                # lib/net/rpn.py:73
            sync(); t1 = time.perf_counter(); stages["backbone"].append(t1 - t); t = t1
            rpn_cls = rpn.rpn_cls_layer(backbone_features).transpose(1, 2).contiguous()               # :75
            rpn_reg = rpn.rpn_reg_layer(backbone_features).transpose(1, 2).contiguous()               # :76
            sync(); t1 = time.perf_counter(); stages["heads"].append(t1 - t); t = t1
            rpn_scores_raw = rpn_cls[:, :, 0]
            seg_result = (torch.sigmoid(rpn_scores_raw) > cfg.RPN.SCORE_THRESH).long()
            sync(); t1 = time.perf_counter(); stages["seg"].append(t1 - t); t = t1
            rois, roi_scores_raw = rpn.proposal_layer(rpn_scores_raw, rpn_reg, backbone_xyz)
            sync(); t1 = time.perf_counter(); stages["proposal_layer"].append(t1 - t); t = t1
            res = (rois.cpu().numpy(), roi_scores_raw.cpu().numpy(), seg_result.cpu().numpy())
            sync(); t1 = time.perf_counter(); stages["d2h"].append(t1 - t)
        if os.environ.get("PRCNN_REF_PROFILE"):
            # where the host time of the eager loop goes (development aid): cProfile of three loop bodies, top entries to stderr
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            for k in range(3):
                _batch(cfg, model, host[k % 2])
            pr.disable()
            pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
    med = lambda v: sorted(v)[len(v) // 2]       # noqa: E731
    out["value_model_only"] = round(B * steps / e_model, 1)
    out["model_only_ms_per_batch"] = round(1e3 * e_model / steps, 3)
    out["value_eval_loop"] = round(B * steps / e_loop, 1)
    out["eval_loop_ms_per_batch"] = round(1e3 * e_loop / steps, 3)
    out["stage_ms"] = {k: round(1e3 * med(v), 3) for k, v in stages.items()}
    out["rois_shape"] = list(res[0].shape)
    return out


if __name__ == "__main__":
    # stand-alone: python bench_reference.py [steps]
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
    import json
    sys.path.insert(0, ROOT)
    from pointrcnn_amd import rpn as _rpn
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    mirror = _rpn.randomize_bn_stats(_rpn.RPN(), seed=7).to(dev).eval()
    clouds = _rpn.synthetic_clouds(32, 16384, seed0=100)
    with torch.no_grad():
        mo = mirror({"pts_input": clouds.to(dev)})
    print(json.dumps(measure(dev, clouds, mirror, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 12, mirror_out=mo)))
