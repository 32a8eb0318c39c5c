"""TEST INFRASTRUCTURE: subprocess side of tests/test_gpu_eval_rcnn_unchanged.py.  Run from `<copy of the reference>/tools` (the cwd
the reference's scripts assume: `cfgs/default.yaml`, `../data`, `../output`) with PYTHONPATH = tests/compat : pointrcnn_amd/dropin :
the repo root -- the environment INTEGRATION.md section 1 describes.

    python eval_rcnn_helpers.py ckpt   <out.pth> <seed>          the reference's own PointRCNN, seeded parameters, saved with the
                                                                 reference's own checkpoint_state / save_checkpoint
                                                                 (tools/train_utils/train_utils.py:60-76)
    python eval_rcnn_helpers.py mirror <ckpt.pth> <out_dir> <bs> this package's mirror of the detector + kitti_output on the SAME inputs
                                                                 (the reference's own KittiRCNNDataset, same numpy seeds and loader
                                                                 settings as tools/eval_rcnn.py:459-470,856-871), one result file
                                                                 per frame
    python eval_rcnn_helpers.py cpu-run <script.py> [args...]    build-container only (no GPU): run a reference script unchanged with
                                                                 `.cuda()` a no-op and the operators routed to the CPU oracle
"""
import logging
import os
import runpy
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _reference_paths():
    sys.path.insert(0, os.getcwd())
    import _init_path  # noqa: F401  (tools/_init_path.py:1-4: the reference root, lib/datasets, lib/net)


def _rcnn_mode_cfg():
    """what tools/eval_rcnn.py:875-890 sets for `--cfg_file cfgs/default.yaml --eval_mode rcnn`"""
    from lib.config import cfg, cfg_from_file
    cfg_from_file("cfgs/default.yaml")
    cfg.TAG = "default"
    cfg.RCNN.ENABLED = True
    cfg.RPN.ENABLED = cfg.RPN.FIXED = True
    return cfg


def make_ckpt(out, seed):
    import cpu_ops
    _reference_paths()
    _rcnn_mode_cfg()
    from lib.net.point_rcnn import PointRCNN
    import tools.train_utils.train_utils as train_utils
    import contextlib
    with (contextlib.nullcontext() if torch.cuda.is_available() else cpu_ops.cuda_is_cpu()):      # ProposalLayer.__init__ calls .cuda()
        model = PointRCNN(num_classes=2, use_xyz=True, mode="TEST")
    cpu_ops.fill_params_by_name(model, int(seed))
    assert out.endswith(".pth")
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    train_utils.save_checkpoint(train_utils.checkpoint_state(model, None, epoch=7, it=123), filename=out[:-4])
    print("saved %s (%d tensors)" % (out, len(model.state_dict())))


def mirror_eval(ckpt, out_dir, batch_size):
    from torch.utils.data import DataLoader
    _reference_paths()
    cfg = _rcnn_mode_cfg()
    from lib.datasets.kitti_rcnn_dataset import KittiRCNNDataset
    from pointrcnn_amd import kitti_output
    from pointrcnn_amd.point_rcnn import PointRCNN
    np.random.seed(1024)                                                          # eval_rcnn.py:26
    ds = KittiRCNNDataset(root_dir=os.path.join("..", "data"), npoints=cfg.RPN.NUM_POINTS, split=cfg.TEST.SPLIT, mode="EVAL",
                          random_select=True, classes=cfg.CLASSES, logger=logging.getLogger("mirror"))
    loader = DataLoader(ds, batch_size=int(batch_size), shuffle=False, pin_memory=True, num_workers=0, collate_fn=ds.collate_batch)
    model = PointRCNN(num_classes=ds.num_class, use_xyz=True, mode="TEST").cuda()
    state = torch.load(ckpt)
    model.load_state_dict(state["model_state"])                                   # train_utils.py:85: strict, same key names
    model.eval()
    os.makedirs(out_dir, exist_ok=True)
    np.random.seed(666)                                                           # eval_rcnn.py:460
    total = 0
    with torch.no_grad():
        for data in loader:
            ids = [int(i) for i in data["sample_id"]]
            inputs = torch.from_numpy(data["pts_input"]).cuda(non_blocking=True).float()
            out = model({"pts_input": inputs})
            pred, raw, keep, num = model.detections(out, score_thresh=cfg.RCNN.SCORE_THRESH, nms_thresh=cfg.RCNN.NMS_THRESH)
            lines = kitti_output.write_detections(pred, raw, keep, num, ids, [ds.get_calib(i) for i in ids],
                                                  [ds.get_image_shape(i) for i in ids], out_dir, class_name=cfg.CLASSES)
            total += sum(len(ln) for ln in lines)
    print("mirror: %d frames, %d detections -> %s" % (len(ds), total, out_dir))


def cpu_run(script, argv):
    import cpu_ops
    assert not torch.cuda.is_available(), "cpu-run is the build container's route; on the GPU box run the script itself"
    sys.argv = [script] + list(argv)
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    with cpu_ops.oracle_ops(), cpu_ops.cuda_is_cpu():
        runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "ckpt":
        make_ckpt(sys.argv[2], sys.argv[3])
    elif cmd == "mirror":
        mirror_eval(sys.argv[2], sys.argv[3], sys.argv[4])
    elif cmd == "cpu-run":
        cpu_run(sys.argv[2], sys.argv[3:])
    else:
        raise SystemExit(__doc__)
