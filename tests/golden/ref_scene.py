"""Run the REFERENCE's own input path (lib/utils/calibration.py, lib/datasets/kitti_rcnn_dataset.py) on synthetic scans.

Only used to GENERATE tests/golden/scene_ref.npz (python tests/golden/ref_scene.py) in the build container: it needs
/root/reference.  A throw-away KITTI directory tree (ImageSets / velodyne / calib / image_2) is written under
tests/golden/_scene_tmp, the reference's KittiRCNNDataset is instantiated on it in TEST mode and asked for its samples;
its Calibration class and get_valid_flag are also called directly for the per-point intermediates.
"""
import logging
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from util import KITTI_CALIB_TXT, synthetic_scan          # noqa: E402

REFERENCE = os.environ.get("PRCNN_REFERENCE", "/root/reference")
FRAMES = [dict(n=30000, seed=11, fov=0.5, far=0.12, hw=(375, 1242)),      # more valid points than npoints, many far ones
          dict(n=9000, seed=12, fov=0.35, far=0.2, hw=(370, 1224)),       # fewer valid points than npoints: top-up branch
          dict(n=20000, seed=13, fov=0.6, far=0.05, hw=(376, 1241))]
NPOINTS = 4096
STRIDE = 8                                                 # per-point intermediates are stored for every 8th point


def main():
    for p in (os.path.join(os.path.dirname(HERE), "compat"), REFERENCE):
        if p not in sys.path:
            sys.path.insert(0, p)
    for name in ("roipool3d_cuda", "iou3d_cuda"):
        sys.modules.setdefault(name, types.ModuleType(name))
    import yaml
    _load = yaml.load
    yaml.load = lambda f, Loader=yaml.SafeLoader: _load(f, Loader=Loader)      # lib/config.py:187 predates PyYAML 6
    from lib.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(REFERENCE, "tools/cfgs/default.yaml"))
    yaml.load = _load
    from PIL import Image
    from lib.datasets.kitti_rcnn_dataset import KittiRCNNDataset
    from lib.utils.calibration import Calibration

    root = os.path.join(HERE, "_scene_tmp")
    shutil.rmtree(root, ignore_errors=True)
    base = os.path.join(root, "KITTI", "object", "training")
    for d in ("velodyne", "calib", "image_2", "label_2"):
        os.makedirs(os.path.join(base, d))
    os.makedirs(os.path.join(root, "KITTI", "ImageSets"))
    with open(os.path.join(root, "KITTI", "ImageSets", "val.txt"), "w") as f:
        f.write("".join("%06d\n" % k for k in range(len(FRAMES))))
    for k, fr in enumerate(FRAMES):
        synthetic_scan(fr["n"], fr["seed"], fr["fov"], fr["far"]).tofile(os.path.join(base, "velodyne", "%06d.bin" % k))
        with open(os.path.join(base, "calib", "%06d.txt" % k), "w") as f:
            f.write(KITTI_CALIB_TXT)
        Image.new("RGB", (fr["hw"][1], fr["hw"][0])).save(os.path.join(base, "image_2", "%06d.png" % k))

    out = {"npoints": NPOINTS, "stride": STRIDE, "area_scope": np.asarray(cfg.PC_AREA_SCOPE, np.float64).reshape(-1)}
    try:
        ds = KittiRCNNDataset(root_dir=root, npoints=NPOINTS, split="val", mode="TEST", random_select=True,
                              logger=logging.getLogger("ref_scene"))
        for k, fr in enumerate(FRAMES):
            calib = ds.get_calib(k)
            scan = ds.get_lidar(k)
            rect = calib.lidar_to_rect(scan[:, 0:3])
            img, depth = calib.rect_to_img(rect)
            flag = ds.get_valid_flag(rect, img, depth, ds.get_image_shape(k))
            np.random.seed(1000 + k)
            sample = ds.get_rpn_sample(k)
            out.update({"f%d_M" % k: np.dot(calib.V2C.T, calib.R0.T), "f%d_P2" % k: calib.P2,
                        "f%d_rect" % k: rect[::STRIDE].astype(np.float32), "f%d_img" % k: img[::STRIDE].astype(np.float32),
                        "f%d_depth" % k: depth[::STRIDE].astype(np.float32), "f%d_flag" % k: np.packbits(flag),
                        "f%d_sample_rect" % k: sample["pts_rect"].astype(np.float32),
                        "f%d_sample_feat" % k: sample["pts_features"].astype(np.float32),
                        "f%d_nvalid" % k: int(flag.sum())})
            print("frame %d: %d raw, %d valid, %d far, sample %s" % (k, scan.shape[0], flag.sum(), (rect[flag][:, 2] >= 40).sum(),
                                                                     sample["pts_rect"].shape))
    finally:
        shutil.rmtree(root, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, "scene_ref.npz"), **out)


if __name__ == "__main__":
    main()
