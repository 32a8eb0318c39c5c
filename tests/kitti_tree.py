"""TEST INFRASTRUCTURE: a synthetic KITTI object tree in the on-disk formats the reference's dataset classes read
(lib/datasets/kitti_dataset.py:11-63: velodyne `.bin` float32 x 4, calib `.txt`, `label_2/*.txt`, `image_2/*.png` -- only the image
SIZE is read, :34-39 -- and `ImageSets/<split>.txt`), so that tools/eval_rcnn.py can run unchanged, end to end, where no KITTI data
exists.  Scans come from pointrcnn_amd.kitti_input.synthetic_scan, the calibration is KITTI training frame 000000's."""
import os

import numpy as np

IMG_SHAPES = ((375, 1242), (370, 1224), (374, 1238), (376, 1241))        # the four image sizes that occur in KITTI


def write_tree(data_root, frames, seed0=500, n_scan=40000):
    """<data_root>/KITTI/object/training/{velodyne,calib,label_2,image_2} + <data_root>/KITTI/ImageSets/{train,val,test}.txt for
    the given frame ids; labels start as one DontCare line per frame (write_labels replaces them).  -> training dir"""
    from PIL import Image
    from pointrcnn_amd import kitti_input
    base = os.path.join(str(data_root), "KITTI", "object", "training")
    for sub in ("velodyne", "calib", "label_2", "image_2"):
        os.makedirs(os.path.join(base, sub), exist_ok=True)
    os.makedirs(os.path.join(str(data_root), "KITTI", "ImageSets"), exist_ok=True)
    for k, f in enumerate(frames):
        scan = kitti_input.synthetic_scan(n_scan + 137 * k, seed=seed0 + f, fov_frac=0.6, far_frac=0.1)
        scan.astype(np.float32).tofile(os.path.join(base, "velodyne", "%06d.bin" % f))
        with open(os.path.join(base, "calib", "%06d.txt" % f), "w") as fh:
            fh.write(kitti_input.KITTI_CALIB_TXT)
        h, w = image_shape(f)
        Image.new("RGB", (w, h)).save(os.path.join(base, "image_2", "%06d.png" % f))
        with open(os.path.join(base, "label_2", "%06d.txt" % f), "w") as fh:
            fh.write("DontCare -1 -1 -10 0.00 0.00 1.00 1.00 -1 -1 -1 -1000 -1000 -1000 -10\n")
    for split in ("train", "val", "test"):
        with open(os.path.join(str(data_root), "KITTI", "ImageSets", split + ".txt"), "w") as fh:
            fh.write("".join("%06d\n" % f for f in frames))
    return base


def image_shape(frame):
    return IMG_SHAPES[frame % len(IMG_SHAPES)]


def label_lines(boxes3d, P2, img_shape, cls="Car"):
    """(K,7) [x, y(bottom), z, h, w, l, ry] rect-camera boxes -> KITTI label lines (lib/utils/object3d.py:12-30 field order): type,
    truncation 0, occlusion 0, alpha, the projected 2-D box clipped to the image, h w l, x y z, ry"""
    from pointrcnn_amd import kitti_output
    b = np.asarray(boxes3d, np.float64).reshape(-1, 7)
    if not len(b):
        return []
    ib = kitti_output.corners_to_image_boxes(kitti_output.box_corners(b), P2)
    H, W = img_shape[:2]
    ib[:, 0::2] = np.clip(ib[:, 0::2], 0, W - 1)
    ib[:, 1::2] = np.clip(ib[:, 1::2], 0, H - 1)
    alpha = kitti_output.observation_angle(b)
    return ["%s 0.00 0 %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f" %
            ((cls, alpha[i]) + tuple(ib[i]) + tuple(b[i, 3:6]) + tuple(b[i, 0:3]) + (b[i, 6],)) for i in range(len(b))]


def write_labels(training_dir, frame, boxes3d, P2, cls="Car"):
    lines = label_lines(boxes3d, P2, image_shape(frame), cls)
    with open(os.path.join(training_dir, "label_2", "%06d.txt" % frame), "w") as fh:
        fh.write("".join(ln + "\n" for ln in lines))
    return lines


def read_result_file(path):
    """a KITTI result file (tools/eval_rcnn.py:69-94) -> list of (class name, 15 floats: trunc occl alpha x1 y1 x2 y2 h w l x y z ry score)"""
    out = []
    with open(path) as fh:
        for ln in fh:
            p = ln.split()
            if p:
                out.append((p[0], [float(v) for v in p[1:]]))
    return out
