"""KITTI input side of the RPN, on the device.

Host-side mirror of what the reference's dataset class does for one inference sample
(lib/datasets/kitti_rcnn_dataset.py:246-310 get_rpn_sample, lib/datasets/kitti_dataset.py:34-48,
lib/utils/calibration.py:5-70): parse the calibration text, read the velodyne ``.bin``, and hand the raw scans of a whole
batch to ``prcnn_scene_prepare`` (csrc/scene.hip), which does lidar->rect, the image / PC_AREA_SCOPE crop and the
``npoints`` sampling for all frames in two launches.  Same names and argument meaning as the reference where a
counterpart exists; there is no CPU fallback -- the transform runs in the HIP library or not at all.
"""
import os

import numpy as np
import torch

from . import ops

PC_AREA_SCOPE = ((-40.0, 40.0), (-1.0, 3.0), (0.0, 70.4))        # tools/cfgs/default.yaml:18


# the calibration of KITTI training frame 000000 (public dataset values), as the text file the reference parses;
# used by the synthetic-scan generator below (bench.py --input raw, tests)
KITTI_CALIB_TXT = """P0: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 0.000000000000e+00 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00
P1: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 -3.875744000000e+02 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 0.000000000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 0.000000000000e+00
P2: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 4.485728000000e+01 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 2.163791000000e-01 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 2.745884000000e-03
P3: 7.215377000000e+02 0.000000000000e+00 6.095593000000e+02 -3.395242000000e+02 0.000000000000e+00 7.215377000000e+02 1.728540000000e+02 2.199936000000e+00 0.000000000000e+00 0.000000000000e+00 1.000000000000e+00 2.729905000000e-03
R0_rect: 9.999239000000e-01 9.837760000000e-03 -7.445048000000e-03 -9.869795000000e-03 9.999421000000e-01 -4.278459000000e-03 7.402527000000e-03 4.351614000000e-03 9.999631000000e-01
Tr_velo_to_cam: 7.533745000000e-03 -9.999714000000e-01 -6.166020000000e-04 -4.069766000000e-03 1.480249000000e-02 7.280733000000e-04 -9.998902000000e-01 -7.631618000000e-02 9.998621000000e-01 7.523790000000e-03 1.480755000000e-02 -2.717806000000e-01
Tr_imu_to_velo: 9.999976000000e-01 7.553071000000e-04 -2.035826000000e-03 -8.086759000000e-01 -7.854027000000e-04 9.998898000000e-01 -1.482298000000e-02 3.195559000000e-01 2.024406000000e-03 1.482454000000e-02 9.998881000000e-01 -7.997231000000e-01
"""


def synthetic_scan(n, seed=0, fov_frac=0.5, far_frac=0.25):
    """(n,4) fp32 velodyne-frame scan [x fwd, y left, z up, intensity]: fov_frac of the points inside the camera frustum /
    PC_AREA_SCOPE region (far_frac of those beyond 40 m), the rest all around the sensor (behind, beside, above)."""
    r = np.random.default_rng(seed)
    nin = int(n * fov_frac)
    nfar = int(nin * far_frac)
    depth = np.concatenate([r.uniform(2.0, 40.0, nin - nfar), r.uniform(40.0, 72.0, nfar)])
    lat = r.uniform(-0.75, 0.75, nin) * depth               # roughly the +-40 deg horizontal field of view
    up = r.uniform(-2.2, 1.2, nin)
    inside = np.stack([depth, lat, up], 1)
    outside = np.stack([r.uniform(-80, 80, n - nin), r.uniform(-80, 80, n - nin), r.uniform(-3, 3, n - nin)], 1)
    pts = np.concatenate([inside, outside])
    out = np.concatenate([pts, r.uniform(0, 1, (n, 1))], 1).astype(np.float32)
    return out[r.permutation(n)]


def get_calib_from_file(calib_file):
    """lib/utils/calibration.py:5-21: lines 2..5 of a KITTI calib txt = P2, P3, R0_rect, Tr_velo_to_cam (fp32)"""
    with open(calib_file) as f:
        return get_calib_from_lines(f.readlines())


def get_calib_from_lines(lines):

    def row(k, shape):
        return np.array(lines[k].strip().split(" ")[1:], dtype=np.float32).reshape(shape)
    return {"P2": row(2, (3, 4)), "P3": row(3, (3, 4)), "R0": row(4, (3, 3)), "Tr_velo2cam": row(5, (3, 4))}


class Calibration:
    """lib/utils/calibration.py:24-42 (the members the input path uses)"""

    def __init__(self, calib_file):
        calib = get_calib_from_file(calib_file) if isinstance(calib_file, str) else calib_file
        self.P2 = np.asarray(calib["P2"], np.float32).reshape(3, 4)
        self.R0 = np.asarray(calib["R0"], np.float32).reshape(3, 3)
        self.V2C = np.asarray(calib["Tr_velo2cam"], np.float32).reshape(3, 4)

    @classmethod
    def from_text(cls, text):
        """the same parse from the file's content"""
        return cls(get_calib_from_lines(text.split("\n")))

    def lidar_to_rect_matrix(self):
        """(4,3) fp32 M with pts_rect = [pts_lidar, 1] . M -- formed exactly as calibration.py:57 forms it"""
        return np.dot(self.V2C.T, self.R0.T)

    def packed(self):
        """(24,) fp32 row of prcnn_scene_prepare's calib argument"""
        return np.concatenate([self.lidar_to_rect_matrix().reshape(-1), self.P2.reshape(-1)]).astype(np.float32)


def get_lidar(lidar_file):
    """lib/datasets/kitti_dataset.py:40-43"""
    assert os.path.exists(lidar_file)
    return np.fromfile(lidar_file, dtype=np.float32).reshape(-1, 4)


class ScenePreparer:
    """Batch version of get_rpn_sample's inference branch.  ``npoints`` / ``random_select`` as in KittiRCNNDataset
    (kitti_rcnn_dataset.py:13); area_scope = cfg.PC_AREA_SCOPE or None for cfg.PC_REDUCE_BY_RANGE = False."""

    def __init__(self, npoints=16384, area_scope=PC_AREA_SCOPE, device="cuda"):
        self.npoints = npoints
        self.scope = None if area_scope is None else [float(v) for ax in area_scope for v in ax]
        self.device = torch.device(device)

    def pack(self, scans, calibs, img_shapes, pin=True):
        """host side of one batch: scans = list of (Ni,4) fp32 arrays, calibs = list of Calibration, img_shapes = list of
        (H, W[, 3]).  Returns pinned host tensors ready for one asynchronous copy each."""
        sizes = [int(s.shape[0]) for s in scans]
        off = np.zeros(len(scans) + 1, np.int64)
        np.cumsum(sizes, out=off[1:])
        raw = torch.empty((int(off[-1]), 4), dtype=torch.float32, pin_memory=pin and torch.cuda.is_available())
        for s, a, b in zip(scans, off[:-1], off[1:]):
            raw[a:b] = torch.from_numpy(np.ascontiguousarray(s, np.float32))
        calib = torch.from_numpy(np.stack([c.packed() for c in calibs]))
        hw = torch.tensor([[int(s[0]), int(s[1])] for s in img_shapes], dtype=torch.int32)
        return {"raw": raw, "offsets": torch.from_numpy(off), "calib": calib, "img_hw": hw, "max_points": max(sizes) if sizes else 0}

    def __call__(self, packed, seed=0):
        """device side: H2D of the packed batch on the current stream + prcnn_scene_prepare.
        -> dict(pts_input (B,npoints,3), pts_rect (same tensor), pts_features (B,npoints,1), src, nvalid, status)"""
        dev = self.device
        raw, off, calib, hw = (packed[k].to(dev, non_blocking=True) for k in ("raw", "offsets", "calib", "img_hw"))
        xyz, inten, src, nvalid, status = ops.scene_prepare(raw, off, packed["max_points"], calib, hw, self.scope, self.npoints, seed)
        return {"pts_input": xyz, "pts_rect": xyz, "pts_features": inten.unsqueeze(-1), "src": src, "nvalid": nvalid, "status": status}


def gt_aug_edit_scene(pts_rect, pts_intensity, accepted_boxes3d, new_pts_list, new_intensity_list, device="cuda"):
    """The point work of KittiRCNNDataset.apply_gt_aug_to_one_scene (lib/datasets/kitti_rcnn_dataset.py:484-507) for one
    scene, on the device: drop every scene point inside an accepted object's box (h + 2, :484-489), keep the rest in order and
    append the pasted objects' points (:501-507).  numpy in, numpy out (pts_rect (n', 3), pts_intensity (n',)) -- what the
    reference's function returns as its second and third value.  Batches of scenes: ops.gt_aug_edit."""
    pts = torch.as_tensor(np.ascontiguousarray(pts_rect, np.float32), device=device)[None]
    inten = torch.as_tensor(np.ascontiguousarray(pts_intensity, np.float32), device=device)[None]
    boxes = torch.as_tensor(np.ascontiguousarray(accepted_boxes3d, np.float32).reshape(-1, 7), device=device)[None]
    new_pts = np.concatenate(new_pts_list, axis=0).astype(np.float32) if len(new_pts_list) else np.zeros((0, 3), np.float32)
    new_int = np.concatenate(new_intensity_list, axis=0).astype(np.float32) if len(new_intensity_list) else np.zeros((0,), np.float32)
    out_pts, out_int, count = ops.gt_aug_edit(pts, inten, boxes, torch.as_tensor(new_pts, device=device)[None],
                                              torch.as_tensor(new_int, device=device)[None])
    n = int(count[0])
    return out_pts[0, :n].cpu().numpy(), out_int[0, :n].cpu().numpy()
