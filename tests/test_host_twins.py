"""Host twins of the reference's two CPU entry points (prcnn_host_pts_in_boxes3d / prcnn_host_roipool3d):
bit-identical to the reference's own roipool3d.cpp compiled for the host (golden fixture + live oracle/_ref), reachable through
the drop-in `roipool3d_cuda.*_cpu` functions with the reference's calling convention, and usable inside FORKED DataLoader workers
(where the reference calls them: kitti_rcnn_dataset.py:487,843) without any device."""
import os
import sys

import numpy as np
import torch

from util import GOLDEN, enlarge, kitti_cloud, rand_boxes3d

import pointrcnn_amd

pointrcnn_amd.install()
import roipool3d_cuda  # noqa: E402


def _pool(pts, boxes, feat, S):
    M, C = boxes.shape[0], feat.shape[1]
    pp = torch.zeros((M, S, 3))
    pf = torch.zeros((M, S, C))
    pe = torch.zeros(M, dtype=torch.int64)
    assert roipool3d_cuda.roipool3d_cpu(torch.from_numpy(pts), torch.from_numpy(boxes), torch.from_numpy(feat), pp, pf, pe) == 1
    return pp.numpy(), pf.numpy(), pe.numpy()


def _flags(pts, boxes):
    f = torch.zeros((boxes.shape[0], pts.shape[0]), dtype=torch.int64)
    assert roipool3d_cuda.pts_in_boxes3d_cpu(f, torch.from_numpy(pts), torch.from_numpy(boxes)) == 1
    return f.numpy()


def test_host_twins_match_reference_golden():
    g = np.load(os.path.join(GOLDEN, "roipool3d_ref.npz"))
    pts, boxes, feat, S = g["xyz"][0], g["boxes"][0], g["feat"][0], int(g["S"])
    pp, pf, pe = _pool(pts, boxes, feat, S)
    assert np.array_equal(pp, g["pooled_pts"]) and np.array_equal(pf, g["pooled_feat"]) and np.array_equal(pe, g["empty"])
    assert np.array_equal(_flags(pts, boxes), g["flags"].astype(np.int64))
    assert pe[-1] == 1 and not pp[-1].any()             # the empty box: flagged, rows left as the caller initialised them


def test_host_twins_match_reference_live(ref):
    """larger, denser case against the reference's own code compiled in place (oracle/_ref)"""
    pts = kitti_cloud(1, 16384, seed=5)[0]
    boxes = enlarge(rand_boxes3d(pts, 100, seed=6), 1.0)
    feat = np.random.default_rng(7).normal(size=(16384, 16)).astype(np.float32)
    for S in (8, 512):
        pp, pf, pe = _pool(pts, boxes, feat, S)
        rp, rf, re = ref.roipool3d_cpu(pts, boxes, feat, S)
        assert np.array_equal(pp, rp) and np.array_equal(pf, rf) and np.array_equal(pe, re)
    assert np.array_equal(_flags(pts, boxes), ref.pts_in_boxes3d_cpu(pts, boxes))


def test_reference_wrapper_calling_convention():
    """roipool3d_utils.py:31-75: uninitialised LongTensor flags / zeroed outputs, float().contiguous() inputs; errors -> RuntimeError"""
    pts = kitti_cloud(1, 2000, seed=1)[0]
    boxes = rand_boxes3d(pts, 5, seed=2)
    flag = torch.LongTensor(torch.Size((5, 2000)))      # uninitialised, as the reference allocates it
    roipool3d_cuda.pts_in_boxes3d_cpu(flag, torch.from_numpy(pts), torch.from_numpy(boxes))
    assert set(np.unique(flag.numpy()).tolist()) <= {0, 1}
    import pytest
    with pytest.raises(RuntimeError):
        roipool3d_cuda.pts_in_boxes3d_cpu(flag, torch.from_numpy(pts).t(), torch.from_numpy(boxes))     # non-contiguous
    with pytest.raises(RuntimeError):
        roipool3d_cuda.pts_in_boxes3d_cpu(flag.int(), torch.from_numpy(pts), torch.from_numpy(boxes))   # wrong dtype


class _BoxDataset(torch.utils.data.Dataset):
    """what KittiRCNNDataset.__getitem__ does with these ops, reduced to the calls"""

    def __len__(self):
        return 6

    def __getitem__(self, i):
        pts = kitti_cloud(1, 4096, seed=100 + i)[0]
        boxes = enlarge(rand_boxes3d(pts, 7, seed=200 + i), 1.0)
        feat = np.random.default_rng(300 + i).normal(size=(4096, 4)).astype(np.float32)
        pp, pf, pe = _pool(pts, boxes, feat, 32)
        return torch.from_numpy(pp), torch.from_numpy(pf), torch.from_numpy(pe), torch.from_numpy(_flags(pts, boxes))


def test_cpu_entry_points_work_in_forked_dataloader_workers(cpu):
    loader = torch.utils.data.DataLoader(_BoxDataset(), batch_size=2, num_workers=2, multiprocessing_context="fork")
    got = [b for b in loader]
    assert len(got) == 3
    for bi, (pp, pf, pe, fl) in enumerate(got):
        for j in range(2):
            i = bi * 2 + j
            pts = kitti_cloud(1, 4096, seed=100 + i)[0]
            boxes = enlarge(rand_boxes3d(pts, 7, seed=200 + i), 1.0)
            want = cpu.pts_in_boxes3d(pts, boxes, trig_mode=0)
            assert np.array_equal(fl[j].numpy(), want)
            feat = np.random.default_rng(300 + i).normal(size=(4096, 4)).astype(np.float32)
            wp, we = cpu.roipool3d(pts[None], boxes[None], feat[None], 32, trig_mode=0)
            assert np.array_equal(pe[j].numpy(), we[0].astype(np.int64))
            keep = we[0] == 0
            assert np.array_equal(pp[j].numpy()[keep], wp[0][keep][:, :, :3]) and np.array_equal(pf[j].numpy()[keep], wp[0][keep][:, :, 3:])
