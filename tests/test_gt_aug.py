"""GT-augmentation scene edit (second half; prcnn_gt_aug_edit): the point work of
KittiRCNNDataset.apply_gt_aug_to_one_scene (lib/datasets/kitti_rcnn_dataset.py:484-507).
CPU: the oracle against those lines restated in numpy on flags from the reference's own pts_in_boxes3d_cpu compiled in place
(oracle/_ref).  GPU: the kernel against the oracle, bit for bit, ragged batches and the degenerate shapes."""
import numpy as np
import pytest
import torch

from util import kitti_cloud, rand_boxes3d


def _scene(seed, N, K, P):
    rng = np.random.default_rng(seed)
    pts = kitti_cloud(1, N, seed=seed)[0] if N else np.zeros((0, 3), np.float32)
    inten = rng.uniform(0, 1, size=(N,)).astype(np.float32)
    if K and N:
        boxes = rand_boxes3d(pts, K, seed=seed + 1)
        boxes[:, 3:6] *= rng.uniform(1.0, 4.0, size=(K, 1)).astype(np.float32)        # big enough to swallow a few hundred points
    else:
        boxes = np.zeros((K, 7), np.float32)
    new_pts = rng.normal(size=(P, 3)).astype(np.float32)
    new_int = rng.uniform(0, 1, size=(P,)).astype(np.float32)
    return pts, inten, boxes, new_pts, new_int


def _reference_edit(ref, pts_rect, pts_intensity, boxes, new_pts, new_int):
    """kitti_rcnn_dataset.py:484-507, with the reference's compiled pts_in_boxes3d_cpu"""
    src_pts_flag = np.ones(pts_rect.shape[0], dtype=np.int32)
    for new_gt_box3d in boxes:
        enlarged_box3d = new_gt_box3d.copy()
        enlarged_box3d[3] += 2
        mask = ref.pts_in_boxes3d_cpu(pts_rect, enlarged_box3d.reshape(1, 7))[0] == 1
        src_pts_flag[mask] = 0
    return (np.concatenate((pts_rect[src_pts_flag == 1], new_pts), axis=0),
            np.concatenate((pts_intensity[src_pts_flag == 1], new_int), axis=0), 1 - src_pts_flag)


def test_oracle_matches_reference_edit(cpu, ref):
    for seed, N, K, P in ((0, 20000, 12, 700), (1, 5000, 1, 0), (2, 3000, 0, 50)):
        pts, inten, boxes, new_pts, new_int = _scene(seed, N, K, P)
        want_pts, want_int, want_removed = _reference_edit(ref, pts, inten, boxes, new_pts, new_int)
        for mode in (0, 1):                                   # libm float trig (the reference's host code) / canonical trig
            o_pts, o_int, cnt, removed = cpu.gt_aug_edit(pts[None], inten[None], boxes[None], new_pts[None], new_int[None], trig_mode=mode)
            n = int(cnt[0])
            assert n == want_pts.shape[0]
            assert np.array_equal(o_pts[0, :n], want_pts) and np.array_equal(o_int[0, :n], want_int)
            assert np.array_equal(removed[0], want_removed)
            assert not o_pts[0, n:].any() and not o_int[0, n:].any()
        assert K == 0 or want_removed.sum() > 0


def test_oracle_ragged_counts(cpu):
    B, N, K, P = 3, 4000, 6, 300
    sc = [_scene(10 + b, N, K, P) for b in range(B)]
    stack = [np.stack([s[i] for s in sc]) for i in range(5)]
    num_pts, num_boxes, num_new = np.array([4000, 1234, 0], np.int32), np.array([6, 0, 3], np.int32), np.array([300, 17, 0], np.int32)
    o_pts, o_int, cnt, removed = cpu.gt_aug_edit(*stack, num_pts=num_pts, num_boxes=num_boxes, num_new=num_new)
    for b in range(B):
        n, k, p = int(num_pts[b]), int(num_boxes[b]), int(num_new[b])
        s_pts, s_int, s_cnt, s_rem = cpu.gt_aug_edit(stack[0][b:b + 1, :n], stack[1][b:b + 1, :n], stack[2][b:b + 1, :k],
                                                      stack[3][b:b + 1, :p], stack[4][b:b + 1, :p])
        c = int(s_cnt[0])
        assert int(cnt[b]) == c
        assert np.array_equal(o_pts[b, :c], s_pts[0, :c]) and np.array_equal(o_int[b, :c], s_int[0, :c])
        assert np.array_equal(removed[b, :n], s_rem[0]) and not removed[b, n:].any() and not o_pts[b, c:].any()


@pytest.mark.gpu
def test_gt_aug_edit_matches_oracle(cpu):
    from pointrcnn_amd import ops
    dev = torch.device("cuda:0")
    for seed, B, N, K, P in ((0, 4, 30000, 15, 2000), (1, 2, 1024, 64, 1), (2, 1, 1023, 3, 0), (3, 2, 2049, 0, 5), (4, 1, 0, 2, 9)):
        sc = [_scene(100 * seed + b, N, K, P) for b in range(B)]
        stack = [np.stack([s[i] for s in sc]) for i in range(5)]
        rng = np.random.default_rng(seed)
        for ragged in (False, True):
            counts = [None, None, None]
            if ragged:
                counts = [rng.integers(0, n + 1, size=(B,)).astype(np.int32) for n in (N, K, P)]
            want = cpu.gt_aug_edit(*stack, num_pts=counts[0], num_boxes=counts[1], num_new=counts[2])
            t = [torch.from_numpy(a).to(dev) for a in stack]
            c = [None if a is None else torch.from_numpy(a).to(dev) for a in counts]
            got = ops.gt_aug_edit(*t, num_pts=c[0], num_boxes=c[1], num_new=c[2], want_removed=True)
            for g, w in zip(got, want):
                assert np.array_equal(g.cpu().numpy(), w)
            if K and N and not ragged:
                assert int(want[3].sum()) > 0
    # without intensities
    t = [torch.from_numpy(a).to(dev) for a in stack]
    o_pts, o_int, cnt = ops.gt_aug_edit(t[0], None, t[2], t[3], None)
    assert o_int is None and np.array_equal(o_pts.cpu().numpy(), cpu.gt_aug_edit(*stack)[0])
    with pytest.raises(Exception):
        ops.gt_aug_edit(t[0], None, torch.zeros((t[0].shape[0], 65, 7), device=dev), t[3], None)      # more than 64 boxes


@pytest.mark.gpu
def test_gt_aug_edit_scene_mirror(cpu):
    from pointrcnn_amd import kitti_input
    pts, inten, boxes, new_pts, new_int = _scene(7, 18000, 9, 600)
    halves = [new_pts[:250], new_pts[250:]], [new_int[:250], new_int[250:]]
    got_pts, got_int = kitti_input.gt_aug_edit_scene(pts, inten, boxes, *halves)
    o_pts, o_int, cnt, _ = cpu.gt_aug_edit(pts[None], inten[None], boxes[None], new_pts[None], new_int[None])
    assert np.array_equal(got_pts, o_pts[0, :int(cnt[0])]) and np.array_equal(got_int, o_int[0, :int(cnt[0])])
