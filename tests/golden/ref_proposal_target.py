"""Run the REFERENCE's own ProposalTargetLayer.sample_rois_for_rcnn / aug_roi_by_noise_torch / random_aug_box3d
(lib/rpn/proposal_target_layer.py:75-300) and iou3d_utils.boxes_iou3d_gpu on CPU, with its random calls answered from the
counter-based table that oracle/prcnn_oracle.c and csrc/proposal_target.hip draw from.

Only used to GENERATE tests/golden/proposal_target_ref.npz and by tests in the build container (needs /root/reference).  What is
replaced: the compiled-extension call inside boxes_iou3d_gpu (iou3d_cuda.boxes_overlap_bev_gpu -> the reference's own iou3d
sources compiled for the host, oracle/_ref), `torch.cuda.FloatTensor` (-> the CPU type), and the four random entry points the
methods call (np.random.permutation, np.random.rand, torch.randint, torch.rand).  The code that decides which RoIs are sampled,
how the noise loop accepts or retries, and what is returned is the reference's Python, unmodified.

The reference consumes random numbers in a data-dependent order from two global generators; the re-specified draw is a pure
function of (purpose, frame, position):
    stream 10, index = RoI number        keys of the fg candidates; the permutation is their ascending (key, RoI) order
    stream 11, index = slot              fg sampling with replacement (no background candidate in the frame)
    stream 12 / 13, index = position     hard / easy background picks (torch.randint)
    stream 20, index = ((slot * 16 + attempt) * 16 + q)   noise loop: q = 8 keep-the-original decision (u < 0.2), q = 0 range
                                         row, q = 1..3 position shift, q = 4..6 size scale, q = 7 rotation
`_Answers` below maps every random CALL of the reference back to its position by following the same control flow from the
outside (it sees the 1 x 1 IoU every attempt returns, so it knows when the loop moves on to the next RoI)."""
import os
import sys
import types

import numpy as np
import torch

REFERENCE = os.environ.get("PRCNN_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
M32 = 0xFFFFFFFF


def available():
    return os.path.isdir(os.path.join(REFERENCE, "lib", "rpn"))


def mix(x):
    x &= M32
    x ^= x >> 16
    x = (x * 0x7feb352d) & M32
    x ^= x >> 15
    x = (x * 0x846ca68b) & M32
    x ^= x >> 16
    return x


def rand32(seed, stream, frame, i):
    return mix(i ^ mix((frame * 0x9E3779B9 + mix((seed + stream * 0x85EBCA6B) & M32)) & M32))


def u01(r):
    return float(np.float32(r >> 8) * np.float32(1.0 / 16777216.0))


def below(r, n):
    return (r * n) >> 32


def load():
    """-> (cfg, ProposalTargetLayer class, iou3d_utils module) of the reference, importable on CPU"""
    import oracle
    ref = oracle.ref()
    if ref is None:
        raise RuntimeError("oracle/_ref is not built")
    for p in (os.path.join(os.path.dirname(HERE), "compat"), REFERENCE):
        if p not in sys.path:
            sys.path.insert(0, p)
    ic = sys.modules.setdefault("iou3d_cuda", types.ModuleType("iou3d_cuda"))
    sys.modules.setdefault("roipool3d_cuda", types.ModuleType("roipool3d_cuda"))

    def boxes_overlap_bev_gpu(a, b, ans):                    # iou3d.cpp:31-50 through the reference's compiled sources
        ans.copy_(torch.from_numpy(ref.boxes_overlap_bev(a.numpy(), b.numpy())))
        return 1
    ic.boxes_overlap_bev_gpu = boxes_overlap_bev_gpu
    import yaml
    _load = yaml.load
    yaml.load = lambda f, Loader=yaml.SafeLoader: _load(f, Loader=Loader)      # lib/config.py predates PyYAML 6
    from lib.config import cfg, cfg_from_file
    cfg_from_file(os.path.join(REFERENCE, "tools/cfgs/default.yaml"))
    yaml.load = _load
    import lib.utils.iou3d.iou3d_utils as iou3d_utils
    from lib.rpn.proposal_target_layer import ProposalTargetLayer
    return cfg, ProposalTargetLayer, iou3d_utils


class _Answers:
    """answers the reference's random calls for ONE frame from the table; follows the reference's control flow from outside"""

    def __init__(self, seed, frame, fg_list, nhard, neasy, n_fg_slots, roi_per_image, aug_times, pos_thresh):
        self.seed, self.frame, self.fg = seed, frame, list(fg_list)
        self.nhard, self.neasy, self.n_fg_slots, self.R = nhard, neasy, n_fg_slots, roi_per_image
        self.aug_times, self.pos_thresh = aug_times, pos_thresh
        self.stage, self.bg_calls = "sample", 0
        self.slot, self.cnt, self.last_iou, self.q = 0, 0, None, None

    def times(self, slot):
        return self.aug_times if slot < self.n_fg_slots else (1 if self.aug_times > 0 else 0)

    def permutation(self, n):
        assert self.stage == "sample" and n == len(self.fg)
        keys = [(rand32(self.seed, 10, self.frame, i), i) for i in self.fg]
        order = sorted(range(n), key=lambda t: keys[t])
        return np.array(order, dtype=np.int64)

    def np_rand(self, *shape):
        if shape:                                            # np.random.rand(ROI_PER_IMAGE): fg with replacement
            assert self.stage == "sample" and shape == (self.R,)
            n = len(self.fg)
            return np.array([(below(rand32(self.seed, 11, self.frame, t), n) + 0.5) / n for t in range(self.R)])
        # the keep-the-original decision that opens every attempt of the noise loop
        if self.stage == "sample":
            self.stage, self.slot, self.cnt = "aug", 0, 0
        elif self.last_iou is None or self.last_iou >= self.pos_thresh or self.cnt >= self.times(self.slot):
            self.slot, self.cnt = self.slot + 1, 0           # the previous RoI's loop ended
        self.last_iou = None
        self.q = (self.slot * 16 + self.cnt) * 16
        self.cnt += 1
        self.sub = 0
        return u01(rand32(self.seed, 20, self.frame, self.q + 8))

    def randint(self, low=0, high=None, size=None, **kw):
        k = size[0]
        if self.stage == "sample":                           # sample_bg_inds: hard first, then easy
            stream = 12 if (self.bg_calls == 0 and self.nhard > 0) else 13
            self.bg_calls += 1
            n = self.nhard if stream == 12 else self.neasy
            assert high == n
            return torch.tensor([below(rand32(self.seed, stream, self.frame, t), n) for t in range(k)], dtype=torch.int64)
        assert k == 1                                        # random_aug_box3d: the range row
        return torch.tensor([below(rand32(self.seed, 20, self.frame, self.q + 0), high)], dtype=torch.int64)

    def torch_rand(self, *size, **kw):
        n = size[0]
        base = {0: 1, 1: 4, 2: 7}[self.sub]                  # position shift (3), size scale (3), rotation (1)
        self.sub += 1
        return torch.tensor([u01(rand32(self.seed, 20, self.frame, self.q + base + c)) for c in range(n)], dtype=torch.float32)

    def saw_iou(self, value):
        self.last_iou = value


def run_reference(roi_boxes3d, gt_boxes3d, seed, oracle_out, aug_method="multiple"):
    """the reference's sample_rois_for_rcnn, frame by frame -> (batch_rois, batch_gt_of_rois, batch_roi_iou) numpy,
    plus its boxes_iou3d_gpu matrix of frame 0.  oracle_out: the oracle's result on the same input (candidate lists / counts:
    what the answers need to know about the frame before the reference asks)."""
    cfg, PTL, iou3d_utils = load()
    cfg.RCNN.REG_AUG_METHOD = aug_method
    layer = PTL()
    fg_thresh = min(cfg.RCNN.REG_FG_THRESH, cfg.RCNN.CLS_FG_THRESH)
    real_iou = iou3d_utils.boxes_iou3d_gpu
    saved = (np.random.permutation, np.random.rand, torch.randint, torch.rand, torch.cuda.FloatTensor)
    outs, iou0 = [], None
    try:
        torch.cuda.FloatTensor = torch.FloatTensor
        for b in range(roi_boxes3d.shape[0]):
            mo = oracle_out["max_overlaps"][b]
            fg = np.flatnonzero(mo >= np.float32(fg_thresh))
            c = oracle_out["counts"][b]
            ans = _Answers(seed, b, fg, int(c[1]), int(c[2]), int(c[3]), cfg.RCNN.ROI_PER_IMAGE, cfg.RCNN.ROI_FG_AUG_TIMES, fg_thresh)

            def tracked(a, bb, _ans=ans):
                r = real_iou(a, bb)
                if a.shape[0] == 1 and bb.shape[0] == 1:
                    _ans.saw_iou(float(r[0][0]))
                return r
            iou3d_utils.boxes_iou3d_gpu = tracked
            np.random.permutation, np.random.rand, torch.randint, torch.rand = ans.permutation, ans.np_rand, ans.randint, ans.torch_rand
            r_t, g_t = torch.from_numpy(roi_boxes3d[b:b + 1].copy()), torch.from_numpy(gt_boxes3d[b:b + 1].copy())
            if b == 0:
                k = gt_boxes3d.shape[1]
                while k > 0 and gt_boxes3d[0, k - 1].sum() == 0:
                    k -= 1
                iou0 = real_iou(r_t[0], g_t[0, :k, 0:7]).numpy()
            outs.append([t.numpy() for t in layer.sample_rois_for_rcnn(r_t, g_t)])
    finally:
        np.random.permutation, np.random.rand, torch.randint, torch.rand, torch.cuda.FloatTensor = saved
        iou3d_utils.boxes_iou3d_gpu = real_iou
    return [np.concatenate([o[i] for o in outs], 0) for i in range(3)] + [iou0]


def scenes(seed, B=3, M=512, G=12):
    """seeded RoIs / ground truth for the fixtures: per frame a few cars, a crowd of RoIs scattered around each (IoUs from 0 to
    ~0.9: foreground, hard and easy background all populated), the rest random boxes; trailing zero rows pad the ground truth.
    Frame 1 has no background candidate below the thresholds turned off (all RoIs near a car), frame 2 no foreground."""
    sys.path.insert(0, os.path.dirname(HERE))
    from util import rand_boxes3d
    rng = np.random.default_rng(seed)
    pts = rng.uniform([-30, 0, 5], [30, 2, 65], (4000, 3)).astype(np.float32)
    gts, rois = [], []
    for b in range(B):
        ng = 3 + (b * 2) % 5
        g = rand_boxes3d(pts, ng, seed=seed * 10 + b)
        gts.append(np.concatenate([g, np.zeros((G - ng, 7), np.float32)]))
        near = g[rng.integers(0, ng, M)] + (rng.normal(0, 1, (M, 7)) * np.array([0.5, 0.15, 0.5, 0.1, 0.1, 0.25, 0.12])).astype(np.float32)
        far = rand_boxes3d(pts, M, seed=seed * 10 + 5 + b)
        if b == 1:
            r = g[rng.integers(0, ng, M)] + (rng.normal(0, 1, (M, 7)) * np.array([0.05, 0.02, 0.05, 0.02, 0.02, 0.04, 0.02])).astype(np.float32)
        elif b == 2:
            r = far
        else:
            take = rng.random(M) < 0.45
            r = np.where(take[:, None], near, far)
        rois.append(r.astype(np.float32))
    return np.stack(rois), np.stack(gts)


def aug_inputs(seed=3, B=2, R=16, S=32):
    """seeded inputs for the data_augmentation fixture (shared with the tests)"""
    g = torch.Generator().manual_seed(seed)
    pts = torch.randn(B, R, S, 3, generator=g) * 2
    rois = torch.cat([torch.randn(B, R, 3, generator=g) * 10 + torch.tensor([0.0, 1.0, 30.0]), torch.rand(B, R, 3, generator=g) * 2 + 1.5,
                      (torch.rand(B, R, 1, generator=g) - 0.5) * 6.2], 2)
    gts = rois + torch.randn(B, R, 7, generator=g) * 0.2
    return pts, rois, gts


def make_golden(path=None):
    """tests/golden/proposal_target_ref.npz: the reference's own outputs on seeded scenes (needs /root/reference)"""
    import zlib
    import oracle
    cpu = oracle.cpu()
    out = {}
    for seed, method in ((1, "multiple"), (2, "multiple"), (3, "single")):
        roi, gt = scenes(seed)
        o = cpu.proposal_target_sample(roi, gt, seed=40 + seed, aug_method=method, trig_mode=0)
        rr, rg, ri, iou0 = run_reference(roi, gt, 40 + seed, o, aug_method=method)
        tag = "s%d_" % seed
        out[tag + "rois"], out[tag + "gt_of_rois"], out[tag + "roi_iou"], out[tag + "iou0"] = rr, rg, ri, iou0
        out[tag + "crc"] = np.uint32(zlib.crc32(gt.tobytes(), zlib.crc32(roi.tobytes())))
        print("scene", seed, method, "counts", o["counts"].tolist(), "reference == oracle:",
              bool(np.array_equal(rr, o["rois"]) and np.array_equal(ri, o["roi_iou"]) and np.array_equal(rg, o["gt_of_rois"])))
    cfg, PTL, _ = load()
    pts, rois, gts = aug_inputs()
    torch.manual_seed(7)
    p2, r2, g2 = PTL().data_augmentation(pts.clone(), rois.clone(), gts.clone())
    out["aug_pts"], out["aug_rois"], out["aug_gt"] = p2.numpy(), r2.numpy(), g2.numpy()
    np.savez_compressed(path or os.path.join(HERE, "proposal_target_ref.npz"), **out)
    print("wrote proposal_target_ref.npz")


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    make_golden()
