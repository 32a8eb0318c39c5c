"""seeded synthetic inputs shared by the tests (shapes follow SURVEY.md 8(d) configs)"""
import numpy as np

GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def unit_cloud(B, N, seed=0):
    return np.random.default_rng(seed).random((B, N, 3), dtype=np.float32)


def kitti_cloud(B, N, seed=100):
    out = np.empty((B, N, 3), np.float32)
    for b in range(B):
        r = np.random.default_rng(seed + b)
        out[b] = np.stack([r.uniform(-40, 40, N), r.uniform(-1, 3, N), r.uniform(0, 70.4, N)], 1)
    return out


def rand_bev(n, spread=8.0, seed=0):
    r = np.random.default_rng(seed)
    cx, cy = r.uniform(-spread, spread, n), r.uniform(0, 2 * spread, n)
    w, l, a = r.uniform(1.2, 2.2, n), r.uniform(3.0, 5.0, n), r.uniform(-np.pi, np.pi, n)
    return np.stack([cx - w / 2, cy - l / 2, cx + w / 2, cy + l / 2, a], 1).astype(np.float32)


def rand_boxes3d(pts, M, seed=0, jitter=0.5):
    """(M,7) [x,y(bottom),z,h,w,l,ry] boxes centred near random points of pts (N,3)"""
    r = np.random.default_rng(seed)
    ctr = pts[r.integers(0, pts.shape[0], M)] + r.normal(0, jitter, (M, 3))
    h, w, l = r.uniform(1.4, 1.8, (M, 1)), r.uniform(1.5, 1.8, (M, 1)), r.uniform(3.5, 4.5, (M, 1))
    ry = r.uniform(-np.pi, np.pi, (M, 1))
    return np.concatenate([ctr[:, :1], ctr[:, 1:2] + h / 2, ctr[:, 2:3], h, w, l, ry], 1).astype(np.float32)


def enlarge(boxes, e):
    """lib/utils/kitti_utils.py:150-160 enlarge_box3d"""
    b = boxes.copy()
    b[..., 3:6] += 2 * e
    b[..., 1] += e
    return b


# oracle trig_mode the HIP kernels of roipool3d / iou3d / NMS / labels / RoI sampling implement (2 = the reference's host libm,
# restated bit for bit; see oracle/prcnn_oracle.c header)
KERNEL_TRIG = 2


def mlp_tol(ref):
    """absolute tolerance for the fp32-MFMA MLP against the double-accumulated oracle (1e-5 relative to scale)"""
    return 1e-5 * max(1.0, float(np.abs(ref).max()))


ANCHOR = np.array([1.52563191462, 1.62856739989, 3.88311640418], np.float32)    # default.yaml:19 CLS_MEAN_SIZE


def encode_rpn_reg(pts, boxes, rng, loc_scope=3.0, loc_bin_size=0.5, num_head_bin=12, noise=0.3, peak=6.0):
    """Synthetic RPN regression rows (n, 76) whose bin-based decode (bbox_transform.py:24-121, LOC_XZ_FINE) lands on
    `boxes` (n,7 with y = box centre): the inverse of the decode, bins get a `peak` logit over N(0, noise) clutter."""
    n = pts.shape[0]
    nb = int(loc_scope / loc_bin_size) * 2
    reg = rng.normal(0, noise, (n, 4 * nb + 1 + 2 * num_head_bin + 3)).astype(np.float32)
    rows = np.arange(n)
    for col, axis in ((0, 0), (1, 2)):                       # x bins, z bins
        d = np.clip(boxes[:, axis] - pts[:, axis], -loc_scope + 1e-3, loc_scope - 1e-3) + loc_scope
        b = np.floor(d / loc_bin_size).astype(np.int64)
        reg[rows, col * nb + b] += peak
        reg[rows, (2 + col) * nb + b] = (d - (b * loc_bin_size + loc_bin_size / 2)) / loc_bin_size
    reg[:, 4 * nb] = boxes[:, 1] - pts[:, 1]
    apc = 2 * np.pi / num_head_bin
    sh = (boxes[:, 6] + apc / 2) % (2 * np.pi)
    hb = np.floor(sh / apc).astype(np.int64)
    reg[rows, 4 * nb + 1 + hb] += peak
    reg[rows, 4 * nb + 1 + num_head_bin + hb] = (sh - (hb * apc + apc / 2)) / (apc / 2)
    reg[:, -3:] = (boxes[:, 3:6] - ANCHOR) / ANCHOR
    return reg


def rpn_like_scene(B, N, seed=0, nobj=24, fg_frac=0.4, z_max=70.4):
    """What the RPN heads emit on a driving scene, synthetically: xyz (B,N,3), raw scores (B,N), reg (B,N,76).
    fg_frac of the points sit on `nobj` cars and vote (with noise) for their car's box -> tight clusters of heavily
    overlapping proposals with high scores, which is what makes the NMS do real work; the rest is background clutter."""
    xyz = np.empty((B, N, 3), np.float32)
    scores = np.empty((B, N), np.float32)
    reg = np.empty((B, N, 76), np.float32)
    for b in range(B):
        r = np.random.default_rng(seed * 1000 + b)
        nfg = int(N * fg_frac)
        obj = np.stack([r.uniform(-35, 35, nobj), r.uniform(0.8, 1.2, nobj), r.uniform(4, z_max - 4, nobj),
                        r.uniform(1.4, 1.7, nobj), r.uniform(1.5, 1.8, nobj), r.uniform(3.4, 4.4, nobj),
                        r.uniform(-np.pi, np.pi, nobj)], 1)
        own = r.integers(0, nobj, nfg)
        pf = obj[own, :3] + r.normal(0, 1, (nfg, 3)) * np.array([0.8, 0.4, 1.2])
        pb = np.stack([r.uniform(-40, 40, N - nfg), r.uniform(-1, 3, N - nfg), r.uniform(0.1, z_max, N - nfg)], 1)
        tgt_f = obj[own] + r.normal(0, 1, (nfg, 7)) * np.array([0.08, 0.03, 0.12, 0.03, 0.03, 0.06, 0.03])
        tgt_b = np.concatenate([pb + r.normal(0, 1.0, (N - nfg, 3)), r.uniform(1.0, 2.0, (N - nfg, 1)),
                                r.uniform(1.2, 2.0, (N - nfg, 1)), r.uniform(3.0, 5.0, (N - nfg, 1)),
                                r.uniform(-np.pi, np.pi, (N - nfg, 1))], 1)
        pts = np.concatenate([pf, pb]).astype(np.float32)
        tgt = np.concatenate([tgt_f, tgt_b])
        perm = r.permutation(N)
        xyz[b] = pts[perm]
        reg[b] = encode_rpn_reg(pts.astype(np.float64), tgt, r)[perm]
        scores[b] = np.concatenate([r.normal(2.5, 1.0, nfg), r.normal(-3.0, 1.0, N - nfg)]).astype(np.float32)[perm]
    return xyz, scores, reg


def kitti_annos(nframes, seed=0, with_score=False, gt=None):
    """Synthetic KITTI annotation dicts (tools/kitti_object_eval_python/kitti_common.py:get_label_anno layout:
    dimensions already reordered to l,h,w).  with_score: detections = jittered copies of `gt` objects + false positives."""
    r = np.random.default_rng(seed)
    names = np.array(["Car", "Car", "Car", "Van", "Pedestrian", "Cyclist", "DontCare", "Person_sitting"])
    annos = []
    for f in range(nframes):
        if not with_score:
            n = int(r.integers(0, 9))
            nm = names[r.integers(0, len(names), n)]
            x1, y1 = r.uniform(0, 1100, n), r.uniform(100, 250, n)
            hgt = r.choice([20.0, 30.0, 45.0, 60.0, 80.0, 140.0], n)
            bbox = np.stack([x1, y1, x1 + hgt * r.uniform(1.0, 2.5, n), y1 + hgt], 1)
            loc = np.stack([r.uniform(-20, 20, n), r.uniform(1.2, 2.0, n), r.uniform(5, 60, n)], 1)
            dims = np.stack([r.uniform(3.2, 4.6, n), r.uniform(1.4, 1.9, n), r.uniform(1.5, 1.9, n)], 1)     # l, h, w
            a = dict(name=nm, truncated=r.choice([0.0, 0.0, 0.1, 0.2, 0.4, 0.6], n), occluded=r.choice([0, 0, 0, 1, 1, 2, 3], n),
                     alpha=r.uniform(-3.1, 3.1, n), bbox=bbox, dimensions=dims, location=loc,
                     rotation_y=r.uniform(-3.1, 3.1, n), score=np.zeros(n))
        else:
            g = gt[f]
            keep = [i for i in range(len(g["name"])) if g["name"][i] != "DontCare" and r.random() < 0.9]
            nfp = int(r.integers(0, 4))
            n = len(keep) + nfp
            jit = lambda s, m: r.normal(0, s, m)        # noqa: E731
            nm = np.concatenate([g["name"][keep], names[r.integers(0, 6, nfp)]]) if n else np.array([], dtype="<U16")
            nm = np.where(nm == "Van", "Car", nm)
            bbox = np.concatenate([g["bbox"][keep] + jit(2.5, (len(keep), 4)),
                                   np.stack([r.uniform(0, 1100, nfp), r.uniform(100, 250, nfp), r.uniform(0, 1100, nfp) + 60,
                                             r.uniform(100, 250, nfp) + 50], 1).reshape(nfp, 4)])
            loc = np.concatenate([g["location"][keep] + jit(0.07, (len(keep), 3)),
                                  np.stack([r.uniform(-20, 20, nfp), r.uniform(1.2, 2.0, nfp), r.uniform(5, 60, nfp)], 1).reshape(nfp, 3)])
            dims = np.concatenate([g["dimensions"][keep] * (1 + jit(0.025, (len(keep), 3))),
                                   np.stack([r.uniform(3.2, 4.6, nfp), r.uniform(1.4, 1.9, nfp), r.uniform(1.5, 1.9, nfp)], 1).reshape(nfp, 3)])
            ry = np.concatenate([g["rotation_y"][keep] + jit(0.04, len(keep)), r.uniform(-3.1, 3.1, nfp)])
            al = np.concatenate([g["alpha"][keep] + jit(0.1, len(keep)), r.uniform(-3.1, 3.1, nfp)])
            a = dict(name=nm, truncated=np.zeros(n), occluded=np.zeros(n, np.int64), alpha=al, bbox=bbox, dimensions=dims,
                     location=loc, rotation_y=ry, score=r.uniform(0.05, 1.0, n))
        annos.append(a)
    return annos


# ---- RPN input builder ------------------------------------------------------------------
from pointrcnn_amd.kitti_input import KITTI_CALIB_TXT, synthetic_scan          # noqa: E402,F401  (shared with bench.py --input raw)


def scene_invariants(out_rect, out_int, valid_rect, valid_int, npoints):
    """What kitti_rcnn_dataset.py:285-306 guarantees about a sample, whatever the random stream: checks it and returns
    the multiplicity of every valid point in the sample.  valid_rect (n,3) / valid_int (n) = the frame's valid points and
    their (intensity - 0.5) in scan order; out_* = the sample."""
    assert out_rect.shape == (npoints, 3) and out_int.shape == (npoints,)
    n = valid_rect.shape[0]
    rows = np.concatenate([valid_rect, valid_int[:, None]], 1).astype(np.float32)
    key = {r.tobytes(): k for k, r in enumerate(rows)}
    assert len(key) == n, "synthetic scan has duplicate points"
    mult = np.zeros(n, np.int64)
    for r in np.concatenate([out_rect, out_int[:, None]], 1).astype(np.float32):
        mult[key[r.tobytes()]] += 1                               # KeyError = a row that is not a valid point of the frame
    far = ~(valid_rect[:, 2] < 40.0)
    if n > npoints:
        assert mult.max() == 1 and mult.sum() == npoints          # a draw without replacement ...
        assert (mult[far] == 1).all()                              # ... that keeps every far point
    else:
        assert mult.min() >= 1 and mult.max() <= 2 and mult.sum() == npoints    # everything, topped up without replacement
    return mult


def assert_same_result(a, b, mlp_mode, what=""):
    """a, b: outputs (torch tensors) of two code paths that compute the same function.  "f32": the kernel variants perform the
    same fp32 operations in the same order -- the outputs are the same BITS.  "split6": the two paths may send a layer to
    different kernel families (an fp32 register-resident chain on one side, split-bf16 layer kernels on the other), so the bar
    is the parity contract of the MLP outputs: 1e-5 of the output scale."""
    import torch
    if mlp_mode == "f32" or not a.is_floating_point():
        assert torch.equal(a, b), what
        return
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= 1e-5 * scale, (what, err, scale)
