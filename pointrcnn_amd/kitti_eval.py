"""KITTI object-detection evaluation (AP for bbox / BEV / 3D / AOS) on the MI355X -- SURVEY.md 8(f) rank 2.

Host-side mirror of the reference's evaluator, `tools/kitti_object_eval_python/{evaluate,eval,kitti_common}.py`:
same entry points (`evaluate`, `get_official_eval_result`, `get_coco_eval_result`, `get_label_annos`), same annotation
dictionaries, same result strings / dictionaries.  The reference needs numba (CPU jit) and numba.cuda (the rotated-IoU
kernel), neither of which exists for ROCm; here the two heavy pieces are HIP kernels behind the C ABI
(csrc/kitti_eval.hip):

  * per-frame overlap blocks (2-D box IoU, rotated BEV IoU, 3-D IoU) for ALL frames in one launch -- the reference
    computes dense (sum gt) x (sum dt) matrices over parts of ~75 frames and slices the diagonal blocks out;
  * the greedy gt<->detection matching (`compute_statistics_jit`) for every (frame, score threshold) pair at once:
    3769 frames x 41 thresholds = 154 k independent sequential problems, one thread each.

Everything else (label parsing, the per-object ignore classification, threshold selection, PR / mAP assembly, printing)
is host bookkeeping, written as whole-array numpy over ONE concatenated table of all frames' objects (`_Table`): the
reference walks objects and frames in Python loops (eval.py:29-84, 400-440, 496-544), which dominates its run time once
the IoU and matching are kernels.  The protocol (class aliases, the height / occlusion / truncation limits, the
41-point recall sampling, the running maximum over recall, 11-point interpolation) fixes the numbers; the summation
orders that decide the last bit (frame order for the similarity sums, left-to-right for the 11-point sum) are kept by
construction (`np.cumsum`, which accumulates sequentially, where `np.sum` would add pairwise).  The compute backend is
injected (`backend=`) so that the tests can run the same host code on the CPU oracle; the default backend is the HIP library and nothing else.
"""
import pathlib
import re

import numpy as np


# ----------------------------------------------------------------------------------------------------------------
# label files -> annotation dictionaries (the dictionaries are the reference's interchange format, kitti_common.py:292-346)
# ----------------------------------------------------------------------------------------------------------------
_FIELDS = 15            # type, truncated, occluded, alpha, bbox(4), h w l, x y z, ry [, score]


def get_image_index_str(img_idx):
    return "%06d" % img_idx


def get_label_anno(label_path):
    """one KITTI label / detection txt file -> annotation dict; `dimensions` reordered from the file's (h, w, l) to (l, h, w)
    as kitti_common.py:292-329 delivers them"""
    with open(label_path, "r") as f:
        rows = [ln.split() for ln in f.read().splitlines() if ln.strip()]
    n = len(rows)
    ncol = len(rows[0]) if n else _FIELDS
    names = np.array([r[0] for r in rows]) if n else np.array([])
    num = np.array([r[1:] for r in rows], dtype=np.float64).reshape(n, ncol - 1)
    return {"name": names, "truncated": num[:, 0], "occluded": num[:, 1].astype(np.int64), "alpha": num[:, 2],
            "bbox": num[:, 3:7], "dimensions": num[:, [9, 7, 8]], "location": num[:, 10:13], "rotation_y": num[:, 13],
            "score": num[:, 14] if ncol == _FIELDS + 1 else np.zeros(n)}


def get_label_annos(label_folder, image_ids=None):
    """all (or the listed) frames of a folder, in index order (kitti_common.py:331-346)"""
    folder = pathlib.Path(label_folder)
    if image_ids is None:
        image_ids = sorted(int(p.stem) for p in folder.glob("*.txt") if re.fullmatch(r"\d{6}\.txt", p.name))
    elif not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    return [get_label_anno(folder / (get_image_index_str(i) + ".txt")) for i in image_ids]


def filter_annos_low_score(image_annos, thresh):
    """drop detections scoring below `thresh` (kitti_common.py:215-227)"""
    return [{k: v[np.flatnonzero(np.asarray(a["score"]) >= thresh)] for k, v in a.items()} for a in image_annos]


# ----------------------------------------------------------------------------------------------------------------
# compute backends
# ----------------------------------------------------------------------------------------------------------------
class HipBackend:
    """overlaps / statistics on the GPU through libprcnn_pointops.so (the product path; no CPU fallback)"""

    def __init__(self, device="cuda:0"):
        import torch
        from . import ops
        self.torch, self.ops, self.dev = torch, ops, torch.device(device)

    def _t(self, a, dtype):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(self.dev)

    def overlaps(self, metric, dt, dt_off, gt, gt_off, ov_off):
        out = self.ops.kitti_overlaps(metric, self._t(dt, np.float64), self._t(dt_off, np.int32), self._t(gt, np.float64),
                                      self._t(gt_off, np.int32), self._t(ov_off, np.int64), int(ov_off[-1]))
        return out.cpu().numpy()

    def statistics(self, overlaps, ov_off, gt_datas, gt_off, dt_datas, dt_off, ign_gt, ign_det, dc, dc_off, metric, min_overlap,
                   thresholds, compute_fp, compute_aos):
        t = self._t
        res, matched = self.ops.kitti_statistics(t(overlaps, np.float64), t(ov_off, np.int64), t(gt_datas, np.float64),
                                                 t(gt_off, np.int32), t(dt_datas, np.float64), t(dt_off, np.int32),
                                                 t(ign_gt, np.int32), t(ign_det, np.int32), t(dc, np.float64), t(dc_off, np.int32),
                                                 metric, float(min_overlap), t(thresholds, np.float64), bool(compute_fp),
                                                 bool(compute_aos))
        return res.cpu().numpy(), matched.cpu().numpy()


# ----------------------------------------------------------------------------------------------------------------
# the evaluation protocol, on one concatenated object table
# ----------------------------------------------------------------------------------------------------------------
_CLASS_TO_NAME = {0: "Car", 1: "Pedestrian", 2: "Cyclist", 3: "Van", 4: "Person_sitting"}
_EVAL_CLASSES = ("car", "pedestrian", "cyclist")             # eval.py:30
_NEIGHBOUR_CLASS = {"car": "van", "pedestrian": "person_sitting"}   # evaluated as "ignored", not as misses (eval.py:45-50)
# per difficulty (easy, moderate, hard): minimum 2-D box height [px], maximum occlusion level, maximum truncation (eval.py:31-33)
_MIN_HEIGHT = np.array([40.0, 25.0, 25.0])
_MAX_OCCLUSION = np.array([0, 1, 2])
_MAX_TRUNCATION = np.array([0.15, 0.3, 0.5])
N_SAMPLE_PTS = 41


def _offsets(counts):
    off = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(counts, out=off[1:])
    return off


def _cat(annos, key, width=None, dtype=np.float64):
    parts = [np.asarray(a[key]).reshape(-1, width) if width else np.asarray(a[key]).reshape(-1) for a in annos]
    if not parts:
        return np.zeros((0, width) if width else (0,), dtype)
    return np.concatenate(parts, 0).astype(dtype, copy=False)


class _Table:
    """every object of every frame of one annotation list in flat arrays + frame offsets; built once per eval_class call and
    shared by all (class, difficulty) settings"""

    def __init__(self, annos, is_dt):
        self.count = np.array([len(a["name"]) for a in annos], np.int64)
        self.off = _offsets(self.count)
        names = [np.asarray(a["name"], dtype=str).reshape(-1) for a in annos]
        self.name = np.concatenate(names) if names else np.array([], dtype=str)
        self.lname = np.char.lower(self.name) if self.name.size else self.name
        self.bbox = _cat(annos, "bbox", 4)
        self.alpha = _cat(annos, "alpha")
        if is_dt:
            self.score = _cat(annos, "score")
            self.data = np.concatenate([self.bbox, self.alpha[:, None], self.score[:, None]], 1)     # eval.py:432-435
        else:
            self.occluded = _cat(annos, "occluded")
            self.truncated = _cat(annos, "truncated")
            self.data = np.concatenate([self.bbox, self.alpha[:, None]], 1)                          # eval.py:430-431
        self.frame = np.repeat(np.arange(len(annos)), self.count)


def _classify(gt, dt, current_class, difficulty):
    """The per-object ignore codes of eval.py:29-84 for all frames at once.
    ground truth: 0 = counts, 1 = matched without reward or penalty (neighbour class, or too hard for this difficulty),
    -1 = other class; detections: 0 = counts, 1 = too small, -1 = other class; DontCare regions per frame."""
    cls = _EVAL_CLASSES[current_class]
    same = gt.lname == cls
    neighbour = gt.lname == _NEIGHBOUR_CLASS.get(cls, "\0")
    height = gt.bbox[:, 3] - gt.bbox[:, 1]
    too_hard = (gt.occluded > _MAX_OCCLUSION[difficulty]) | (gt.truncated > _MAX_TRUNCATION[difficulty]) \
        | (height <= _MIN_HEIGHT[difficulty])
    ign_gt = np.full(gt.name.shape, -1, np.int32)
    ign_gt[neighbour | (same & too_hard)] = 1
    ign_gt[same & ~too_hard] = 0
    is_dc = gt.name == "DontCare"
    ign_det = np.where(dt.lname == cls, 0, -1).astype(np.int32)
    ign_det[np.abs(dt.bbox[:, 3] - dt.bbox[:, 1]) < _MIN_HEIGHT[difficulty]] = 1
    n_frames = len(gt.count)
    return dict(ign_gt=ign_gt, ign_det=ign_det, dc=gt.bbox[is_dc],
                dc_off=_offsets(np.bincount(gt.frame[is_dc], minlength=n_frames)),
                total_num_valid_gt=int(np.count_nonzero(ign_gt == 0)))


def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """single-frame form with the reference's return convention (eval.py:29): (num_valid_gt, ignored_gt, ignored_dt, dc_bboxes)"""
    c = _classify(_Table([gt_anno], False), _Table([dt_anno], True), current_class, difficulty)
    return c["total_num_valid_gt"], c["ign_gt"].tolist(), c["ign_det"].tolist(), list(c["dc"])


def get_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    """Score thresholds at which recall crosses the sample levels 0, 1/40, 2/40, ... (eval.py:7-26).
    With the matched scores in descending order, detection i stands for the recall interval [(i+1)/num_gt, (i+2)/num_gt];
    a level takes the first not yet used detection whose interval's far end is at least as close to the level as its near
    end (the last detection always qualifies).  Both distances are monotone in i, so per level the first qualifying
    detection is the first False of one whole-array comparison; <= 41 levels are ever consumed below full recall."""
    s = np.sort(np.asarray(scores, dtype=np.float64))[::-1]
    n = len(s)
    if n == 0:
        return []
    rank = np.arange(n)
    near = (rank + 1) / num_gt
    far = np.where(rank < n - 1, (rank + 2) / num_gt, near)
    step = 1 / (num_sample_pts - 1.0)
    picked, level, start = [], 0, 0
    while start < n:
        passed_over = (far[start:] - level) < (level - near[start:])
        passed_over[-1] = False
        start += int(np.argmin(passed_over))
        picked.append(start)
        start += 1
        level += step                    # accumulated, not t * step: the reference's levels carry this rounding
    return s[picked].tolist()


def _boxes7(annos):
    """(x, y, z, l, h, w, ry) rows (the operand layout of eval.py:372-384)"""
    return np.concatenate([_cat(annos, "location", 3), _cat(annos, "dimensions", 3), _cat(annos, "rotation_y")[:, None]], 1)


def calculate_iou_partly(gt_annos, dt_annos, metric, backend):
    """eval.py:326-397, per-frame blocks only.  Called like the reference's eval_class does -- with (dt_annos, gt_annos) --
    so the FIRST argument indexes the rows.  -> (list of per-frame (n_first, n_second) float64 arrays, flat, ov_off)"""
    if len(gt_annos) != len(dt_annos):
        raise ValueError("frame lists differ in length")
    if metric not in (0, 1, 2):
        raise ValueError("unknown metric")
    n_first = np.array([len(a["name"]) for a in gt_annos], np.int64)
    n_second = np.array([len(a["name"]) for a in dt_annos], np.int64)
    ov_off = _offsets(n_first * n_second)
    rows = (lambda an: _cat(an, "bbox", 4)) if metric == 0 else _boxes7
    flat = backend.overlaps(metric, rows(gt_annos), _offsets(n_first), rows(dt_annos), _offsets(n_second), ov_off)
    blocks = [flat[ov_off[i]:ov_off[i + 1]].reshape(n_first[i], n_second[i]) for i in range(len(n_first))]
    return blocks, flat, ov_off


def _pr_curves(res, compute_aos):
    """(frames, thresholds, 4) per-frame [tp, fp, fn, similarity] -> recall, precision, aos per threshold with the running
    maximum towards higher recall (eval.py:514-544).  cumsum adds the frames in order, one at a time -- the order in which
    the reference accumulates them, which fixes the last bit of the similarity sum."""
    sim = np.where(res[..., 3] != -1, res[..., 3], 0.0)
    tp, fp, fn = (res[..., c].sum(0) for c in range(3))               # integer-valued: any order is exact
    sim = np.cumsum(sim, axis=0)[-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        curves = [tp / (tp + fn), tp / (tp + fp), sim / (tp + fp) if compute_aos else None]
    return [None if c is None else np.maximum.accumulate(c[::-1])[::-1] for c in curves]


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, backend=None):
    """eval.py:443-544.  -> dict(recall, precision, orientation), each [num_class, num_difficulty, num_minoverlap, 41]"""
    backend = backend or HipBackend()
    _, flat, ov_off = calculate_iou_partly(dt_annos, gt_annos, metric, backend)          # rows = detections (:468)
    gt, dt = _Table(gt_annos, False), _Table(dt_annos, True)
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, current_class in enumerate(current_classes):
        for l, difficulty in enumerate(difficultys):
            P = _classify(gt, dt, current_class, difficulty)
            args = (flat, ov_off, gt.data, gt.off, dt.data, dt.off, P["ign_gt"], P["ign_det"], P["dc"], P["dc_off"], metric)
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                # pass 1 (:479-494): scores of the detections that match a counted gt at this overlap
                _, matched = backend.statistics(*args, min_overlap, np.zeros(1), False, False)
                thresholds = np.array(get_thresholds(matched[~np.isnan(matched)], P["total_num_valid_gt"]))
                if len(thresholds) == 0:
                    continue
                # pass 2 (:496-524): tp / fp / fn / similarity per (frame, threshold)
                res, _ = backend.statistics(*args, min_overlap, thresholds, True, compute_aos)
                r, p, a = _pr_curves(res.reshape(len(gt_annos), len(thresholds), 4), compute_aos)
                T = len(thresholds)
                recall[m, l, k, :T], precision[m, l, k, :T] = r, p
                if compute_aos:
                    aos[m, l, k, :T] = a
    return {"recall": recall, "precision": precision, "orientation": aos}


def get_mAP(prec):
    """11-point interpolated AP [%] from the 41 sample points: every fourth point, added left to right (eval.py:547-551)"""
    return np.cumsum(prec[..., ::4], axis=-1)[..., -1] / 11 * 100


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, backend=None):
    """eval.py:563-583 -> (mAP_bbox, mAP_bev, mAP_3d, mAP_aos), each [class, difficulty, min_overlap]"""
    backend = backend or HipBackend()
    difficultys = [0, 1, 2]
    by_metric = [eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps,
                            compute_aos and metric == 0, backend) for metric in (0, 1, 2)]
    maps = [get_mAP(r["precision"]) for r in by_metric]
    return maps[0], maps[1], maps[2], (get_mAP(by_metric[0]["orientation"]) if compute_aos else None)


def do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges, compute_aos, backend=None):
    """eval.py:586-602: overlap_ranges[(start, stop, count), metric, class] -> APs averaged over the overlap grid"""
    lo, hi, num = overlap_ranges
    min_overlaps = np.zeros((10,) + lo.shape)
    for ij in np.ndindex(lo.shape):
        min_overlaps[(slice(None),) + ij] = np.linspace(lo[ij], hi[ij], int(num[ij]))
    return tuple(None if v is None else v.mean(-1)
                 for v in do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, backend))


def print_str(value, *arg, sstream=None):
    """what print() would write for the arguments, as a string (eval.py:554-560)"""
    text = " ".join(str(v) for v in (value,) + arg) + "\n"
    if sstream is not None:
        sstream.truncate(0)
        sstream.seek(0)
        sstream.write(text)
    return text


def _classes_int(current_classes):
    by_name = {v: n for n, v in _CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    return [by_name[c] if isinstance(c, str) else c for c in current_classes]


def _has_alpha(dt_annos):
    """AOS is evaluated when the first non-empty detection frame carries a real alpha (-10 marks "none"; eval.py:633-640)"""
    first = next((a["alpha"] for a in dt_annos if a["alpha"].shape[0] != 0), None)
    return first is not None and bool(first[0] != -10)


def _ap_lines(title, rows):
    """one result block: the title line, then `<label> AP:easy, moderate, hard` per (label, values, number format) row"""
    return title + "\n" + "".join("%s AP:%s\n" % (label, ", ".join(format(v, fmt) for v in vals)) for label, vals, fmt in rows)


def get_official_eval_result(gt_annos, dt_annos, current_classes, backend=None):
    """eval.py:605-675: -> (result string, dict of the nine Car AP numbers)"""
    # [overlap set, metric (bbox, bev, 3d), class]: the official thresholds and the relaxed set
    per_class, per_class_relaxed = np.array([0.7, 0.5, 0.5, 0.7, 0.5]), np.array([0.5, 0.25, 0.25, 0.5, 0.25])
    strict = np.stack([per_class, per_class, per_class])                       # bbox, bev, 3d
    relaxed = np.stack([per_class, per_class_relaxed, per_class_relaxed])      # the 2-D box threshold is not relaxed
    current_classes = _classes_int(current_classes)
    min_overlaps = np.stack([strict, relaxed], axis=0)[:, :, current_classes]
    compute_aos = _has_alpha(dt_annos)
    mAPbbox, mAPbev, mAP3d, mAPaos = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, backend)
    result = ""
    for j, curcls in enumerate(current_classes):
        for i in range(min_overlaps.shape[0]):
            rows = [("bbox", mAPbbox[j, :, i], ".4f"), ("bev ", mAPbev[j, :, i], ".4f"), ("3d  ", mAP3d[j, :, i], ".4f")]
            if compute_aos:
                rows.append(("aos ", mAPaos[j, :, i], ".2f"))
            result += _ap_lines("%s AP@%.2f, %.2f, %.2f:" % ((_CLASS_TO_NAME[curcls],) + tuple(min_overlaps[i, :, j])), rows)
    ret_dict = {"Car_%s_%s" % (mname, dname): table[0, d, 0]
                for mname, table in (("3d", mAP3d), ("bev", mAPbev), ("image", mAPbbox))
                for d, dname in enumerate(("easy", "moderate", "hard"))}
    return result, ret_dict


def get_coco_eval_result(gt_annos, dt_annos, current_classes, backend=None):
    """eval.py:678-740: APs averaged over ten overlap thresholds per class"""
    class_to_range = {0: (0.5, 0.95, 10), 1: (0.25, 0.7, 10), 2: (0.25, 0.7, 10), 3: (0.5, 0.95, 10), 4: (0.25, 0.7, 10)}
    current_classes = _classes_int(current_classes)
    overlap_ranges = np.stack([np.tile(np.array(class_to_range[c])[:, None], (1, 3)) for c in current_classes], axis=-1)
    compute_aos = _has_alpha(dt_annos)
    mAPbbox, mAPbev, mAP3d, mAPaos = do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges, compute_aos, backend)
    result = ""
    for j, curcls in enumerate(current_classes):
        start, stop, count = class_to_range[curcls]
        rows = [("bbox", mAPbbox[j], ".2f"), ("bev ", mAPbev[j], ".2f"), ("3d  ", mAP3d[j], ".2f")]
        if compute_aos:
            rows.append(("aos ", mAPaos[j], ".2f"))
        result += _ap_lines("%s coco AP@%.2f:%.2f:%.2f:" % (_CLASS_TO_NAME[curcls], start, (stop - start) / (count - 1), stop), rows)
    return result


def evaluate(label_path, result_path, label_split_file, current_class=0, coco=False, score_thresh=-1, backend=None):
    """tools/kitti_object_eval_python/evaluate.py:14-28 (what tools/eval_rcnn.py:449,677 calls as kitti_evaluate)"""
    dt_annos = get_label_annos(result_path)
    if score_thresh > 0:
        dt_annos = filter_annos_low_score(dt_annos, score_thresh)
    with open(label_split_file, "r") as f:
        val_image_ids = [int(tok) for tok in f.read().split()]
    gt_annos = get_label_annos(label_path, val_image_ids)
    if coco:
        return get_coco_eval_result(gt_annos, dt_annos, current_class, backend)
    return get_official_eval_result(gt_annos, dt_annos, current_class, backend)
