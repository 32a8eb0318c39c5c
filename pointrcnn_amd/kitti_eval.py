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

Everything else (label parsing, `clean_data`, threshold selection, PR / mAP assembly, printing) is cheap bookkeeping
and stays in numpy, restated from eval.py line by line.  The compute backend is injected (`backend=`) so that the tests
can run the same host code on the CPU oracle; the default backend is the HIP library and nothing else.
"""
import io as sysio
import pathlib
import re

import numpy as np


# ----------------------------------------------------------------------------------------------------------------
# kitti_common.py subset
# ----------------------------------------------------------------------------------------------------------------
def get_image_index_str(img_idx):
    return "{:06d}".format(img_idx)


def get_label_anno(label_path):
    """kitti_common.py:292-329 -- one label / detection txt file -> annotation dict (dimensions reordered to l,h,w)"""
    with open(label_path, "r") as f:
        content = [line.strip().split(" ") for line in f.readlines()]
    a = {}
    a["name"] = np.array([x[0] for x in content])
    a["truncated"] = np.array([float(x[1]) for x in content])
    a["occluded"] = np.array([int(x[2]) for x in content])
    a["alpha"] = np.array([float(x[3]) for x in content])
    a["bbox"] = np.array([[float(v) for v in x[4:8]] for x in content]).reshape(-1, 4)
    a["dimensions"] = np.array([[float(v) for v in x[8:11]] for x in content]).reshape(-1, 3)[:, [2, 0, 1]]
    a["location"] = np.array([[float(v) for v in x[11:14]] for x in content]).reshape(-1, 3)
    a["rotation_y"] = np.array([float(x[14]) for x in content]).reshape(-1)
    if len(content) != 0 and len(content[0]) == 16:
        a["score"] = np.array([float(x[15]) for x in content])
    else:
        a["score"] = np.zeros([len(a["bbox"])])
    return a


def get_label_annos(label_folder, image_ids=None):
    """kitti_common.py:331-346"""
    if image_ids is None:
        prog = re.compile(r"^\d{6}.txt$")
        image_ids = sorted(int(p.stem) for p in pathlib.Path(label_folder).glob("*.txt") if prog.match(p.name))
    if not isinstance(image_ids, list):
        image_ids = list(range(image_ids))
    folder = pathlib.Path(label_folder)
    return [get_label_anno(folder / (get_image_index_str(i) + ".txt")) for i in image_ids]


def filter_annos_low_score(image_annos, thresh):
    """kitti_common.py:215-227"""
    out = []
    for anno in image_annos:
        keep = [i for i, s in enumerate(anno["score"]) if s >= thresh]
        out.append({k: anno[k][keep] for k in anno.keys()})
    return out


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
# eval.py
# ----------------------------------------------------------------------------------------------------------------
def get_thresholds(scores, num_gt, num_sample_pts=41):
    """eval.py:7-26"""
    scores = np.sort(scores)[::-1]
    current_recall = 0
    thresholds = []
    for i, score in enumerate(scores):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < (len(scores) - 1) else l_recall
        if ((r_recall - current_recall) < (current_recall - l_recall)) and (i < (len(scores) - 1)):
            continue
        thresholds.append(score)
        current_recall += 1 / (num_sample_pts - 1.0)
    return thresholds


def clean_data(gt_anno, dt_anno, current_class, difficulty):
    """eval.py:29-84"""
    CLASS_NAMES = ["car", "pedestrian", "cyclist"]
    MIN_HEIGHT = [40, 25, 25]
    MAX_OCCLUSION = [0, 1, 2]
    MAX_TRUNCATION = [0.15, 0.3, 0.5]
    dc_bboxes, ignored_gt, ignored_dt = [], [], []
    current_cls_name = CLASS_NAMES[current_class].lower()
    num_valid_gt = 0
    for i in range(len(gt_anno["name"])):
        bbox = gt_anno["bbox"][i]
        gt_name = gt_anno["name"][i].lower()
        height = bbox[3] - bbox[1]
        if gt_name == current_cls_name:
            valid_class = 1
        elif current_cls_name == "pedestrian" and gt_name == "person_sitting":
            valid_class = 0
        elif current_cls_name == "car" and gt_name == "van":
            valid_class = 0
        else:
            valid_class = -1
        ignore = (gt_anno["occluded"][i] > MAX_OCCLUSION[difficulty]) or (gt_anno["truncated"][i] > MAX_TRUNCATION[difficulty]) \
            or (height <= MIN_HEIGHT[difficulty])
        if valid_class == 1 and not ignore:
            ignored_gt.append(0)
            num_valid_gt += 1
        elif valid_class == 0 or (ignore and valid_class == 1):
            ignored_gt.append(1)
        else:
            ignored_gt.append(-1)
        if gt_anno["name"][i] == "DontCare":
            dc_bboxes.append(gt_anno["bbox"][i])
    for i in range(len(dt_anno["name"])):
        valid_class = 1 if dt_anno["name"][i].lower() == current_cls_name else -1
        height = abs(dt_anno["bbox"][i, 3] - dt_anno["bbox"][i, 1])
        if height < MIN_HEIGHT[difficulty]:
            ignored_dt.append(1)
        elif valid_class == 1:
            ignored_dt.append(0)
        else:
            ignored_dt.append(-1)
    return num_valid_gt, ignored_gt, ignored_dt, dc_bboxes


def _offsets(counts):
    off = np.zeros(len(counts) + 1, np.int64)
    np.cumsum(counts, out=off[1:])
    return off


def _boxes7(annos):
    """(x, y, z, l, h, w, ry) rows of a list of annotation dicts (eval.py:372-384)"""
    parts = [np.concatenate([a["location"].reshape(-1, 3), a["dimensions"].reshape(-1, 3), a["rotation_y"].reshape(-1, 1)], 1)
             for a in annos]
    return np.concatenate(parts, 0) if parts else np.zeros((0, 7))


def calculate_iou_partly(gt_annos, dt_annos, metric, backend):
    """eval.py:326-397, per-frame blocks only.  Called like the reference's eval_class does -- with (dt_annos, gt_annos) --
    so the FIRST argument indexes the rows.  -> (list of per-frame (n_first, n_second) float64 arrays, flat, ov_off)"""
    assert len(gt_annos) == len(dt_annos)
    n_first = np.array([len(a["name"]) for a in gt_annos], np.int64)
    n_second = np.array([len(a["name"]) for a in dt_annos], np.int64)
    off1, off2 = _offsets(n_first), _offsets(n_second)
    ov_off = _offsets(n_first * n_second)
    if metric == 0:
        b1 = np.concatenate([a["bbox"].reshape(-1, 4) for a in gt_annos], 0) if len(gt_annos) else np.zeros((0, 4))
        b2 = np.concatenate([a["bbox"].reshape(-1, 4) for a in dt_annos], 0) if len(dt_annos) else np.zeros((0, 4))
    elif metric in (1, 2):
        b1, b2 = _boxes7(gt_annos), _boxes7(dt_annos)
    else:
        raise ValueError("unknown metric")
    flat = backend.overlaps(metric, b1, off1, b2, off2, ov_off)
    blocks = [flat[ov_off[i]:ov_off[i + 1]].reshape(n_first[i], n_second[i]) for i in range(len(gt_annos))]
    return blocks, flat, ov_off


def _prepare_data(gt_annos, dt_annos, current_class, difficulty):
    """eval.py:400-440, concatenated over frames with offsets"""
    ign_gt, ign_det, dcs, gt_datas, dt_datas, dc_counts = [], [], [], [], [], []
    total_num_valid_gt = 0
    for g, d in zip(gt_annos, dt_annos):
        num_valid_gt, ig, idt, dc = clean_data(g, d, current_class, difficulty)
        ign_gt.append(np.array(ig, dtype=np.int32).reshape(-1))
        ign_det.append(np.array(idt, dtype=np.int32).reshape(-1))
        dc = np.stack(dc, 0).astype(np.float64) if len(dc) else np.zeros((0, 4))
        dcs.append(dc)
        dc_counts.append(dc.shape[0])
        total_num_valid_gt += num_valid_gt
        gt_datas.append(np.concatenate([g["bbox"].reshape(-1, 4), g["alpha"].reshape(-1, 1)], 1))
        dt_datas.append(np.concatenate([d["bbox"].reshape(-1, 4), d["alpha"].reshape(-1, 1), d["score"].reshape(-1, 1)], 1))
    cat = lambda xs, w: np.concatenate(xs, 0) if xs else np.zeros((0, w))      # noqa: E731
    return dict(gt_datas=cat(gt_datas, 5), dt_datas=cat(dt_datas, 6), ign_gt=np.concatenate(ign_gt) if ign_gt else np.zeros(0, np.int32),
                ign_det=np.concatenate(ign_det) if ign_det else np.zeros(0, np.int32), dc=cat(dcs, 4),
                dc_off=_offsets(dc_counts), total_num_valid_gt=total_num_valid_gt)


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, backend=None):
    """eval.py:443-544.  -> dict(recall, precision, orientation), each [num_class, num_difficulty, num_minoverlap, 41]"""
    backend = backend or HipBackend()
    assert len(gt_annos) == len(dt_annos)
    _, flat, ov_off = calculate_iou_partly(dt_annos, gt_annos, metric, backend)          # rows = detections (:468)
    gt_off = _offsets([len(a["name"]) for a in gt_annos])
    dt_off = _offsets([len(a["name"]) for a in dt_annos])
    N_SAMPLE_PTS = 41
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, current_class in enumerate(current_classes):
        for l, difficulty in enumerate(difficultys):
            P = _prepare_data(gt_annos, dt_annos, current_class, difficulty)
            args = (flat, ov_off, P["gt_datas"], gt_off, P["dt_datas"], dt_off, P["ign_gt"], P["ign_det"], P["dc"], P["dc_off"], metric)
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                # pass 1 (:479-494): scores of the detections that match a valid gt at this overlap
                _, matched = backend.statistics(*args, min_overlap, np.zeros(1), False, False)
                scores = matched[~np.isnan(matched)]
                thresholds = np.array(get_thresholds(scores, P["total_num_valid_gt"]))
                if len(thresholds) == 0:
                    continue
                # pass 2 (:496-524): tp / fp / fn / similarity per (frame, threshold), summed over frames in frame order
                res, _ = backend.statistics(*args, min_overlap, thresholds, True, compute_aos)
                res = res.reshape(len(gt_annos), len(thresholds), 4)
                pr = np.zeros([len(thresholds), 4])
                for f in range(res.shape[0]):
                    pr[:, :3] += res[f, :, :3]
                    sim = res[f, :, 3]
                    pr[:, 3] += np.where(sim != -1, sim, 0.0)
                with np.errstate(divide="ignore", invalid="ignore"):
                    for i in range(len(thresholds)):
                        recall[m, l, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 2])
                        precision[m, l, k, i] = pr[i, 0] / (pr[i, 0] + pr[i, 1])
                        if compute_aos:
                            aos[m, l, k, i] = pr[i, 3] / (pr[i, 0] + pr[i, 1])
                for i in range(len(thresholds)):
                    precision[m, l, k, i] = np.max(precision[m, l, k, i:], axis=-1)
                    recall[m, l, k, i] = np.max(recall[m, l, k, i:], axis=-1)
                    if compute_aos:
                        aos[m, l, k, i] = np.max(aos[m, l, k, i:], axis=-1)
    return {"recall": recall, "precision": precision, "orientation": aos}


def get_mAP(prec):
    """eval.py:547-551 (11-point interpolation over the 41 sample points)"""
    sums = 0
    for i in range(0, prec.shape[-1], 4):
        sums = sums + prec[..., i]
    return sums / 11 * 100


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos=False, backend=None):
    """eval.py:563-583"""
    backend = backend or HipBackend()
    difficultys = [0, 1, 2]
    ret = eval_class(gt_annos, dt_annos, current_classes, difficultys, 0, min_overlaps, compute_aos, backend)
    mAP_bbox = get_mAP(ret["precision"])
    mAP_aos = get_mAP(ret["orientation"]) if compute_aos else None
    mAP_bev = get_mAP(eval_class(gt_annos, dt_annos, current_classes, difficultys, 1, min_overlaps, backend=backend)["precision"])
    mAP_3d = get_mAP(eval_class(gt_annos, dt_annos, current_classes, difficultys, 2, min_overlaps, backend=backend)["precision"])
    return mAP_bbox, mAP_bev, mAP_3d, mAP_aos


def do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges, compute_aos, backend=None):
    """eval.py:586-602"""
    min_overlaps = np.zeros([10, *overlap_ranges.shape[1:]])
    for i in range(overlap_ranges.shape[1]):
        for j in range(overlap_ranges.shape[2]):
            min_overlaps[:, i, j] = np.linspace(overlap_ranges[0, i, j], overlap_ranges[1, i, j], int(overlap_ranges[2, i, j]))
    mAP_bbox, mAP_bev, mAP_3d, mAP_aos = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, backend)
    mAP_bbox, mAP_bev, mAP_3d = mAP_bbox.mean(-1), mAP_bev.mean(-1), mAP_3d.mean(-1)
    if mAP_aos is not None:
        mAP_aos = mAP_aos.mean(-1)
    return mAP_bbox, mAP_bev, mAP_3d, mAP_aos


def print_str(value, *arg, sstream=None):
    if sstream is None:
        sstream = sysio.StringIO()
    sstream.truncate(0)
    sstream.seek(0)
    print(value, *arg, file=sstream)
    return sstream.getvalue()


_CLASS_TO_NAME = {0: "Car", 1: "Pedestrian", 2: "Cyclist", 3: "Van", 4: "Person_sitting"}


def _classes_int(current_classes):
    name_to_class = {v: n for n, v in _CLASS_TO_NAME.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    return [name_to_class[c] if isinstance(c, str) else c for c in current_classes]


def _has_alpha(dt_annos):
    for anno in dt_annos:
        if anno["alpha"].shape[0] != 0:
            return bool(anno["alpha"][0] != -10)
    return False


def get_official_eval_result(gt_annos, dt_annos, current_classes, backend=None):
    """eval.py:605-675: -> (result string, dict of the nine Car AP numbers)"""
    overlap_0_7 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5], [0.7, 0.5, 0.5, 0.7, 0.5], [0.7, 0.5, 0.5, 0.7, 0.5]])
    overlap_0_5 = np.array([[0.7, 0.5, 0.5, 0.7, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25], [0.5, 0.25, 0.25, 0.5, 0.25]])
    min_overlaps = np.stack([overlap_0_7, overlap_0_5], axis=0)
    current_classes = _classes_int(current_classes)
    min_overlaps = min_overlaps[:, :, current_classes]
    compute_aos = _has_alpha(dt_annos)
    mAPbbox, mAPbev, mAP3d, mAPaos = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, compute_aos, backend)
    result = ""
    for j, curcls in enumerate(current_classes):
        for i in range(min_overlaps.shape[0]):
            result += print_str((f"{_CLASS_TO_NAME[curcls]} " "AP@{:.2f}, {:.2f}, {:.2f}:".format(*min_overlaps[i, :, j])))
            result += print_str((f"bbox AP:{mAPbbox[j, 0, i]:.4f}, " f"{mAPbbox[j, 1, i]:.4f}, " f"{mAPbbox[j, 2, i]:.4f}"))
            result += print_str((f"bev  AP:{mAPbev[j, 0, i]:.4f}, " f"{mAPbev[j, 1, i]:.4f}, " f"{mAPbev[j, 2, i]:.4f}"))
            result += print_str((f"3d   AP:{mAP3d[j, 0, i]:.4f}, " f"{mAP3d[j, 1, i]:.4f}, " f"{mAP3d[j, 2, i]:.4f}"))
            if compute_aos:
                result += print_str((f"aos  AP:{mAPaos[j, 0, i]:.2f}, " f"{mAPaos[j, 1, i]:.2f}, " f"{mAPaos[j, 2, i]:.2f}"))
    ret_dict = {"Car_3d_easy": mAP3d[0, 0, 0], "Car_3d_moderate": mAP3d[0, 1, 0], "Car_3d_hard": mAP3d[0, 2, 0],
                "Car_bev_easy": mAPbev[0, 0, 0], "Car_bev_moderate": mAPbev[0, 1, 0], "Car_bev_hard": mAPbev[0, 2, 0],
                "Car_image_easy": mAPbbox[0, 0, 0], "Car_image_moderate": mAPbbox[0, 1, 0], "Car_image_hard": mAPbbox[0, 2, 0]}
    return result, ret_dict


def get_coco_eval_result(gt_annos, dt_annos, current_classes, backend=None):
    """eval.py:678-740"""
    class_to_range = {0: [0.5, 0.95, 10], 1: [0.25, 0.7, 10], 2: [0.25, 0.7, 10], 3: [0.5, 0.95, 10], 4: [0.25, 0.7, 10]}
    current_classes = _classes_int(current_classes)
    overlap_ranges = np.zeros([3, 3, len(current_classes)])
    for i, curcls in enumerate(current_classes):
        overlap_ranges[:, :, i] = np.array(class_to_range[curcls])[:, np.newaxis]
    compute_aos = _has_alpha(dt_annos)
    mAPbbox, mAPbev, mAP3d, mAPaos = do_coco_style_eval(gt_annos, dt_annos, current_classes, overlap_ranges, compute_aos, backend)
    result = ""
    for j, curcls in enumerate(current_classes):
        o_range = np.array(class_to_range[curcls])[[0, 2, 1]]
        o_range[1] = (o_range[2] - o_range[0]) / (o_range[1] - 1)
        result += print_str((f"{_CLASS_TO_NAME[curcls]} " "coco AP@{:.2f}:{:.2f}:{:.2f}:".format(*o_range)))
        result += print_str((f"bbox AP:{mAPbbox[j, 0]:.2f}, " f"{mAPbbox[j, 1]:.2f}, " f"{mAPbbox[j, 2]:.2f}"))
        result += print_str((f"bev  AP:{mAPbev[j, 0]:.2f}, " f"{mAPbev[j, 1]:.2f}, " f"{mAPbev[j, 2]:.2f}"))
        result += print_str((f"3d   AP:{mAP3d[j, 0]:.2f}, " f"{mAP3d[j, 1]:.2f}, " f"{mAP3d[j, 2]:.2f}"))
        if compute_aos:
            result += print_str((f"aos  AP:{mAPaos[j, 0]:.2f}, " f"{mAPaos[j, 1]:.2f}, " f"{mAPaos[j, 2]:.2f}"))
    return result


def evaluate(label_path, result_path, label_split_file, current_class=0, coco=False, score_thresh=-1, backend=None):
    """tools/kitti_object_eval_python/evaluate.py:14-28 (what tools/eval_rcnn.py:449,677 calls as kitti_evaluate)"""
    dt_annos = get_label_annos(result_path)
    if score_thresh > 0:
        dt_annos = filter_annos_low_score(dt_annos, score_thresh)
    with open(label_split_file, "r") as f:
        val_image_ids = [int(line) for line in f.readlines()]
    gt_annos = get_label_annos(label_path, val_image_ids)
    if coco:
        return get_coco_eval_result(gt_annos, dt_annos, current_class, backend)
    return get_official_eval_result(gt_annos, dt_annos, current_class, backend)
