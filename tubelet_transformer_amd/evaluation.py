"""Evaluation loop, result files and frame-mAP.

Mirrors ``utils/video_action_recognition.py:222-454`` (``validate_tuber_detection``): eval-mode forward on the HIP path,
``postprocessors['bbox']``, the per-rank text files ``{BASE_PATH}/{RES_DIR}/{rank}.txt`` / ``GT_{rank}.txt`` in the reference's
exact format (``"{image_id} [x1, y1, x2, y2, <C scores>, <actor probs>]"`` and ``"{image_id} [<6 raw-box numbers>, <C labels>]"``),
so the reference's own evaluator can be pointed at them unchanged, and a compact numpy restatement of the metric it computes
(``evaluates/evaluate_ava.py:17-171`` driving the vendored PASCAL evaluator ``evaluates/utils/object_detection_evaluation.py:309``:
per-class VOC average precision at IoU >= 0.5, greedy score-ordered matching, one detection per ground-truth box, classes without
ground truth excluded from the mean).  ``tests/golden/frame_map_case.json`` pins it against the reference evaluator run in the
build container (``oracle/gen_eval_golden.py``).  Host-side code: runs once per epoch on rank 0 (not throughput relevant).
"""
import glob
import math
import os
import time

import numpy as np
import torch


# ---------------------------------------------------------------------------------------------------------------------
# result files
# ---------------------------------------------------------------------------------------------------------------------
def write_result_files(base_path, res_dir, rank, det_ids, det_boxes, det_scores, det_binary, gt_ids, gt_boxes, gt_labels):
    """video_action_recognition.py:397-407: one line per (frame, query) / per ground-truth box."""
    d = os.path.join(base_path, res_dir)
    os.makedirs(d, exist_ok=True)
    det_path = os.path.join(d, "%d.txt" % rank)
    gt_path = os.path.join(d, "GT_%d.txt" % rank)
    with open(det_path, "w") as f:
        for x in range(len(det_ids)):
            data = np.concatenate([det_boxes[x], det_scores[x], det_binary[x]])
            f.write("{} {}\n".format(det_ids[x], data.tolist()))
    with open(gt_path, "w") as f:
        for x in range(len(gt_ids)):
            data = np.concatenate([gt_boxes[x], gt_labels[x]])
            f.write("{} {}\n".format(gt_ids[x], data.tolist()))
    return det_path, gt_path


def _parse(line):
    key = line.split(" [")[0]
    vals = [float(v) for v in line.split(" [")[1].split("]")[0].split(",")]
    return key, vals


# ---------------------------------------------------------------------------------------------------------------------
# frame-mAP
# ---------------------------------------------------------------------------------------------------------------------
def _iou_one_to_many(box, boxes):
    x1 = np.maximum(box[0], boxes[:, 0]); y1 = np.maximum(box[1], boxes[:, 1])
    x2 = np.minimum(box[2], boxes[:, 2]); y2 = np.minimum(box[3], boxes[:, 3])
    inter = np.maximum(x2 - x1, 0.0) * np.maximum(y2 - y1, 0.0)
    a = (box[2] - box[0]) * (box[3] - box[1])
    b = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    return inter / (a + b - inter)


def _average_precision(precision, recall):
    """area under the monotone precision envelope (VOC 2010+; object_detection/utils/metrics.py:compute_average_precision)"""
    if precision is None or len(precision) == 0:
        return float("nan")
    r = np.concatenate([[0.0], recall, [1.0]])
    p = np.concatenate([[0.0], precision, [0.0]])
    for i in range(len(p) - 2, -1, -1):
        p[i] = max(p[i], p[i + 1])
    idx = np.where(r[1:] != r[:-1])[0] + 1
    return float(np.sum((r[idx] - r[idx - 1]) * p[idx]))


class FrameMAP:
    """Frame-level mean average precision from the result files (STDetectionEvaluater semantics, evaluate_ava.py:17-171)."""

    def __init__(self, class_num, class_whitelist=None, exclude_keys=(), iou_threshold=0.5, gt_min_score=1e-2):
        self.class_num, self.iou = class_num, iou_threshold
        self.whitelist = set(class_whitelist) if class_whitelist is not None else None
        self.exclude = set(exclude_keys)
        self.gt_min_score = gt_min_score
        self.gt, self.det = {}, {}

    def _wanted(self, cls):
        return self.whitelist is None or cls in self.whitelist

    def load_gt(self, paths):
        for path in paths:
            for line in open(path):
                key, v = _parse(line)
                if key in self.exclude:
                    continue
                labels = np.asarray(v[6:])
                for x in np.nonzero(labels > self.gt_min_score)[0]:
                    if self._wanted(int(x) + 1):
                        self.gt.setdefault(key, []).append((int(x) + 1, np.asarray(v[2:6], dtype=float)))

    def load_detections(self, paths):
        for path in paths:
            for line in open(path):
                key, v = _parse(line)
                if key in self.exclude:
                    continue
                box = np.asarray(v[0:4], dtype=float)
                scores = v[4:self.class_num + 4]
                for x, s in enumerate(scores):
                    if self._wanted(x + 1):
                        self.det.setdefault(key, []).append((x + 1, box, float(s)))

    def evaluate(self):
        """-> (mAP, {class_id: AP}).  Detections of images without ground truth count as false positives (the PASCAL evaluator
        scores them against an empty ground-truth list, object_detection_evaluation.py:601-620)."""
        # The data flow (and therefore the resolution of score ties) follows the reference exactly: per image all (box, class)
        # entries in file order are ordered by np.argsort(-score) (evaluate_ava.py:150), matched greedily per class in that
        # order (per_image_evaluation.py:354-366), concatenated per class in image order, and ranked by np.argsort(score)[::-1]
        # (metrics.py:56-57).  numpy's default sort is deterministic, so equal inputs give the reference's permutation.
        per_class = {}
        n_gt = {}
        for key, items in self.gt.items():
            for cls, _ in items:
                n_gt[cls] = n_gt.get(cls, 0) + 1
        scores, tps = {}, {}
        for key, dets in self.det.items():
            gts = self.gt.get(key, [])
            cls_a = np.asarray([d[0] for d in dets], dtype=int)
            box_a = np.vstack([d[1] for d in dets])
            sc_a = np.asarray([d[2] for d in dets], dtype=float)
            index = np.argsort(-sc_a)
            cls_a, box_a, sc_a = cls_a[index], box_a[index], sc_a[index]
            valid = (box_a[:, 0] < box_a[:, 2]) & (box_a[:, 1] < box_a[:, 3])      # per_image_evaluation.py:445-449
            cls_a, box_a, sc_a = cls_a[valid], box_a[valid], sc_a[valid]
            for cls in range(1, self.class_num + 1):
                sel = cls_a == cls
                if not sel.any():
                    continue
                gboxes = np.asarray([b for c, b in gts if c == cls], dtype=float).reshape(-1, 4)
                taken = np.zeros(len(gboxes), dtype=bool)
                tp = np.zeros(int(sel.sum()), dtype=bool)
                if len(gboxes):
                    for i, box in enumerate(box_a[sel]):
                        iou = _iou_one_to_many(box, gboxes)
                        j = int(np.argmax(iou))
                        if iou[j] >= self.iou and not taken[j]:
                            taken[j] = True
                            tp[i] = True
                scores.setdefault(cls, []).append(sc_a[sel])
                tps.setdefault(cls, []).append(tp)
        for cls in range(1, self.class_num + 1):
            if not self._wanted(cls) or n_gt.get(cls, 0) == 0:
                continue
            if cls not in scores:
                per_class[cls] = 0.0
                continue
            s = np.concatenate(scores[cls]); t = np.concatenate(tps[cls])
            order = np.argsort(s)[::-1]
            t = t[order]
            ctp = np.cumsum(t).astype(float); cfp = np.cumsum(~t).astype(float)
            per_class[cls] = _average_precision(ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps), ctp / n_gt[cls])
        # the reference's mean: np.nanmean over an array indexed by class id up to the largest category id, NaN where a class has
        # no ground truth (object_detection_evaluation.py: average_precision_per_class) -- same summation order, same bits
        ncat = getattr(self, "num_categories", None) or (max(self.whitelist) if self.whitelist else self.class_num)
        arr = np.full(max(ncat, max(per_class) if per_class else 0), np.nan)
        for cls, ap in per_class.items():
            arr[cls - 1] = ap
        mAP = float(np.nanmean(arr)) if per_class else float("nan")
        return mAP, per_class


def read_labelmap(path):
    """utils/utils.py:10-25: pbtxt label map -> ([{id, name}], {ids})."""
    labelmap, ids, name = [], set(), ""
    with open(path) as f:
        for line in f:
            if line.startswith("  name:"):
                name = line.split('"')[1]
            elif line.startswith("  id:") or line.startswith("  label_id:"):
                cid = int(line.strip().split(" ")[-1])
                labelmap.append({"id": cid, "name": name})
                ids.add(cid)
    return labelmap, ids


# ---------------------------------------------------------------------------------------------------------------------
# the loop
# ---------------------------------------------------------------------------------------------------------------------
def _forward_checked(model, samples):
    """model(samples) of a validation iteration.  The cooperative decoder launch (csrc/decoder_coop.hip) fails safe -- a timed-out
    barrier leaves NaN outputs and an error word -- and the loop copies the outputs to the host right after this anyway, so the word is
    read here (16 bytes) every iteration: on a failure the engine switches to the launch chain and the batch is run again (ADVICE r05:
    the evaluation path never looked at the word)."""
    outputs = model(samples)
    store = model.engine()[0]
    if not store.coop_off and store.coop_failed():
        outputs = model(samples)
    return outputs


@torch.no_grad()
def validate_tuber_detection(cfg, model, criterion, postprocessors, data_loader, epoch, writer=None, excluded_timestamps=None,
                             verbose=True):
    """video_action_recognition.py:222-454.  Differences: no 30 s sleep, no hard-coded /xxx/ path (``excluded_timestamps`` = csv path
    or None), the distributed barriers are only issued when torch.distributed is initialised, the metric is FrameMAP."""
    import torch.distributed as dist
    C = cfg.CONFIG
    ddp = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if ddp else 0                   # the reference reads cfg.DDP_CONFIG.GPU_WORLD_RANK / _SIZE, which its
    world = dist.get_world_size() if ddp else 1            # launcher fills from the same process group
    model.eval()
    criterion.eval()
    dev = next(model.parameters()).device
    res = os.path.join(C.LOG.BASE_PATH, C.LOG.RES_DIR)
    if rank == 0:
        os.makedirs(res, exist_ok=True)
        for p in glob.glob(os.path.join(res, "*.txt")):
            os.remove(p)
    det_scores, det_boxes, det_binary, det_ids, gt_labels, gt_boxes, gt_ids = [], [], [], [], [], [], []
    meters = {k: [0.0, 0] for k in ("loss", "loss_bbox", "loss_giou", "loss_ce", "loss_ce_b", "class_error")}
    end = time.time()
    for idx, data in enumerate(data_loader):
        samples, targets = data[0], data[1]
        samples = samples.to(dev)            # reference :280 / :513; runs the HIP clip pre-pass when the loader yields ClipBatch
        batch_id = [t["image_id"] for t in targets]
        targets = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in t.items() if k != "image_id"} for t in targets]
        outputs = _forward_checked(model, samples)
        loss_dict = criterion(outputs, targets)
        sizes = torch.stack([t["size"] for t in targets], dim=0)
        scores, boxes, output_b = postprocessors["bbox"](outputs, sizes)
        Q = C.MODEL.QUERY_NUM
        for b in range(scores.shape[0]):
            frame_id, key_pos = batch_id[b][0], batch_id[b][1]
            if not C.MODEL.SINGLE_FRAME:
                k = key_pos // C.MODEL.DS_RATE
                sl = slice(k * Q, (k + 1) * Q)
                det_scores.append(scores[b, sl]); det_boxes.append(boxes[b, sl]); det_binary.append(output_b[b, sl])
            else:
                det_scores.append(scores[b]); det_boxes.append(boxes[b]); det_binary.append(output_b[b])
            det_ids.extend([frame_id] * Q)
            raw = targets[b]["raw_boxes"]
            sel = (raw[:, 1] == key_pos).nonzero().reshape(-1)
            lab = targets[b]["labels"][sel].reshape(len(sel), -1)
            rb = raw[sel].reshape(len(sel), -1)
            gt_labels.append(lab.cpu().numpy()); gt_boxes.append(rb.cpu().numpy())
            first = float(targets[0]["raw_boxes"][0, 0])
            gt_ids.extend(batch_id[int(float(rb[x, 0]) - first)][0] for x in range(len(rb)))
        wd = criterion.weight_dict
        total = float(sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd))
        if not math.isfinite(total):
            raise FloatingPointError("loss is %r in evaluation: %r" % (total, {k: float(v) for k, v in loss_dict.items()}))
        n = len(targets)
        for k, v in (("loss", total), ("loss_bbox", loss_dict["loss_bbox"]), ("loss_giou", loss_dict["loss_giou"]),
                     ("loss_ce", loss_dict["loss_ce"]), ("loss_ce_b", loss_dict.get("loss_ce_b", 0.0)),
                     ("class_error", loss_dict["class_error"])):
            meters[k][0] += float(v) * n; meters[k][1] += n
        if verbose and rank == 0:
            print("Epoch: [%d][%d/%d]  batch time %.3f  " % (epoch, idx + 1, len(data_loader), time.time() - end) +
                  ", ".join("%s: %.3f" % (k, s / max(c, 1)) for k, (s, c) in meters.items()))
        end = time.time()
    cat = lambda xs, w: np.concatenate(xs, axis=0) if xs else np.zeros((0, w))
    nc = C.DATA.NUM_CLASSES
    write_result_files(C.LOG.BASE_PATH, C.LOG.RES_DIR, rank, det_ids, cat(det_boxes, 4), cat(det_scores, nc), cat(det_binary, 1),
                       gt_ids, cat(gt_boxes, 6), cat(gt_labels, nc))
    if writer is not None and rank == 0:
        for k, (s, c) in meters.items():
            writer.add_scalar("val/" + k, s / max(c, 1), epoch)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    mAP = 0.0
    if rank == 0:
        white = None
        if getattr(C.DATA, "LABEL_PATH", None) and os.path.isfile(C.DATA.LABEL_PATH):
            _, white = read_labelmap(C.DATA.LABEL_PATH)
        excl = []
        if excluded_timestamps and os.path.isfile(excluded_timestamps):
            excl = [l.strip().replace(",", "_") for l in open(excluded_timestamps) if l.strip()]
        ev = FrameMAP(nc, class_whitelist=white if nc == 80 else None, exclude_keys=excl)
        ev.load_gt([os.path.join(res, "GT_%d.txt" % r) for r in range(world)])
        ev.load_detections([os.path.join(res, "%d.txt" % r) for r in range(world)])
        mAP, per_class = ev.evaluate()
        if verbose:
            print("mAP: %.5f" % mAP)
        if writer is not None:
            writer.add_scalar("val/val_mAP_epoch", mAP, epoch)
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    return mAP


# ---------------------------------------------------------------------------------------------------------------------
# JHMDB / UCF101-24 (utils/video_action_recognition.py:456-689, evaluates/evaluate_ucf.py)
# ---------------------------------------------------------------------------------------------------------------------
class FrameMAPUCF(FrameMAP):
    """``STDetectionEvaluaterUCF`` semantics (evaluates/evaluate_ucf.py:22-166) on top of the same PASCAL matching:
    ground-truth lines whose box is smaller than 10 px^2 put their image on an exclude list (:60-62); a detection line is
    ``[x1,y1,x2,y2, <C class probabilities>, <no-object probability>]`` and counts ONCE, as its arg-max class with that score,
    unless the no-object column is the overall arg-max (:109-126).  Pinned against the reference evaluator by
    ``oracle/gen_eval_golden.py`` -> ``tests/golden/frame_map_ucf_case.json``."""

    def __init__(self, class_num=24, iou_threshold=0.5):
        super().__init__(class_num, None, (), iou_threshold)
        self.num_categories = 24          # evaluate_ucf.py:15-20: the category list is the 24 UCF101-24 names whatever class_num is

    def load_gt(self, paths):
        for path in paths:
            for line in open(path):
                key, v = _parse(line)
                if (v[4] - v[2]) * (v[5] - v[3]) < 10:
                    self.exclude.add(key)
                    continue
                labels = np.asarray(v[6:])
                self.gt.setdefault(key, [])
                for x in np.nonzero(~(labels <= 1e-2))[0]:
                    self.gt[key].append((int(x) + 1, np.asarray(v[2:6], dtype=float)))
        self.gt = {k: g for k, g in self.gt.items() if g}

    def load_detections(self, paths):
        for path in paths:
            for line in open(path):
                key, v = _parse(line)
                if key in self.exclude:
                    continue
                rest = np.asarray(v[4:])
                if int(np.argmax(rest)) == len(rest) - 1:
                    continue
                scores = np.asarray(v[4:self.class_num + 4])
                x = int(np.argmax(scores))
                self.det.setdefault(key, []).append((x + 1, np.asarray(v[0:4], dtype=float), float(scores[x])))


@torch.no_grad()
def validate_tuber_ucf_detection(cfg, model, criterion, postprocessors, data_loader, epoch, writer=None, verbose=True):
    """utils/video_action_recognition.py:456-689 (called by train_tuber_jhmdb.py:83 / eval_tuber_jhmdb.py:77): eval-mode forward on
    the HIP path, ``PostProcess``, the key frame's QUERY_NUM tubelet queries of every clip written to ``{rank}.txt`` (box + C+1
    class probabilities), ``binary_{rank}.txt`` (visibility probabilities) and ``GT_{rank}.txt`` (raw box + one-hot label), then
    frame-mAP@0.5 by ``FrameMAPUCF`` on rank 0.  Differences from the reference: barriers only when torch.distributed is
    initialised; the one-hot width is max(21, NUM_CLASSES) (the reference hard-codes 21, :564, which UCF101-24 would overflow)."""
    import torch.distributed as dist
    C = cfg.CONFIG
    ddp = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank() if ddp else 0
    world = dist.get_world_size() if ddp else 1
    model.eval()
    criterion.eval()
    dev = next(model.parameters()).device
    res = os.path.join(C.LOG.BASE_PATH, C.LOG.RES_DIR)
    if rank == 0:
        os.makedirs(res, exist_ok=True)
        for p in glob.glob(os.path.join(res, "*.txt")):
            os.remove(p)
    Q, nc = C.MODEL.QUERY_NUM, C.DATA.NUM_CLASSES
    width = max(21, nc)
    buff_output, buff_anno, buff_id, buff_binary, gt_label, gt_anno, gt_id = [], [], [], [], [], [], []
    meters = {k: [0.0, 0] for k in ("loss", "loss_bbox", "loss_giou", "loss_ce", "class_error")}
    end = time.time()
    for idx, data in enumerate(data_loader):
        samples, targets = data[0], data[1]
        samples = samples.to(dev)
        batch_id = [t["image_id"] for t in targets]
        targets = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in t.items() if k != "image_id"} for t in targets]
        outputs = _forward_checked(model, samples)
        loss_dict = criterion(outputs, targets)
        sizes = torch.stack([t["size"] for t in targets], dim=0)
        scores, boxes, output_b = postprocessors["bbox"](outputs, sizes)
        for b in range(scores.shape[0]):
            raw = targets[b]["raw_boxes"]
            if len(raw) == 0:
                continue
            frame_id, key_pos = batch_id[b][0], int(batch_id[b][1])
            buff_output.append(scores[b, key_pos * Q:(key_pos + 1) * Q, :])
            buff_anno.append(boxes[b, key_pos * Q:(key_pos + 1) * Q, :])
            for _ in range(Q):
                buff_id.append(frame_id)
                buff_binary.append(output_b[..., 0])
            lab = targets[b]["labels"]
            onehot = np.zeros((len(lab), width), dtype=np.int64)
            for i in range(len(lab)):
                onehot[i, int(lab[i])] = 1
            raw = raw.reshape(-1, raw.shape[-1])
            gt_label.append(onehot)
            gt_anno.append(raw.detach().cpu().numpy())
            first = float(targets[0]["raw_boxes"].reshape(-1, raw.shape[-1])[0, 0])
            gt_id.extend(batch_id[int(float(raw[x, 0]) - first)][0] for x in range(len(raw)))
        wd = criterion.weight_dict
        total = float(sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd))
        if not math.isfinite(total):
            raise FloatingPointError("loss is %r in evaluation: %r" % (total, {k: float(v) for k, v in loss_dict.items()}))
        n = len(targets)
        for k, v in (("loss", total), ("loss_bbox", loss_dict["loss_bbox"]), ("loss_giou", loss_dict["loss_giou"]),
                     ("loss_ce", loss_dict["loss_ce"]), ("class_error", loss_dict["class_error"])):
            meters[k][0] += float(v) * n; meters[k][1] += n
        if verbose and rank == 0:
            print("Epoch: [%d][%d/%d]  batch time %.3f  " % (epoch, idx + 1, len(data_loader), time.time() - end) +
                  ", ".join("%s: %.3f" % (k, s / max(c, 1)) for k, (s, c) in meters.items()))
        end = time.time()
    if writer is not None and rank == 0:
        for k, name in (("class_error", "class_error"), ("loss", "totall_loss"), ("loss_bbox", "loss_bbox"), ("loss_giou", "loss_giou"), ("loss_ce", "loss_ce")):
            writer.add_scalar("val/" + name, meters[k][0] / max(meters[k][1], 1), epoch)
    cat = lambda xs, w: np.concatenate(xs, axis=0) if xs else np.zeros((0, w))
    out_a, anno_a = cat(buff_output, nc + 1), cat(buff_anno, 4)
    gl, ga = cat(gt_label, width), cat(gt_anno, 6)
    with open(os.path.join(res, "%d.txt" % rank), "w") as f:
        for x in range(len(buff_id)):
            f.write("{} {}\n".format(buff_id[x], np.concatenate([anno_a[x], out_a[x]]).tolist()))
    with open(os.path.join(res, "binary_%d.txt" % rank), "w") as f:
        for x in range(len(buff_id)):
            f.write("{} {}\n".format(buff_id[x], np.asarray(buff_binary[x]).tolist()))
    with open(os.path.join(res, "GT_%d.txt" % rank), "w") as f:
        for x in range(len(gt_id)):
            f.write("{} {}\n".format(gt_id[x], np.concatenate([ga[x], gl[x]]).tolist()))
    if ddp:
        dist.barrier()
    mAP = 0
    if rank == 0:
        ev = FrameMAPUCF(class_num=nc)
        ev.load_gt([os.path.join(res, "GT_%d.txt" % r) for r in range(world)])
        ev.load_detections([os.path.join(res, "%d.txt" % r) for r in range(world)])
        mAP, per_class = ev.evaluate()
        if verbose:
            print("mAP: %.5f" % mAP)
        if writer is not None:
            writer.add_scalar("val/val_mAP_epoch", mAP, epoch)
    if ddp:
        dist.barrier()
    return mAP
