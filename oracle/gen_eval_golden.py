"""Golden vector for the frame-mAP evaluator (row N2): synthetic result files scored by the REFERENCE's own evaluator
(evaluates/evaluate_ava.py:STDetectionEvaluater over the vendored PASCAL evaluator), run in the build container only.
Writes tests/golden/frame_map_case.json = {class_num, gt_lines, det_lines, expected mAP, per-class AP}."""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import  # noqa: E402


def synth(seed=0, K=12, n_img=40, Q=15):
    rng = np.random.default_rng(seed)
    gt_lines, det_lines = [], []
    for i in range(n_img):
        key = "vid%02d_%04d" % (i % 7, 900 + i)
        n = int(rng.integers(0, 4)) if i % 9 else 0                     # some frames without ground truth
        gts = []
        for _ in range(n):
            x1, y1 = rng.uniform(0, 200, 2)
            w, h = rng.uniform(30, 150, 2)
            lab = np.zeros(K)
            lab[rng.choice(K - 2, size=int(rng.integers(1, 4)), replace=False)] = 1.0      # the last 2 classes never occur
            gts.append((np.array([x1, y1, x1 + w, y1 + h]), lab))
            gt_lines.append("%s %s" % (key, np.concatenate([[i, 16], gts[-1][0], lab]).tolist()))
        for q in range(Q):
            if gts and q < 2 * len(gts):                                # two detections per box: duplicates must be false positives
                b, lab = gts[q % len(gts)]
                box = b + rng.normal(0, 6 if q < len(gts) else 25, 4)
                sc = np.clip(lab * rng.uniform(0.3, 1.0, K) + rng.uniform(0, 0.35, K), 0, 1)
            else:
                x1, y1 = rng.uniform(0, 250, 2)
                box = np.array([x1, y1, x1 + rng.uniform(20, 120), y1 + rng.uniform(20, 120)])
                sc = rng.uniform(0, 0.5, K)
            if q == 5:
                sc = np.round(sc, 1)                                    # score ties
            det_lines.append("%s %s" % (key, np.concatenate([box, sc, rng.uniform(0, 1, 3)]).tolist()))
    return K, gt_lines, det_lines


def main():
    assert ref_import.available(), "needs /root/reference"
    ref_import.install_shims()
    sys.path.insert(0, ref_import.REF)
    for alias, typ in (("float", float), ("int", int), ("bool", bool), ("object", object)):     # numpy < 1.24 aliases the vendored
        if not hasattr(np, alias):                                                               # evaluator still uses
            setattr(np, alias, typ)
    from evaluates.evaluate_ava import STDetectionEvaluater
    K, gt_lines, det_lines = synth()
    with tempfile.TemporaryDirectory() as d:
        lm = os.path.join(d, "labels.pbtxt")
        with open(lm, "w") as f:
            for c in range(1, K + 1):
                f.write('item {\n  name: "c%d"\n  id: %d\n}\n' % (c, c))
        gp, dp = os.path.join(d, "GT_0.txt"), os.path.join(d, "0.txt")
        open(gp, "w").write("\n".join(gt_lines) + "\n")
        open(dp, "w").write("\n".join(det_lines) + "\n")
        ev = STDetectionEvaluater(lm, class_num=K)
        ev.load_GT_from_path([gp])
        ev.load_detection_from_path([dp])
        mAP, metrics = ev.evaluate()
    per_class = {k.split("/")[-1]: (None if v != v else float(v)) for k, v in metrics.items() if "PerformanceByCategory" in k}
    out = {"class_num": K, "gt_lines": gt_lines, "det_lines": det_lines, "mAP": float(mAP[0]), "per_class_ap": per_class,
           "generator": "oracle/gen_eval_golden.py (reference evaluates/evaluate_ava.py:STDetectionEvaluater, IoU 0.5)"}
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "frame_map_case.json")
    json.dump(out, open(path, "w"))
    print("reference mAP %.6f over %d classes with ground truth -> %s" % (mAP[0], sum(v is not None for v in per_class.values()), path))


def synth_ucf(seed=1, K=21, n_img=36, Q=10):
    """JHMDB-style result files (utils/video_action_recognition.py:646-662): detection lines = box + K class probabilities + the
    no-object probability; ground-truth lines = 6 raw-box numbers + one-hot label (21 wide); one tiny (< 10 px^2) box."""
    rng = np.random.default_rng(seed)
    gt_lines, det_lines = [], []
    for i in range(n_img):
        key = "clip%02d_%05d" % (i % 5, 3 + i)
        x1, y1 = rng.uniform(0, 150, 2)
        w, h = (2.0, 3.0) if i == 7 else rng.uniform(40, 120, 2)
        cls = int(rng.integers(0, K - 3))                              # the last 3 classes never occur
        lab = np.zeros(21)
        lab[cls] = 1
        box = np.array([x1, y1, x1 + w, y1 + h])
        if i % 11 != 10:                                                # a few frames without ground truth
            gt_lines.append("%s %s" % (key, np.concatenate([[i, 16], box, lab]).tolist()))
        for q in range(Q):
            p = rng.uniform(0, 1, K + 1)
            if q < 2:
                b = box + rng.normal(0, 5 if q == 0 else 30, 4)
                p[cls if rng.uniform() < 0.8 else int(rng.integers(0, K))] += 2.0
            else:
                xx, yy = rng.uniform(0, 200, 2)
                b = np.array([xx, yy, xx + rng.uniform(20, 100), yy + rng.uniform(20, 100)])
                p[K] += 1.5 if q % 3 else 0.0                           # mostly "no object" on top -> line skipped
            p = p / p.sum()
            if q == 4:
                p = np.round(p, 2)                                      # ties
            det_lines.append("%s %s" % (key, np.concatenate([b, p]).tolist()))
    return K, gt_lines, det_lines


def main_ucf():
    assert ref_import.available(), "needs /root/reference"
    ref_import.install_shims()
    for alias, typ in (("float", float), ("int", int), ("bool", bool), ("object", object)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)
    with ref_import.reference_on_path():
        from evaluates.evaluate_ucf import STDetectionEvaluaterUCF
        K, gt_lines, det_lines = synth_ucf()
        with tempfile.TemporaryDirectory() as d:
            gp, dp = os.path.join(d, "GT_0.txt"), os.path.join(d, "0.txt")
            open(gp, "w").write("\n".join(gt_lines) + "\n")
            open(dp, "w").write("\n".join(det_lines) + "\n")
            ev = STDetectionEvaluaterUCF(class_num=K)
            ev.load_GT_from_path([gp])
            ev.load_detection_from_path([dp])
            mAP, metrics = ev.evaluate()
    per_class = {k.split("/")[-1]: (None if v != v else float(v)) for k, v in metrics.items() if "PerformanceByCategory" in k}
    out = {"class_num": K, "gt_lines": gt_lines, "det_lines": det_lines, "mAP": float(mAP[0]), "per_class_ap": per_class,
           "generator": "oracle/gen_eval_golden.py:main_ucf (reference evaluates/evaluate_ucf.py:STDetectionEvaluaterUCF, IoU 0.5)"}
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "frame_map_ucf_case.json")
    json.dump(out, open(path, "w"))
    print("reference UCF/JHMDB frame-mAP %.6f over %d classes with ground truth -> %s" % (mAP[0], sum(v is not None for v in per_class.values()), path))


if __name__ == "__main__":
    if "--ucf" in sys.argv:
        main_ucf()
    else:
        main()
