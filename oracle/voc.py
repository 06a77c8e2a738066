"""Oracle (test infrastructure only): VOC AP / mAP, numpy.

Restates reference utils/utils.py:215-319 (``voc_ap``, ``voc_eval``) including
its quirk that a class with zero detections appends ap = -1 and then BREAKS
out of the class loop (:248-255), so later classes are not evaluated.
Known answer shipped by the reference (``test_eval``, :321-324): cat 0.8333,
dog 1.0, mAP 0.916666...
"""
import numpy as np


def voc_ap(rec, prec, use_07_metric=False):
    """utils/utils.py:215-238."""
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap += p / 11.0
        return ap
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = max(mpre[i - 1], mpre[i])
    idx = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[idx + 1] - mrec[idx]) * mpre[idx + 1]))


def voc_eval(preds, target, classes, threshold=0.5, use_07_metric=False):
    """utils/utils.py:240-319.  Returns (mAP, [ap per evaluated class]).

    preds  {class: [[image_id, conf, x1,y1,x2,y2], ...]}
    target {(image_id, class): [[x1,y1,x2,y2], ...]}  (consumed: matched GT
    boxes are removed, :296-298)
    Overlap uses the +1 pixel convention (:285-289), strict ``> threshold``
    (:294), greedy first-match over the GT list order (:278-299).
    """
    target = {k: [list(b) for b in v] for k, v in target.items()}
    aps = []
    for class_ in classes:
        pred = preds.get(class_, [])
        if len(pred) == 0:
            aps.append(-1)
            break                                             # :249-255
        image_ids = [x[0] for x in pred]
        confidence = np.array([float(x[1]) for x in pred])
        BB = np.array([x[2:] for x in pred], dtype=np.float64)
        order = np.argsort(-confidence)
        BB = BB[order, :]
        image_ids = [image_ids[i] for i in order]
        npos = 0.0
        for (k1, k2) in target:
            if k2 == class_:
                npos += len(target[(k1, k2)])
        nd = len(image_ids)
        tp = np.zeros(nd)
        fp = np.zeros(nd)
        for d, image_id in enumerate(image_ids):
            bb = BB[d]
            key = (image_id, class_)
            if key in target:
                gts = target[key]
                for g in gts:
                    iw = max(min(g[2], bb[2]) - max(g[0], bb[0]) + 1.0, 0.0)
                    ih = max(min(g[3], bb[3]) - max(g[1], bb[1]) + 1.0, 0.0)
                    inters = iw * ih
                    union = ((bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0)
                             + (g[2] - g[0] + 1.0) * (g[3] - g[1] + 1.0) - inters)
                    if inters / union > threshold:
                        tp[d] = 1
                        gts.remove(g)
                        if len(gts) == 0:
                            del target[key]
                        break
                fp[d] = 1 - tp[d]
            else:
                fp[d] = 1
        fp = np.cumsum(fp)
        tp = np.cumsum(tp)
        rec = tp / float(npos)
        prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
        aps.append(voc_ap(rec, prec, use_07_metric))
    return float(np.mean(aps)), aps
