"""Oracle (test infrastructure only): box helpers, target encoder, decoder, NMS.

numpy fp32 restatements.  All arithmetic is done on ``np.float32`` scalars /
arrays with one rounding per reference op (no FMA), because the NMS keep
indices and decoder class indices must be bit-exact.
"""
import numpy as np

f32 = np.float32


def compute_iou_matrix(b1, b2):
    """Pairwise IoU, no +1, negative w/h clipped to 0.

    Follows reference utils/utils.py:10-57: left_top = max, right_bottom = min
    (:38,:43), ``w_h[w_h<0]=0`` (:46), ``I/(a1+a2-I)`` (:55).
    b1 [N,4], b2 [M,4] (x1,y1,x2,y2) -> [N,M] fp32.
    """
    b1 = np.asarray(b1, dtype=f32)
    b2 = np.asarray(b2, dtype=f32)
    lt = np.maximum(b1[:, None, :2], b2[None, :, :2])
    rb = np.minimum(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).astype(f32)
    wh[wh < 0] = 0
    inter = (wh[..., 0] * wh[..., 1]).astype(f32)
    a1 = ((b1[:, 2] - b1[:, 0]).astype(f32) * (b1[:, 3] - b1[:, 1]).astype(f32)).astype(f32)
    a2 = ((b2[:, 2] - b2[:, 0]).astype(f32) * (b2[:, 3] - b2[:, 1]).astype(f32)).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        union = ((a1[:, None] + a2[None, :]).astype(f32) - inter).astype(f32)
        return (inter / union).astype(f32)


def convert_cxcywh_to_x1y1x2y2(t, S):
    """``[x/S - w/2, y/S - h/2, x/S + w/2, y/S + h/2]`` (utils/utils.py:59-75).

    The cell origin is omitted on purpose (it cancels inside a cell).  ``/S``
    is a true fp32 division by the integer S (:72), ``0.5*wh`` is exact.
    """
    t = np.asarray(t, dtype=f32)
    assert t.shape[-1] == 4, "convert position tensor must [n, 4]"
    out = np.empty_like(t)
    c = (t[:, :2] / f32(S)).astype(f32)
    half = (f32(0.5) * t[:, 2:]).astype(f32)
    out[:, :2] = c - half
    out[:, 2:] = c + half
    return out


def encode_target(boxes, labels, S, B=2, C=20):
    """Target encoder, utils/YOLODataLoader.py:200-230.

    boxes [k,4] normalised (cx,cy,w,h); labels [k].  Cell index is
    ``ceil(c / cell_size) - 1`` with ``cell_size = 1./S`` a Python double that
    torch casts to fp32 for the tensor division (:207,:219); row = y, col = x
    (:220); the whole cell is zeroed first so the last writer wins (:220);
    both conf slots are 1 (:221); the box is duplicated into every slot
    (:225-227).
    """
    D = B * 5 + C
    tgt = np.zeros((S, S, D), dtype=f32)
    boxes = np.asarray(boxes, dtype=f32).reshape(-1, 4)
    cs = f32(1.0 / S)
    for k in range(boxes.shape[0]):
        cxcy = boxes[k, :2]
        ij = np.ceil((cxcy / cs).astype(f32)) - f32(1)
        col, row = int(ij[0]), int(ij[1])
        tgt[row, col, :] = 0
        tgt[row, col, :B] = 1
        tgt[row, col, B * 5 + int(labels[k])] = 1
        xy = (ij * cs).astype(f32)
        dxy = ((cxcy - xy).astype(f32) / cs).astype(f32)
        for b in range(B):
            tgt[row, col, B + 4 * b:B + 4 * b + 2] = dxy
            tgt[row, col, B + 4 * b + 2:B + 4 * b + 4] = boxes[k, 2:]
    return tgt


def decode_candidates(pred, S, B=2, thresh=0.3):
    """Decode stage of ``decoder`` (utils/utils.py:94-141) for ONE image.

    pred [S,S,B*5+C] fp32.  Returns (boxes[n,4] f32, cls[n] int64, probs[n]
    f32, slots[n] int64) in (row i, col j, box b) order, where
    slot = (i*S + j)*B + b.  n may be 0 (the caller substitutes the reference's
    single zero box, :134-137).

      mask  = conf > fp32(1e-4)  or  conf == max(conf)            (:108-114)
      cxcy  = box[:2]*fp32(1/S) + fp32([j,i])*fp32(1/S)           (:122-123)
      xyxy  = cxcy -/+ 0.5*wh                                     (:125-126)
      score = conf * max_c cls  (first max index)                 (:127,:132)
      keep  = double(score) > thresh (a Python double compare)    (:129)
    The reference's in-place write into ``pred`` (T7) is not reproduced.
    """
    pred = np.asarray(pred, dtype=f32)
    cs = f32(1.0 / S)
    conf = pred[:, :, :B]
    cmax = conf.max()
    mask = (conf > f32(0.0001)) | (conf == cmax)
    boxes, cls, probs, slots = [], [], [], []
    for i in range(S):
        for j in range(S):
            for b in range(B):
                if not mask[i, j, b]:
                    continue
                box = pred[i, j, B + 4 * b:B + 4 * b + 4]
                cx = f32(f32(box[0] * cs) + f32(f32(j) * cs))
                cy = f32(f32(box[1] * cs) + f32(f32(i) * cs))
                hw = f32(f32(0.5) * box[2])
                hh = f32(f32(0.5) * box[3])
                c = pred[i, j, 5 * B:]
                ci = int(np.argmax(c))  # first max, as torch.max(dim) on CPU
                score = f32(pred[i, j, b] * c[ci])
                if float(score) > float(thresh):
                    boxes.append([f32(cx - hw), f32(cy - hh), f32(cx + hw), f32(cy + hh)])
                    cls.append(ci)
                    probs.append(score)
                    slots.append((i * S + j) * B + b)
    return (np.asarray(boxes, dtype=f32).reshape(-1, 4), np.asarray(cls, dtype=np.int64),
            np.asarray(probs, dtype=f32), np.asarray(slots, dtype=np.int64))


def nms(bboxes, scores, threshold=0.25):
    """Greedy class-agnostic NMS, utils/utils.py:150-184, PyTorch-0.4 semantics.

    Line by line: areas (:159); sort descending (:161); loop: keep the top
    (:164-165), stop if it was the last (:167-168); IoU of the top against the
    rest via clamp (:170-179) as ``inter / ((area_i + area_j) - inter)``;
    survivors ``ovr <= fp32(threshold)`` (:180); none -> stop (:181-182).
    The 0-dim ``squeeze()`` crash of torch>=0.5 is replaced by
    what 0.4 did: a single survivor is kept and the loop ends.
    NaN overlap (0/0) compares false -> suppressed.
    Score ties: the reference's sort is unstable; the defined order here (and
    in the HIP kernel) is descending score, then ascending index.
    Returns int64 indices into the input, in keep order.
    """
    bboxes = np.asarray(bboxes, dtype=f32).reshape(-1, 4)
    scores = np.asarray(scores, dtype=f32).reshape(-1)
    x1, y1, x2, y2 = bboxes[:, 0], bboxes[:, 1], bboxes[:, 2], bboxes[:, 3]
    areas = ((x2 - x1).astype(f32) * (y2 - y1).astype(f32)).astype(f32)
    order = np.argsort(-scores, kind="stable")
    thr = f32(threshold)
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        if order.size == 1:
            break
        rest = order[1:]
        xx1 = np.maximum(x1[rest], x1[i])
        yy1 = np.maximum(y1[rest], y1[i])
        xx2 = np.minimum(x2[rest], x2[i])
        yy2 = np.minimum(y2[rest], y2[i])
        w = np.maximum((xx2 - xx1).astype(f32), f32(0))
        h = np.maximum((yy2 - yy1).astype(f32), f32(0))
        inter = (w * h).astype(f32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = (inter / (((areas[i] + areas[rest]).astype(f32)) - inter).astype(f32)).astype(f32)
        ids = np.nonzero(ovr <= thr)[0]
        if ids.size == 0:
            break
        order = order[ids + 1]
    return np.asarray(keep, dtype=np.int64)


def decoder(pred, grid_num=7, B=2, thresh=0.3, nms_th=0.5, gt=False):
    """Full ``decoder`` for one image (utils/utils.py:94-147).

    pred [1,S,S,D] or [S,S,D].  Returns (boxes[K,4], cls[K] int64, probs[K],
    keep[K] int64 candidate indices).  Zero candidates -> one zero box
    (:134-137), which NMS keeps (n == 1).
    """
    pred = np.asarray(pred, dtype=f32)
    if pred.ndim == 4:
        pred = pred[0]
    boxes, cls, probs, _ = decode_candidates(pred, grid_num, B, thresh)
    if boxes.shape[0] == 0:
        boxes = np.zeros((1, 4), f32)
        probs = np.zeros((1,), f32)
        cls = np.zeros((1,), np.int64)
    keep = nms(boxes, probs, 1.0 if gt else nms_th)
    return boxes[keep], cls[keep], probs[keep], keep
