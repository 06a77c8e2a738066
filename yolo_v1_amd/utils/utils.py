"""Drop-in for the hot-path part of the reference's ``utils/utils.py`` on MI355X.

``compute_iou_matrix`` (:10-57), ``convert_CxCyWH_to_X1Y1X2Y2`` (:59-75),
``decoder`` (:94-147) and ``nms`` (:150-184) keep the reference's names,
argument meaning and error behaviour, but run as HIP kernels
(csrc/decode_nms.hip).  ``decode_batch`` is the batched entry point the
reference does not have (its eval loop is batch 1, :393-411).
Host-side helpers the eval loop needs (VOC AP, ``bbox_un_norm``,
``create_logger``) stay host code, as in the reference.
"""
import logging
import os

import numpy as np
import torch

from .. import _lib

VOC_CLASSES = ('aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
               'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train',
               'tvmonitor')


def _f32c(t):
    return t.to(dtype=torch.float32).contiguous()


def compute_iou_matrix(bbox1, bbox2):
    """[N,4] x [M,4] (x1,y1,x2,y2) -> IoU [N,M]; reference utils/utils.py:10-57."""
    if not isinstance(bbox1, torch.Tensor) or not isinstance(bbox2, torch.Tensor):
        print('compute iou input must be Tensor !!!')      # same idiom as the reference (:30-32)
        exit()
    _lib.require_cuda(bbox1, bbox2)
    b1, b2 = _f32c(bbox1), _f32c(bbox2)
    n, m = b1.shape[0], b2.shape[0]
    out = torch.empty((n, m), dtype=torch.float32, device=b1.device)
    _lib.check(_lib.lib().yv1_iou_matrix(_lib.ptr(b1), n, _lib.ptr(b2), m, _lib.ptr(out), _lib.stream_ptr(b1.device)),
               "yv1_iou_matrix")
    return out


def convert_CxCyWH_to_X1Y1X2Y2(input_tensor, S, B, device):
    """[n,4] (cx,cy,w,h) -> (x/S - w/2, y/S - h/2, x/S + w/2, y/S + h/2); utils/utils.py:59-75."""
    assert input_tensor.size()[-1] == 4, \
        'convert position tensor must [n, 4], but this input last dim is %d' % (input_tensor.size()[-1])
    _lib.require_cuda(input_tensor)
    x = _f32c(input_tensor)
    out = torch.empty_like(x)
    _lib.check(_lib.lib().yv1_convert_cxcywh_to_xyxy(_lib.ptr(x), x.shape[0], int(S), _lib.ptr(out),
                                                     _lib.stream_ptr(x.device)), "yv1_convert_cxcywh_to_xyxy")
    return out


def nms(bboxes, scores, threshold=0.25):
    """Greedy class-agnostic NMS; utils/utils.py:150-184 with PyTorch-0.4 semantics for the
    single-survivor round (the reference crashes there on torch >= 0.5).
    Returns a LongTensor of indices into the input, in keep order (descending score;
    score ties broken by ascending index)."""
    _lib.require_cuda(bboxes, scores)
    b, s = _f32c(bboxes).reshape(-1, 4), _f32c(scores).reshape(-1)
    n = s.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=b.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=b.device)
    _lib.check(_lib.lib().yv1_nms(_lib.ptr(b), _lib.ptr(s), n, float(threshold), _lib.ptr(keep), _lib.ptr(cnt),
                                  _lib.stream_ptr(b.device)), "yv1_nms")
    return keep[:int(cnt.item())]


def decode_batch(pred, grid_num=7, B=2, thresh=0.3, nms_th=0.5, gt=False):
    """Batched decoder + NMS, one workgroup per image.

    pred [N,S,S,B*5+C].  Returns (boxes [N,M,4], cls [N,M] int64, probs [N,M],
    keep [N,M] int64, counts [N] int32, ncand [N] int32) with M = S*S*B; row n is
    valid up to counts[n].  No host sync.
    """
    _lib.require_cuda(pred)
    p = _f32c(pred.detach())
    N, S = p.shape[0], int(grid_num)
    C = p.shape[-1] - 5 * B
    if tuple(p.shape) != (N, S, S, 5 * B + C) or C <= 0:
        raise _lib.Yv1Error("decoder expects [N,%d,%d,%d+C], got %s" % (S, S, 5 * B, tuple(p.shape)))
    M = S * S * B
    dev = p.device
    boxes = torch.empty((N, M, 4), dtype=torch.float32, device=dev)
    cls = torch.empty((N, M), dtype=torch.int64, device=dev)
    probs = torch.empty((N, M), dtype=torch.float32, device=dev)
    keep = torch.empty((N, M), dtype=torch.int64, device=dev)
    counts = torch.empty((N,), dtype=torch.int32, device=dev)
    ncand = torch.empty((N,), dtype=torch.int32, device=dev)
    nms_thresh = 1.0 if gt else float(nms_th)                        # utils/utils.py:143-145
    _lib.check(_lib.lib().yv1_decode_nms_batched(_lib.ptr(p), N, S, B, C, float(thresh), nms_thresh, _lib.ptr(boxes),
                                                 _lib.ptr(cls), _lib.ptr(probs), _lib.ptr(keep), _lib.ptr(counts),
                                                 _lib.ptr(ncand), _lib.stream_ptr(dev)), "yv1_decode_nms_batched")
    return boxes, cls, probs, keep, counts, ncand


def decoder(pred, grid_num=7, B=2, device='cpu', thresh=0.3, nms_th=0.5, gt=False):
    """pred [1,S,S,30] -> (boxes [K,4], cls_indexs [K] int64, probs [K]); utils/utils.py:94-147.

    Zero candidates give the reference's single all-zero box (shapes [1,4],[1],[1]).
    Unlike the reference, ``pred`` is not modified (not observable by any caller).
    """
    if pred.dim() == 3:
        pred = pred.unsqueeze(0)
    boxes, cls, probs, _, counts, _ = decode_batch(pred[:1], grid_num, B, thresh, nms_th, gt)
    k = int(counts[0].item())
    return boxes[0, :k], cls[0, :k], probs[0, :k]


# ---------------------------------------------------------------- host-side helpers (CPU in the reference too)
def voc_ap(rec, prec, use_07_metric=False):
    """VOC AP, area or 11-point; utils/utils.py:215-238."""
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def voc_eval(preds, target, VOC_CLASSES=VOC_CLASSES, threshold=0.5, use_07_metric=False, logger=None):
    """mAP over ``VOC_CLASSES``; utils/utils.py:240-319, including its quirk that the first class
    without detections records ap=-1 and ends the class loop (:248-255)."""
    say = logger.info if logger else print
    aps = []
    for class_ in VOC_CLASSES:
        pred = preds[class_] if class_ in preds else []
        if len(pred) == 0:
            say('---class {} ap {}---'.format(class_, -1))
            aps += [-1]
            break
        ids = [x[0] for x in pred]
        conf = np.array([float(x[1]) for x in pred])
        BB = np.array([x[2:] for x in pred], dtype=np.float64)
        order = np.argsort(-conf)
        BB = BB[order, :]
        ids = [ids[k] for k in order]
        npos = float(sum(len(v) for (k1, k2), v in target.items() if k2 == class_))
        tp = np.zeros(len(ids))
        fp = np.zeros(len(ids))
        for d, image_id in enumerate(ids):
            bb = BB[d]
            gts = target.get((image_id, class_))
            if gts is None:
                fp[d] = 1
                continue
            for g in gts:
                iw = max(min(g[2], bb[2]) - max(g[0], bb[0]) + 1., 0.)
                ih = max(min(g[3], bb[3]) - max(g[1], bb[1]) + 1., 0.)
                inter = iw * ih
                union = (bb[2] - bb[0] + 1.) * (bb[3] - bb[1] + 1.) + (g[2] - g[0] + 1.) * (g[3] - g[1] + 1.) - inter
                if inter / union > threshold:
                    tp[d] = 1
                    gts.remove(g)                       # matched GT boxes are consumed (:296-298)
                    if len(gts) == 0:
                        del target[(image_id, class_)]
                    break
            fp[d] = 1 - tp[d]
        fp = np.cumsum(fp)
        tp = np.cumsum(tp)
        rec = tp / npos
        prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
        ap = voc_ap(rec, prec, use_07_metric)
        say('---class {} ap {}---'.format(class_, ap))
        aps += [ap]
    mAP = np.mean(aps).item()
    say('---map {}---'.format(mAP))
    return mAP


def from_img_path_get_label_list(img_path, img_size=(448, 448)):
    """darknet label file next to the image -> [[label, x0, y0, x1, y1], ...] in pixels; utils/utils.py:326-345."""
    label_path = img_path.replace('JPEGImages', 'labels').replace('jpg', 'txt')
    out = []
    with open(label_path, 'r') as f:
        for line in f:
            ll = line.strip().split(' ')
            x, y, w, h = float(ll[1]), float(ll[2]), float(ll[3]), float(ll[4])
            out.append([int(ll[0]), int((x - 0.5 * w) * img_size[0]), int((y - 0.5 * h) * img_size[1]),
                        int((x + 0.5 * w) * img_size[0]), int((y + 0.5 * h) * img_size[1])])
    return out


def prep_test_data(file_path, little_test=None):
    """ground-truth dict {(image_id, class_name): [[x0,y0,x1,y1], ...]} for voc_eval; utils/utils.py:356-387."""
    from collections import defaultdict
    target = defaultdict(list)
    with open(file_path) as f:
        files = [ln.strip() for ln in f.readlines()]
    if little_test:
        files = files[:little_test]
    for image_file in files:
        image_id = image_file.split('/')[-1].split('.')[0]
        for lab in from_img_path_get_label_list(image_file):
            target[(image_id, VOC_CLASSES[lab[0]])].append(lab[1:])
    return target


def detections_from_batch(pred, fnames, preds, S=7, B=2, thresh=0.005, nms_th=.45, img_size=(448, 448)):
    """Decodes a whole batch on the GPU and appends the reference's detection records
    ``[img_id, conf, int(x1*w), int(y1*h), int(x2*w), int(y2*h)]`` (utils/utils.py:405-411) to
    ``preds[class_name]``: boxes clamped to [0,1] first, the all-zero placeholder box skipped."""
    boxes, cls, probs, _, counts, ncand = decode_batch(pred, S, B, thresh, nms_th)
    boxes = boxes.clamp(min=0., max=1.).cpu()          # one D2H copy per tensor per batch
    cls, probs, counts, ncand = cls.cpu(), probs.cpu(), counts.cpu(), ncand.cpu()
    w, h = img_size
    for n, fname in enumerate(fnames):
        if int(ncand[n]) == 0:                         # `len(confs) == 1 and confs[0] == 0` -> continue (:408)
            continue
        img_id = fname.split('/')[-1].split('.')[0]
        for j in range(int(counts[n])):
            b = boxes[n, j]
            preds[VOC_CLASSES[int(cls[n, j])]].append([img_id, float(probs[n, j]), int(b[0] * w), int(b[1] * h),
                                                       int(b[2] * w), int(b[3] * h)])
    return preds


def run_test_mAP(YOLONet, target, test_datasets, data_len, S=7, device='cuda:0', reversed=False, logger=None,
                 little_test=None, batch_size=64):
    """mAP over a dataset yielding (image, target, fname); utils/utils.py:389-418.

    The reference forwards one image at a time and decodes it in Python; here ``batch_size`` images go
    through the backbone together and one workgroup per image decodes + suppresses them on the GPU.
    Results are identical per image (eval-mode BatchNorm is batch-independent).  ``reversed`` (a layout
    swap for a third-party loss, testCodes/tensor_test.py:99-107) is not supported."""
    from collections import defaultdict
    if reversed:
        raise _lib.Yv1Error("run_test_mAP(reversed=True) is not supported")
    preds = defaultdict(list)
    n_total = len(test_datasets) if little_test is None else min(little_test, len(test_datasets))
    with torch.no_grad():
        for i0 in range(0, n_total, batch_size):
            items = [test_datasets[i] for i in range(i0, min(n_total, i0 + batch_size))]
            images = torch.stack([it[0] for it in items]).to(device)
            fnames = [it[2] for it in items]
            pred = YOLONet(images)
            detections_from_batch(pred, fnames, preds, S=S, img_size=(images.shape[3], images.shape[2]))
    (logger.info if logger else print)('---start evaluate---')
    return voc_eval(preds, target, VOC_CLASSES=VOC_CLASSES, threshold=0.5, use_07_metric=False, logger=logger)


def bbox_un_norm(bboxes, img_size=(448, 448)):
    """utils/utils.py:347-354 (in place, truncating like ``int()``)."""
    (w, h) = img_size
    for bbox in bboxes:
        bbox[0] = int(bbox[0] * w)
        bbox[1] = int(bbox[1] * h)
        bbox[2] = int(bbox[2] * w)
        bbox[3] = int(bbox[3] * h)
    return bboxes


def create_logger(base_path, log_name):
    """File + stream logger; utils/utils.py:484-504."""
    os.makedirs(base_path, exist_ok=True)
    logger = logging.getLogger(log_name)
    logger.setLevel(logging.DEBUG)
    if not logger.handlers:
        fmt = logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s')
        fh = logging.FileHandler('%s/%s.log' % (base_path, log_name))
        fh.setLevel(logging.INFO)
        sh = logging.StreamHandler()
        sh.setLevel(logging.DEBUG)
        for h in (fh, sh):
            h.setFormatter(fmt)
            logger.addHandler(h)
    return logger
