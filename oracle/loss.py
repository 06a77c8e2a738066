"""Oracle (test infrastructure only): YOLO-v1 grid loss, torch-CPU fp32.

Vectorised restatement of reference v1Loss.py:22-118 (``YOLOLossV1.forward``).
Gradients come from torch autograd over this restatement, so the hand-derived
backward inside the HIP kernel is checked against an independent derivation.

Reference quirks that are reproduced on purpose:
  T4  location loss slices ROWS of the responsible-box list: rows 0-1 plain
      squared error on x,y,w,h; rows >=2 squared error of sqrt on x,y,w,h
      (v1Loss.py:101);
  T5  the IoU "target" of the confidence loss is NOT detached -- gradient flows
      into the responsible box's coordinates (v1Loss.py:72-78, :90);
  T9  the total is divided by the constructor batch size (v1Loss.py:105).
"""
import torch


def _xyxy(box, S):
    # utils/utils.py:72-73 : xy/S -/+ 0.5*wh   (no cell origin)
    c = box[..., :2] / S
    h = 0.5 * box[..., 2:]
    return torch.cat([c - h, c + h], dim=-1)


def responsible_iou(pred_boxes, gt_box, S):
    """pred_boxes [K,B,4], gt_box [K,4] -> IoU [K,B].

    utils/utils.py:10-57 applied per object cell with N=B, M=1
    (v1Loss.py:66-72).  The clip ``w_h[w_h<0]=0`` has zero gradient on the
    clipped entries (:46).
    """
    p = _xyxy(pred_boxes, S)                       # [K,B,4]
    g = _xyxy(gt_box, S).unsqueeze(1)              # [K,1,4]
    lt = torch.max(p[..., :2], g[..., :2].expand_as(p[..., :2]))
    rb = torch.min(p[..., 2:], g[..., 2:].expand_as(p[..., 2:]))
    wh = rb - lt
    wh = torch.where(wh < 0, torch.zeros_like(wh), wh)
    inter = wh[..., 0] * wh[..., 1]
    a1 = (p[..., 2] - p[..., 0]) * (p[..., 3] - p[..., 1])
    a2 = (g[..., 2] - g[..., 0]) * (g[..., 3] - g[..., 1])
    return inter / (a1 + a2 - inter)


def yolo_loss_components(pred, target, S, B, C, batch_size):
    """Returns (loc, hit, nohit, cls) un-weighted, un-normalised sums.

    pred/target [N,S,S,B*5+C] fp32 torch tensors (pred may require grad).
    """
    N = pred.shape[0]
    obj = target[..., 0] == 1                                    # v1Loss.py:28,44
    # class loss on object cells                                   :33-41
    cls = ((pred[..., 5 * B:][obj] - target[..., 5 * B:][obj]) ** 2).sum()

    conf = pred[..., :B]                                          # :51
    K = int(obj.sum())
    if K > 0:
        pb = pred[..., B:5 * B][obj].reshape(K, B, 4)             # :52,:67
        gb_all = target[..., B:5 * B][obj].reshape(K, B, 4)       # :55,:95
        iou = responsible_iou(pb, gb_all[:, 0, :], S)             # :66-72 (gt slot 0)
        max_iou, max_idx = iou.max(dim=1)                         # :74 first index on ties
        ar = torch.arange(K)
        conf_obj = conf[obj]                                      # [K,B]
        hit = ((conf_obj[ar, max_idx] - max_iou) ** 2).sum()      # :90
        resp = torch.zeros(K, B, dtype=torch.bool)
        resp[ar, max_idx] = True
        # not-responsible slots: every (n,i,j,b) except the K responsible ones  :80,:91
        full_resp = torch.zeros(N, S, S, B, dtype=torch.bool)
        full_resp[obj] = resp
        nohit = (conf[~full_resp] ** 2).sum()
        # location loss (row slicing, T4)                                      :94-101
        pr = pb[ar, max_idx]                                      # [K,4] row-major (n,i,j) order
        gr = gb_all[ar, max_idx]                                  # gt box of the SAME slot
        loc = ((pr[:2] - gr[:2]) ** 2).sum() + ((torch.sqrt(pr[2:]) - torch.sqrt(gr[2:])) ** 2).sum()
    else:
        hit = pred.new_zeros(())
        nohit = (conf ** 2).sum()
        loc = pred.new_zeros(())
    return loc, hit, nohit, cls


def yolo_loss(pred, target, S, B, C, l_coord=5.0, l_noobj=0.5, batch_size=None):
    """total = (l_coord*loc + hit + l_noobj*nohit + cls) / batch_size  (v1Loss.py:104-105).

    Returns (total, (loc, hit, nohit, cls)); the components are the raw sums
    (the reference logs them divided by batch_size, :108).
    """
    if batch_size is None:
        batch_size = pred.shape[0]
    loc, hit, nohit, cls = yolo_loss_components(pred, target, S, B, C, batch_size)
    total = l_coord * loc + hit + l_noobj * nohit + cls
    total = total / batch_size
    return total, (loc, hit, nohit, cls)


def yolo_loss_and_grad(pred, target, S, B, C, l_coord=5.0, l_noobj=0.5, batch_size=None):
    """numpy/torch in -> (loss float32, comps[4] float32, grad_pred ndarray)."""
    p = torch.as_tensor(pred, dtype=torch.float32).clone().requires_grad_(True)
    t = torch.as_tensor(target, dtype=torch.float32)
    total, comps = yolo_loss(p, t, S, B, C, l_coord, l_noobj, batch_size)
    total.backward()
    return (total.detach().numpy(), torch.stack([c.detach() for c in comps]).numpy(),
            p.grad.detach().numpy())
