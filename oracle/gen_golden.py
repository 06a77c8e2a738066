"""Generate tests/golden/* by running the REFERENCE's own modules on CPU.

Run in the build container only (``python -m oracle.gen_golden``); needs
/root/reference.  The reference is imported with stub modules for the
packages that are not installed (cv2, imgaug, torchvision -- none of them is
touched by the functions exercised here).  Output is data only: inputs and the
reference's outputs.  Nothing from the reference's source text is written.
"""
import json
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    for m in ("imgaug", "imgaug.augmenters", "torchvision", "torchvision.transforms"):
        sys.modules.setdefault(m, mock.MagicMock())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import utils.utils as ru
    import v1Loss as rl
    import utils.YOLODataLoader as rd
    from backbones import OriginResNet as rr
    from backbones import OriginDenseNet as rdn
    return ru, rl, rd, rr, rdn


class _FProxy:
    """Stands in for ``torch.nn.functional`` inside v1Loss to record the five
    ``mse_loss`` sums in call order: cls, hit, nohit, loc(rows<2), loc(rows>=2)."""

    def __init__(self, real):
        self._real = real
        self.vals = []

    def __getattr__(self, k):
        return getattr(self._real, k)

    def mse_loss(self, *a, **kw):
        v = self._real.mse_loss(*a, **kw)
        self.vals.append(float(v.detach()))
        return v


def _ref_encoder(rd, S):
    ds = rd.yoloDataset.__new__(rd.yoloDataset)
    ds.S, ds.B, ds.C = S, 2, 20
    return ds.encoder


def gen_loss(ru, rl, rd):
    rng = np.random.RandomState(7)
    cases = []

    def rand_target(N, S, K, same_cell=False):
        enc = _ref_encoder(rd, S)
        tg = np.zeros((N, S, S, 30), np.float32)
        left = K
        for n in range(N):
            k = left if n == N - 1 else min(left, max(0, K // N + (1 if n < K % N else 0)))
            left -= k
            if k == 0:
                continue
            cxcy = rng.uniform(0.02, 0.98, size=(k, 2))
            if same_cell and k >= 2:
                cxcy[1] = cxcy[0] + 0.001
            wh = rng.uniform(0.05, 0.9, size=(k, 2))
            lab = rng.randint(0, 20, size=(k,))
            tg[n] = enc(torch.tensor(np.concatenate([cxcy, wh], 1), dtype=torch.float32),
                        torch.tensor(lab)).numpy()
        return tg

    specs = []
    for S in (7, 14):
        for N, K in ((1, 0), (1, 1), (1, 2), (2, 3), (2, 7), (5, 12), (5, 1)):
            specs.append((S, N, K, N, "plain"))
    specs += [(7, 2, 5, 4, "partial_batch"), (14, 3, 6, 16, "partial_batch"),
              (7, 2, 4, 2, "tie_iou"), (7, 2, 4, 2, "slot1_gt_differs"), (7, 1, 3, 1, "no_overlap"),
              (14, 2, 30, 2, "dense")]
    for S, N, K, bs, kind in specs:
        tg = rand_target(N, S, K)
        pred = rng.uniform(0.02, 0.98, size=(N, S, S, 30)).astype(np.float32)
        if kind == "tie_iou":          # both predicted boxes identical in every object cell
            pred[..., 6:10] = pred[..., 2:6]
        if kind == "slot1_gt_differs":  # hand-made target whose two gt slots differ
            obj = tg[..., 0] == 1
            tg[..., 6:10][obj] = rng.uniform(0.1, 0.9, size=(int(obj.sum()), 4)).astype(np.float32)
        if kind == "no_overlap":       # tiny far-away predicted boxes -> IoU 0 for both, argmax 0
            pred[..., 4:6] = 0.01
            pred[..., 8:10] = 0.01
            pred[..., 2:4] = 0.99
            pred[..., 6:8] = 0.99
            tg[..., 2:4][tg[..., 0] == 1] = 0.01
            tg[..., 4:6][tg[..., 0] == 1] = 0.05
        p = torch.tensor(pred, requires_grad=True)
        t = torch.tensor(tg)
        proxy = _FProxy(torch.nn.functional)
        rl.F = proxy
        layer = rl.YOLOLossV1(bs, S, 2, 20, 5.0, 0.5, _device="cpu", _logger=mock.MagicMock())
        loss = layer(p, t)
        loss.backward()
        rl.F = torch.nn.functional
        cls, hit, nohit, la, lb = proxy.vals
        cases.append(dict(S=S, N=N, bs=bs, kind=kind, pred=pred, target=tg,
                          loss=np.float32(loss.item()),
                          comps=np.array([la + lb, hit, nohit, cls], np.float64),
                          grad=p.grad.numpy().copy()))
    # non-contiguous (permuted) pred, as the backbones return it (OriginResNet.py:189)
    out = {"n": len(cases)}
    for i, c in enumerate(cases):
        for k, v in c.items():
            out["c%d_%s" % (i, k)] = v
    np.savez_compressed(os.path.join(OUT, "loss_cases.npz"), **out)
    print("loss cases:", len(cases))


def gen_iou(ru):
    b1 = torch.tensor([[10, 20, 100, 123], [200, 300, 300, 350]], dtype=torch.float32)
    b2 = torch.tensor([[50, 60, 150, 120], [0, 10, 123, 150], [170, 190, 310, 400]], dtype=torch.float32)
    rng = np.random.RandomState(3)
    r1 = rng.uniform(0, 1, size=(17, 4)).astype(np.float32)
    r2 = rng.uniform(0, 1, size=(9, 4)).astype(np.float32)
    r1[:, 2:] += r1[:, :2]
    r2[:, 2:] += r2[:, :2]
    r2[3, 2:] = r2[3, :2] - 0.1       # inverted box -> negative w/h
    cx = rng.uniform(0, 1, size=(6, 4)).astype(np.float32)
    np.savez(os.path.join(OUT, "iou_cases.npz"),
             known_b1=b1.numpy(), known_b2=b2.numpy(), known_iou=ru.compute_iou_matrix(b1, b2).numpy(),
             r1=r1, r2=r2, r_iou=ru.compute_iou_matrix(torch.tensor(r1), torch.tensor(r2)).numpy(),
             cx=cx, cx7=ru.convert_CxCyWH_to_X1Y1X2Y2(torch.tensor(cx), 7, 2, "cpu").numpy(),
             cx14=ru.convert_CxCyWH_to_X1Y1X2Y2(torch.tensor(cx), 14, 2, "cpu").numpy())


def gen_encoder(rd):
    out = {}
    rng = np.random.RandomState(11)
    i = 0
    for S in (7, 14):
        enc = _ref_encoder(rd, S)
        for k in (0, 1, 4, 9):
            boxes = np.concatenate([rng.uniform(0.01, 1.0, size=(k, 2)), rng.uniform(0.02, 0.9, size=(k, 2))], 1)
            boxes = boxes.astype(np.float32)
            if k >= 4:
                boxes[1, :2] = boxes[0, :2]            # same cell: last writer wins
                boxes[2, :2] = [1.0, 1.0]              # right/bottom border
                boxes[3, :2] = [1.0 / S, 2.0 / S]      # exactly on a cell edge
            lab = rng.randint(0, 20, size=(k,))
            t = enc(torch.tensor(boxes).reshape(-1, 4), torch.tensor(lab))
            out["c%d_S" % i] = S
            out["c%d_boxes" % i] = boxes
            out["c%d_labels" % i] = lab
            out["c%d_target" % i] = t.numpy()
            i += 1
    out["n"] = i
    np.savez_compressed(os.path.join(OUT, "encoder_cases.npz"), **out)


def _cluster_boxes(rng, n_clusters, sizes, jitter=0.01):
    b, s = [], []
    for c in range(n_clusters):
        cx, cy = rng.uniform(0.1, 0.9, size=2)
        w, h = rng.uniform(0.05, 0.12, size=2)
        for _ in range(sizes[c]):
            d = rng.uniform(-jitter, jitter, size=4)
            b.append([cx - w / 2 + d[0], cy - h / 2 + d[1], cx + w / 2 + d[2], cy + h / 2 + d[3]])
            s.append(rng.uniform(0.01, 1.0))
    return np.asarray(b, np.float32), np.asarray(s, np.float32)


def gen_nms(ru):
    """Unmodified reference ``nms`` on inputs where it does not reach its
    0-dim ``squeeze`` crash; crashing inputs are skipped."""
    rng = np.random.RandomState(5)
    out, acc, tried = {}, 0, 0
    wanted_n = [1, 2, 3, 5, 8, 13, 21, 34, 55, 64, 65, 98, 128, 200, 256, 300, 392]
    while acc < 80 and tried < 20000:
        tried += 1
        mode = tried % 3
        thr = float(rng.choice([0.25, 0.45, 0.5, 1.0]))
        if mode == 0:                      # clusters of >=2 heavily overlapping boxes
            target_n = int(rng.choice(wanted_n))
            if target_n == 1:
                sizes = [1]
            else:
                sizes = []
                left = target_n
                while left > 0:
                    k = min(left, int(rng.randint(2, 7)))
                    if left - k == 1:
                        k += 1
                    sizes.append(k)
                    left -= k
            b, s = _cluster_boxes(rng, len(sizes), sizes)
        elif mode == 1:                    # random boxes, small n
            n = int(rng.randint(1, 12))
            xy = rng.uniform(0, 0.8, size=(n, 2))
            wh = rng.uniform(0.05, 0.5, size=(n, 2))
            b = np.concatenate([xy, xy + wh], 1).astype(np.float32)
            s = rng.uniform(0.01, 1, size=n).astype(np.float32)
        else:                              # decoder-shaped: cells with two near-identical boxes
            S = int(rng.choice([7, 14]))
            n = S * S * 2
            b, s = _cluster_boxes(rng, S * S, [2] * (S * S), jitter=0.004)
        if len(np.unique(s)) != len(s):
            continue
        try:
            keep = ru.nms(torch.tensor(b), torch.tensor(s), thr)
        except IndexError:
            continue
        out["c%d_boxes" % acc] = b
        out["c%d_scores" % acc] = s
        out["c%d_thr" % acc] = np.float64(thr)
        out["c%d_keep" % acc] = keep.numpy().astype(np.int64)
        acc += 1
    out["n"] = acc
    np.savez_compressed(os.path.join(OUT, "nms_cases.npz"), **out)
    print("nms cases accepted:", acc, "of", tried, "n:", sorted({len(out['c%d_scores' % i]) for i in range(acc)}))


def gen_decoder(ru, rd):
    """Reference ``decoder`` with its ``nms`` call swapped for keep-all (decode
    stage only), plus the full decoder where the unmodified nms survives."""
    rng = np.random.RandomState(9)
    out, i = {}, 0
    real_nms = ru.nms
    for S in (7, 14):
        for kind in ("uniform", "sparse", "allzero", "encoded_gt", "one_hot_max"):
            pred = rng.uniform(0.0, 1.0, size=(1, S, S, 30)).astype(np.float32)
            thresh, nms_th = 0.1, 0.5
            if kind == "sparse":
                pred[..., :2] *= (rng.uniform(size=(1, S, S, 2)) > 0.8)
                pred[..., :2] *= 0.5
                thresh, nms_th = 0.005, 0.45
            if kind == "allzero":
                pred[...] = 0
            if kind == "encoded_gt":
                enc = _ref_encoder(rd, S)
                boxes = np.concatenate([rng.uniform(0.05, 0.95, size=(4, 2)), rng.uniform(0.1, 0.5, size=(4, 2))], 1)
                pred = enc(torch.tensor(boxes, dtype=torch.float32), torch.tensor(rng.randint(0, 20, size=4))).numpy()[None]
                thresh = 0.3
            if kind == "one_hot_max":   # every conf below 1e-4 except the max -> mask2 path (utils.py:113)
                pred[..., :2] = 1e-5
                pred[0, S // 2, 1, 1] = 9e-5
                thresh = 0.0
            ru.nms = lambda b, s, t: torch.arange(len(s))
            bx, cl, pr = ru.decoder(torch.tensor(pred.copy()), grid_num=S, thresh=thresh, nms_th=nms_th)
            ru.nms = real_nms
            out["c%d_S" % i] = S
            out["c%d_kind" % i] = kind
            out["c%d_pred" % i] = pred
            out["c%d_thresh" % i] = np.float64(thresh)
            out["c%d_nms_th" % i] = np.float64(nms_th)
            out["c%d_cand_boxes" % i] = bx.numpy()
            out["c%d_cand_cls" % i] = cl.numpy().astype(np.int64)
            out["c%d_cand_probs" % i] = pr.numpy()
            try:
                fb, fc, fp = ru.decoder(torch.tensor(pred.copy()), grid_num=S, thresh=thresh, nms_th=nms_th)
                out["c%d_full_ok" % i] = 1
                out["c%d_full_boxes" % i] = fb.numpy()
                out["c%d_full_cls" % i] = fc.numpy().astype(np.int64)
                out["c%d_full_probs" % i] = fp.numpy()
            except IndexError:
                out["c%d_full_ok" % i] = 0
            i += 1
    out["n"] = i
    np.savez_compressed(os.path.join(OUT, "decoder_cases.npz"), **out)
    print("decoder cases:", i, "full ok:", sum(int(out["c%d_full_ok" % k]) for k in range(i)))


def gen_voc(ru):
    logger = mock.MagicMock()
    preds = {'cat': [['image01', 0.9, 20, 20, 40, 40], ['image01', 0.8, 20, 20, 50, 50], ['image02', 0.8, 30, 30, 50, 50]],
             'dog': [['image01', 0.78, 60, 60, 90, 90]]}
    target = {('image01', 'cat'): [[20, 20, 41, 41]], ('image01', 'dog'): [[60, 60, 91, 91]],
              ('image02', 'cat'): [[30, 30, 51, 51]]}
    import copy
    m = ru.voc_eval(copy.deepcopy(preds), copy.deepcopy(target), VOC_CLASSES=['cat', 'dog'], logger=logger)
    # zero-detection quirk: class 'bird' has no preds -> ap -1 then break (utils.py:248-255)
    from collections import defaultdict
    dpreds = defaultdict(list, copy.deepcopy(preds))      # run_test_mAP builds preds as defaultdict(list), utils.py:390
    m2 = ru.voc_eval(dpreds, copy.deepcopy(target), VOC_CLASSES=['cat', 'bird', 'dog'], logger=logger)
    rec = np.array([0.2, 0.2, 0.4, 0.6, 0.6, 1.0])
    prec = np.array([1.0, 0.5, 0.66, 0.75, 0.6, 0.5])
    json.dump({"preds": preds, "target": {"%s|%s" % k: v for k, v in target.items()},
               "classes": ["cat", "dog"], "mAP": m, "classes_quirk": ["cat", "bird", "dog"], "mAP_quirk": m2,
               "rec": rec.tolist(), "prec": prec.tolist(),
               "ap_area": float(ru.voc_ap(rec, prec, False)), "ap_07": float(ru.voc_ap(rec, prec, True))},
              open(os.path.join(OUT, "voc_eval_known.json"), "w"), indent=1)
    print("voc known mAP", m, m2)


def _sd_np(mod):
    return {k: v.detach().numpy().copy() for k, v in mod.state_dict().items()}


def gen_blocks(rr, rdn):
    """Tiny-width instances of the reference's own block classes: fwd + bwd."""
    torch.manual_seed(0)
    out = {}

    def run(name, mod, x, extra=None):
        mod.train()
        for m in mod.modules():           # non-trivial affine + running stats
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.uniform_(-0.3, 0.3)
        sd0 = _sd_np(mod)
        x = x.clone().requires_grad_(True)
        y = mod(x) if extra is None else extra(mod, x)
        gy = torch.randn_like(y)
        y.backward(gy)
        out[name + "/x"] = x.detach().numpy()
        out[name + "/y"] = y.detach().numpy()
        out[name + "/gy"] = gy.numpy()
        out[name + "/gx"] = x.grad.numpy()
        for k, v in sd0.items():
            out[name + "/p/" + k] = v
        for k, v in _sd_np(mod).items():
            if "running" in k:
                out[name + "/after/" + k] = v
        for k, p in mod.named_parameters():
            out[name + "/g/" + k] = p.grad.numpy()

    ds = lambda i, o, s: torch.nn.Sequential(rr.conv1x1(i, o, s), torch.nn.BatchNorm2d(o))
    run("bneck_s1_ds", rr.Bottleneck(32, 16, 1, ds(32, 64, 1)), torch.randn(2, 32, 8, 8))
    run("bneck_s2_ds", rr.Bottleneck(64, 32, 2, ds(64, 128, 2)), torch.randn(2, 64, 8, 8))
    run("bneck_plain", rr.Bottleneck(64, 16), torch.randn(2, 64, 6, 6))
    run("dense_layer", rdn._DenseLayer(64, 32, 4, 0), torch.randn(2, 64, 8, 8))
    run("transition", rdn._Transition(64, 32), torch.randn(2, 64, 8, 8))
    # round 2: the same stride-2 projection block on a 16x16 input (its BatchNorms after the stride then see 128 samples
    # per channel instead of 32: gate flips of the bf16 path no longer dominate its gradients) and an identity block at
    # widths the HIP kernels take without zero-padding
    run("bneck_s2_ds_16", rr.Bottleneck(64, 32, 2, ds(64, 128, 2)), torch.randn(2, 64, 16, 16))
    run("bneck_plain_32", rr.Bottleneck(128, 32), torch.randn(2, 128, 12, 12))
    np.savez_compressed(os.path.join(OUT, "block_cases.npz"), **out)
    print("block fixtures:", len(out), "arrays")


def gen_wholenet(rr, rdn):
    """state_dict key/shape inventory + whole-net forward on a 64x64 / 128x128
    input with weights from oracle.backbones.init_params(seed) (regenerated by
    the test, not stored)."""
    from oracle import backbones as ob
    inv, outs = {}, {}
    for kind, ctor, shapes_fn in (("resnet", rr.resnet50, ob.resnet50_param_shapes),
                                  ("densenet", rdn.densenet121, ob.densenet121_param_shapes)):
        for S in (7, 14):
            m = ctor(S=S)
            sd = m.state_dict()
            inv["%s_S%d" % (kind, S)] = [[k, list(v.shape)] for k, v in sd.items()]
            P = ob.init_params(shapes_fn(S), kind, seed=S)
            m.load_state_dict(P)
            m.train()
            hw = 128
            x = torch.randn(2, 3, hw, hw, generator=torch.Generator().manual_seed(100 + S))
            with torch.no_grad():
                y = m(x)
            outs["%s_S%d_y" % (kind, S)] = y.contiguous().numpy()
            outs["%s_S%d_bn_end_rm" % (kind, S)] = m.state_dict()["bn_end.running_mean"].numpy().copy()
    json.dump({"torch": torch.__version__, "inventory": inv}, open(os.path.join(OUT, "state_dict_keys.json"), "w"))
    np.savez_compressed(os.path.join(OUT, "wholenet_fwd.npz"), **outs)
    print("whole-net forward fixtures done")


def gen_train_steps(rl, rr):
    """a11: three iterations of the loop body train.py:155-172 around the
    reference's modules (ResNet-50 S=7, N=2, 448x448), weights from
    oracle.backbones.init_params(seed 0)."""
    from oracle import backbones as ob
    from oracle import train_step as ots
    res = {"torch": torch.__version__}
    for tag, epoch, lr_map in (("warmup", 0, {}), ("epoch1", 1, {1: 0.001})):
        m = rr.resnet50(S=7)
        m.load_state_dict(ob.init_params(ob.resnet50_param_shapes(7), "resnet", seed=0))
        m.train()
        images, target = ots.synthetic_batch(2, 7)
        opt = torch.optim.SGD(m.parameters(), lr=0.0, momentum=0.99)
        proxy = _FProxy(torch.nn.functional)
        rl.F = proxy
        layer = rl.YOLOLossV1(2, 7, 2, 20, 5.0, 0.5, _device="cpu", _logger=mock.MagicMock())
        lr, it, steps = 0.0, 0, []
        for _ in range(3):
            it += 1
            lr = ots.learning_rate_policy(it, epoch, lr, lr_map)
            for g in opt.param_groups:
                g["lr"] = lr
            proxy.vals.clear()
            pred = m(images)
            loss = layer(pred, target)
            opt.zero_grad()
            loss.backward()
            opt.step()
            cls, hit, nohit, la, lb = proxy.vals
            steps.append({"loss": float(loss.item()), "comps": [la + lb, hit, nohit, cls], "lr": lr})
        rl.F = torch.nn.functional
        res[tag] = steps
        print(tag, steps)
    json.dump(res, open(os.path.join(OUT, "train_steps.json"), "w"), indent=1)


def main():
    os.makedirs(OUT, exist_ok=True)
    ru, rl, rd, rr, rdn = _import_reference()
    which = set(sys.argv[1:]) or {"loss", "iou", "encoder", "nms", "decoder", "voc", "blocks", "wholenet", "train"}
    if "loss" in which:
        gen_loss(ru, rl, rd)
    if "iou" in which:
        gen_iou(ru)
    if "encoder" in which:
        gen_encoder(rd)
    if "nms" in which:
        gen_nms(ru)
    if "decoder" in which:
        gen_decoder(ru, rd)
    if "voc" in which:
        gen_voc(ru)
    if "blocks" in which:
        gen_blocks(rr, rdn)
    if "wholenet" in which:
        gen_wholenet(rr, rdn)
    if "train" in which:
        gen_train_steps(rl, rr)


if __name__ == "__main__":
    main()
