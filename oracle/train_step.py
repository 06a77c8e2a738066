"""Oracle (test infrastructure only): the training-step glue, torch-CPU fp32.

Restates the loop body of reference train.py:155-172 (the file itself does
not parse -- unresolved merge conflicts at :49-53,:136-141,:193-197 -- so the
order of operations is taken from the text): per-iteration LR write
(:157-160, policy :22-32), forward (:166), loss (:167), ``zero_grad``,
``backward``, SGD(momentum 0.99, no weight decay) ``step`` (:84, :170-172).
Also provides the synthetic VOC-shaped batch of SURVEY.md section 8d and is
the ``cpu_baseline`` ("port") leg of bench.py.
"""
import time

import numpy as np
import torch

from . import backbones as ob
from . import boxes as obx
from . import loss as ol


def warmming_up_policy(now_iter, now_lr, stop_down_iter=1000):
    """train.py:22-25 (+1e-6 per iteration for the first 1000 iterations)."""
    if now_iter <= stop_down_iter:
        now_lr += 0.000001
    return now_lr


def learning_rate_policy(now_iter, now_epoch, now_lr, lr_adjust_map, stop_down_iter=1000):
    """train.py:27-32."""
    now_lr = warmming_up_policy(now_iter, now_lr, stop_down_iter)
    if now_epoch in lr_adjust_map:
        now_lr = lr_adjust_map[now_epoch]
    return now_lr


def synthetic_batch(N, S, B=2, C=20, img_seed=1234, tgt_seed=4321, hw=448, objs=3):
    """images randn(N,3,hw,hw) (seed img_seed); targets ``objs``
    objects per image, cx,cy~U(0,1), w,h~U(0.05,0.9), class~U{0..C-1}
    (seed tgt_seed), encoded with the encoder restatement."""
    g = torch.Generator().manual_seed(img_seed)
    images = torch.randn(N, 3, hw, hw, generator=g)
    rng = np.random.RandomState(tgt_seed)
    tg = np.zeros((N, S, S, B * 5 + C), np.float32)
    for n in range(N):
        cxcy = rng.uniform(0.0, 1.0, size=(objs, 2))
        cxcy = np.clip(cxcy, 1e-3, 1.0)            # cx=0 would give cell -1
        wh = rng.uniform(0.05, 0.9, size=(objs, 2))
        lab = rng.randint(0, C, size=(objs,))
        tg[n] = obx.encode_target(np.concatenate([cxcy, wh], 1).astype(np.float32), lab, S, B, C)
    return images, torch.from_numpy(tg)


class SGDMomentum:
    """torch.optim.SGD(momentum=m, dampening=0, nesterov=False, wd=0) restated:
    buf = g (first step) else m*buf + g ;  p -= lr*buf."""

    def __init__(self, params, momentum=0.99):
        self.params = list(params)
        self.momentum = momentum
        self.bufs = [None] * len(self.params)

    def step(self, lr):
        with torch.no_grad():
            for i, p in enumerate(self.params):
                if p.grad is None:
                    continue
                if self.bufs[i] is None:
                    self.bufs[i] = p.grad.clone()
                else:
                    self.bufs[i].mul_(self.momentum).add_(p.grad)
                p.add_(self.bufs[i], alpha=-lr)

    def zero_grad(self):
        for p in self.params:
            p.grad = None


def make_state(kind="resnet", S=7, seed=0):
    shapes = ob.resnet50_param_shapes(S) if kind == "resnet" else ob.densenet121_param_shapes(S)
    P = ob.init_params(shapes, kind, seed)
    for k, v in P.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    return P


def train_steps(P, images, target, S, steps, kind="resnet", B=2, C=20, batch_size=None, lr0=0.0,
                epoch=0, lr_map=None, start_iter=0, timings=None, fwd_kwargs=None, grid=None):
    """Runs ``steps`` iterations of train.py:155-172 on CPU.  Returns a list of
    dicts {loss, comps[4], lr} per step.  ``fwd_kwargs`` reach the backbone restatement (e.g. the bf16-storage
    emulation hook ``q=`` the GPU parity tests use); ``grid`` is the output grid when the images are not 448x448
    (S then only selects the architecture variant, as in the backbone restatement)."""
    fwd = ob.resnet50_forward if kind == "resnet" else ob.densenet121_forward
    lr_map = lr_map or {}
    params = [v for v in P.values() if v.requires_grad]
    opt = SGDMomentum(params, 0.99)
    lr = lr0
    it = start_iter
    out = []
    bs = batch_size or images.shape[0]
    for _ in range(steps):
        t0 = time.perf_counter()
        it += 1
        lr = learning_rate_policy(it, epoch, lr, lr_map)
        pred = fwd(images, P, S, training=True, **(fwd_kwargs or {}))
        total, comps = ol.yolo_loss(pred, target, grid or S, B, C, 5.0, 0.5, bs)
        opt.zero_grad()
        total.backward()
        opt.step(lr)
        if timings is not None:
            timings.append(time.perf_counter() - t0)
        out.append({"loss": float(total.detach()), "comps": [float(c.detach()) for c in comps], "lr": lr})
    return out
