"""Fused SGD-with-momentum for the HIP training path (drop-in ``torch.optim.Optimizer``).

Same update as ``torch.optim.SGD(lr, momentum, dampening=0, nesterov=False, weight_decay=0)`` -- what
the reference builds at train.py:84 -- in one pass over memory (csrc/optim.hip).  The learning rate
lives in a device scalar: writing ``param_group['lr']`` every iteration (train.py:158-160) only
updates that scalar, so a captured hipGraph of the whole training step stays valid.
"""
import ctypes

import torch

from . import _lib, ops


def _same_memory_order(a, b):
    """True when two dense tensors of one shape enumerate their elements in the same memory order
    (strides of size-1 dimensions do not matter)."""
    return all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=0.0, momentum=0.0):
        if momentum < 0 or lr < 0:
            raise ValueError("invalid lr/momentum")
        super().__init__(params, dict(lr=lr, momentum=momentum))
        self._lr_dev = {}
        self._lr_host = {}
        self.grad_scale = 1.0

    def _lr_tensor(self, gi, group, device):
        t = self._lr_dev.get(gi)
        if t is None or t.device != device:
            t = torch.zeros((), dtype=torch.float32, device=device)
            self._lr_dev[gi] = t
            self._lr_host[gi] = None
        lr = float(group['lr'])
        if self._lr_host[gi] != lr:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.Yv1Error("set the learning rate with FusedSGD.set_lr() before replaying a captured step")
            t.fill_(lr)
            self._lr_host[gi] = lr
        return t

    def set_lr(self, lr):
        """Updates the device-side learning rate of every group (safe between graph replays)."""
        for gi, group in enumerate(self.param_groups):
            group['lr'] = lr
            if gi in self._lr_dev and self._lr_host.get(gi) != float(lr):
                self._lr_dev[gi].fill_(float(lr))
                self._lr_host[gi] = float(lr)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        maxn = L.yv1_sgd_max_tensors()
        for gi, group in enumerate(self.param_groups):
            ws, gs, ms, ns = [], [], [], []
            keep = []                    # temporaries must outlive the launches below
            dev = None
            for p in group['params']:
                if p.grad is None:
                    continue
                _lib.require_cuda(p)
                dev = p.device
                g = p.grad
                if g.dtype != torch.float32 or not _same_memory_order(g, p):
                    g2 = torch.empty_like(p)
                    g2.copy_(g)
                    g = g2
                    keep.append(g2)
                st = self.state[p]
                if 'momentum_buffer' not in st:
                    st['momentum_buffer'] = torch.zeros_like(p)       # preserves the parameter's memory order
                    if not _same_memory_order(st['momentum_buffer'], p):
                        raise _lib.Yv1Error("FusedSGD needs dense parameters")
                ws.append(p.data_ptr()); gs.append(g.data_ptr()); ms.append(st['momentum_buffer'].data_ptr())
                ns.append(p.numel())
            if not ws:
                continue
            lr_t = self._lr_tensor(gi, group, dev)
            stream = _lib.stream_ptr(dev)
            for i in range(0, len(ws), maxn):
                k = min(maxn, len(ws) - i)
                PA = ctypes.c_void_p * k
                NA = ctypes.c_longlong * k
                _lib.check(L.yv1_sgd_momentum_step(PA(*ws[i:i + k]), PA(*gs[i:i + k]), PA(*ms[i:i + k]), NA(*ns[i:i + k]), k,
                                                   _lib.ptr(lr_t), float(group['momentum']), float(self.grad_scale), stream),
                           "yv1_sgd_momentum_step")
        del keep
        ops.bump_weight_epoch()      # the kernel wrote the parameters behind torch's back (no _version bump)
        return loss
