"""Drop-in for the reference's ``v1Loss`` module (v1Loss.py:9-118) on MI355X.

``YOLOLossV1`` keeps the reference's constructor and ``forward(pred, target)``
contract, but the whole forward *and* its backward are one fused HIP launch
sequence (csrc/loss.hip) instead of a Python loop over object cells with ~25
tiny tensor ops and several host syncs each.  The four component values the
reference logs with 4 ``.item()`` syncs (v1Loss.py:107-116) come back in one
small device buffer and cost one D2H copy.
"""
import torch
import torch.nn as nn

from . import _lib


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, S, B, C, l_coord, l_noobj, batch_size, comps_out):
        _lib.require_cuda(pred, target)
        if pred.dtype != torch.float32:
            pred = pred.float()
        target = target.to(dtype=torch.float32).contiguous()
        N = pred.shape[0]
        D = B * 5 + C
        if tuple(pred.shape) != (N, S, S, D) or tuple(target.shape) != (N, S, S, D):
            raise _lib.Yv1Error("loss expects [N,%d,%d,%d] tensors, got %s and %s"
                                % (S, S, D, tuple(pred.shape), tuple(target.shape)))
        L = _lib.lib()
        dev = pred.device
        ws_bytes = L.yv1_loss_workspace_bytes(N, S)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty((N, S, S, D), dtype=torch.float32, device=dev) if need_grad else None
        st = pred.stride()
        _lib.check(L.yv1_loss_fwd_bwd(_lib.ptr(pred), st[0], st[1], st[2], st[3], _lib.ptr(target), N, S, B, C,
                                      l_coord, l_noobj, float(batch_size), _lib.ptr(loss), _lib.ptr(comps_out),
                                      _lib.ptr(grad), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev)),
                   "yv1_loss_fwd_bwd")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grad
        ctx.grad = None
        if g is None:
            return (None,) * 9
        L = _lib.lib()
        gout = gout.to(dtype=torch.float32).contiguous()
        _lib.check(L.yv1_scale_by_device_scalar(_lib.ptr(g), _lib.ptr(gout), g.numel(), _lib.stream_ptr(g.device)),
                   "yv1_scale_by_device_scalar")
        return (g,) + (None,) * 8


class YOLOLossV1(nn.Module):
    """Same signature as reference v1Loss.py:10; extra keyword ``_quiet`` skips the
    per-call logging (and with it the only host sync)."""

    def __init__(self, _batch_size, _S, _B, _clsN, _l_coord=5., _l_noobj=0.5, _device='cuda:0', _logger=None,
                 _vis=None, _quiet=False):
        super().__init__()
        self.S = _S
        self.B = _B
        self.device = _device
        self.C = _clsN
        self.lambda_coord = _l_coord
        self.lambda_noobj = _l_noobj
        self.batch_size = _batch_size
        self.logger = _logger
        self.vis = _vis
        self.quiet = _quiet
        self.last_components = None   # device tensor [4]: location, contain, not-contain, classify (raw sums)

    def loss_and_grad(self, pred_tensor, target_tensor):
        """(total loss, d total / d pred) from the one fused launch, without an autograd graph -- what
        ``forward`` + ``backward`` produce, for executors that drive the backward pass themselves
        (train.GraphedStep with several ranks)."""
        _lib.require_cuda(pred_tensor, target_tensor)
        pred = pred_tensor.detach()
        if pred.dtype != torch.float32:
            pred = pred.float()
        target = target_tensor.to(dtype=torch.float32).contiguous()
        N, D = pred.shape[0], self.B * 5 + self.C
        if tuple(pred.shape) != (N, self.S, self.S, D) or tuple(target.shape) != (N, self.S, self.S, D):
            raise _lib.Yv1Error("loss expects [N,%d,%d,%d] tensors, got %s and %s"
                                % (self.S, self.S, D, tuple(pred.shape), tuple(target.shape)))
        L = _lib.lib()
        dev = pred.device
        ws_bytes = L.yv1_loss_workspace_bytes(N, self.S)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        comps = torch.empty(4, dtype=torch.float32, device=dev)
        grad = torch.empty((N, self.S, self.S, D), dtype=torch.float32, device=dev)
        st = pred.stride()
        _lib.check(L.yv1_loss_fwd_bwd(_lib.ptr(pred), st[0], st[1], st[2], st[3], _lib.ptr(target), N, self.S, self.B,
                                      self.C, float(self.lambda_coord), float(self.lambda_noobj), float(self.batch_size),
                                      _lib.ptr(loss), _lib.ptr(comps), _lib.ptr(grad), _lib.ptr(ws), ws_bytes,
                                      _lib.stream_ptr(dev)), "yv1_loss_fwd_bwd")
        self.last_components = comps
        return loss, grad

    def forward(self, pred_tensor, target_tensor):
        comps = torch.empty(4, dtype=torch.float32, device=pred_tensor.device)
        total = _LossFn.apply(pred_tensor, target_tensor, self.S, self.B, self.C, float(self.lambda_coord),
                              float(self.lambda_noobj), self.batch_size, comps)
        self.last_components = comps
        if not self.quiet:
            loc, hit, nohit, cls = (comps / self.batch_size).tolist()      # one D2H copy
            msg = 'location loss : %.5f contain loss : %.5f not contain loss: %.5f classify loss : %.5f' % (
                loc, hit, nohit, cls)                                      # format of v1Loss.py:108
            if self.logger:
                self.logger.info(msg)
            else:
                print('location loss : %.5f' % loc, 'contain loss : %.5f' % hit, 'not contain loss: %.5f' % nohit,
                      'classify loss : %.5f' % cls)                        # v1Loss.py:110
            if self.vis:
                self.vis.plot('location loss', loc)                        # v1Loss.py:113-116
                self.vis.plot('confidence loss', hit)
                self.vis.plot('no object loss', nohit)
                self.vis.plot('classify loss', cls)
        return total


yoloLoss = YOLOLossV1   # the name BASELINE.json / testCodes/tensor_test.py:9 use
