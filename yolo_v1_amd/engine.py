"""Whole-backbone autograd node and shared executor pieces.

The reference lets autograd record ~500 ATen nodes per step for the backbone
(backbones/OriginResNet.py:173-195, OriginDenseNet.py:114-129).  Here the backbone is ONE
autograd node: its forward runs the HIP kernels layer by layer and keeps the tensors the
backward needs; its backward walks the layers in reverse with explicit dgrad / wgrad / BN-backward
launches and hands every parameter gradient back to autograd in one go, so ``loss.backward()``
and any ``torch.optim`` optimizer keep working unchanged (train.py:170-172).
"""
import os

import torch
import torch.nn as nn

from . import _lib, ops


class ConvParam(nn.Module):
    """Holds one bias-free convolution weight under the key ``<name>.weight`` (what
    conv3x3/conv1x1, backbones/OriginResNet.py:21-29, contribute to the state_dict).  The
    parameter is logically OIHW but stored channels_last, i.e. physically [O][kh][kw][I] -- the
    layout the MFMA kernels and the weight-gradient kernel use."""

    def __init__(self, cin, cout, k, stride=1, pad=0):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.kernel_size, self.stride, self.padding = k, stride, pad
        w = torch.empty(cout, cin, k, k).contiguous(memory_format=torch.channels_last)
        self.weight = nn.Parameter(w)

    def extra_repr(self):
        return "%d, %d, kernel_size=%d, stride=%d, padding=%d, bias=False" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding)


def make_bn(c):
    """nn.BatchNorm2d used as the parameter/buffer container (weight, bias, running_mean,
    running_var, num_batches_tracked keys); its own forward is never called."""
    return nn.BatchNorm2d(c)


class BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, images, *params):
        _lib.require_cuda(images)
        train = net.training
        # activations are only kept for a training-mode forward: eval-mode BatchNorm (running statistics) has no
        # backward here, a later .backward() then fails loudly instead of differentiating the wrong function
        need_bwd = any(ctx.needs_input_grad) and train
        with torch.no_grad():
            pred, saved = net._run_forward(images, train, need_bwd)
        ctx.net = net
        ctx.saved = saved
        ctx.params = params
        return pred

    @staticmethod
    def backward(ctx, gpred):
        net, saved = ctx.net, ctx.saved
        ctx.saved = None
        if saved is None:
            raise _lib.Yv1Error("backward through the HIP backbone needs a training-mode forward with gradients "
                                "enabled (and can run only once per forward)")
        with torch.no_grad():
            grads = net._run_backward(saved, gpred)
        out = []
        for p in ctx.params:
            g = grads.get(p)
            out.append(g)
        return (None, None) + tuple(out)


class HipBackbone(nn.Module):
    """Base: parameter bookkeeping shared by the ResNet and DenseNet executors."""

    def __init__(self):
        super().__init__()
        self._convw = {}      # ConvParam -> ops.ConvWeights
        self._grad_ready_hook = None
        self._phase_boundary = None
        self.phase_boundaries = 2          # how many of the executor's phase boundaries call back (train.GraphedStep sets it)
        self.wgrad_side_stream = os.environ.get("YV1_WGRAD_SIDE_STREAM", "1") != "0"
        # weight gradients of the last N residual blocks of the backward (+ the stem) run on the MAIN stream (see
        # OriginResNet._run_backward); measured on ResNet-50 at batch 64
        self.wgrad_main_tail = int(os.environ.get("YV1_WGRAD_MAIN_TAIL", "0"))
        # weight gradients of the last N blocks of the backward (+ the stem) stay on the side stream but use the split-K width
        # tuned for kernels that run ALONE: the main chain has ended by the time the side stream gets to them
        # (measured, interleaved: 0 / 2 / 3 / 5 / 8 / 12 blocks -> 2909-2912 / 2920-2921 / 2919-2924 / 2918 / 2917-2921 / 2900 img/s)
        self.wgrad_wide_tail = int(os.environ.get("YV1_WGRAD_WIDE_TAIL", "3"))
        # ResNet: bn1's / bn2's reduction pass inside conv2's / conv3's data gradient (ops.conv_dgrad_bn_sums).  Measured on the
        # step, interleaved A/B.  First build: conv2 + conv3 2983 / 2986 -> 2967 / 2982 img/s (slower).  After the epilogue work
        # (per-channel vectors through LDS, operand loads ahead of the LDS staging): conv2 (k_conv_h3 with the epilogue)
        # 3023 / 3023 / 3025 -> 3042 / 3039 / 3039 (+0.5 %): ON; conv3 (its plain form runs the persistent kernel, the fused
        # one the one-tile-per-workgroup kernel) 3026 / 3022 vs 3024 / 3031: level, off.
        self.bn_sums_conv2 = os.environ.get("YV1_BN_SUMS_CONV2", "1") != "0"
        self.bn_sums_conv3 = os.environ.get("YV1_BN_SUMS_CONV3", "0") == "1"
        self.bn_dual = os.environ.get("YV1_BN_DUAL", "1") != "0"         # projection blocks: bn3 + downsample BN backward in one pass
        self.fused_eval = os.environ.get("YV1_FUSED_EVAL", "1") != "0"   # eval(): BatchNorm folded into the conv epilogue
        # training: run the forward convolutions on the fp8 (e4m3) MFMA path -- "fp8 forward GEMMs, bf16 backward"
        # (BASELINE config 5); off by default: the headline configuration computes in bf16
        self.fp8_forward = os.environ.get("YV1_FP8_FORWARD", "0") == "1"
        self._convw8 = {}     # ConvParam -> ops.Fp8Weights

    def set_grad_ready_hook(self, fn):
        """``fn([(param, grad), ...])`` is called from inside the backward executor as soon as the
        kernels producing those parameter gradients have been launched (used by distributed.GradSync
        to overlap the all-reduce with the rest of the backward)."""
        self._grad_ready_hook = fn

    def set_phase_boundary(self, fn):
        """``fn(grads_so_far)`` is called from inside the backward executor at each of its first ``phase_boundaries``
        phase boundaries: points where a parameter-heavy stage is done (ResNet: after layer4 -- head, layer5, layer4 =
        79 % of the gradient bytes after ~10 % of the backward time -- and after layer3) and every kernel launched so
        far has been joined back onto the main stream.  train.GraphedStep ends a hipGraph there, so the RCCL all-reduce
        of the gradients finished since the previous boundary runs beside the replay of the next graph."""
        self._phase_boundary = fn

    def _emit(self, grads, params):
        if self._grad_ready_hook is not None:
            self._grad_ready_hook([(p, grads[p]) for p in params if p in grads])

    def cw(self, conv, **kw):
        w = self._convw.get(conv)
        if w is None or w.param is not conv.weight:
            w = ops.ConvWeights(conv.weight, conv.kernel_size, conv.stride, conv.padding, **kw)
            self._convw[conv] = w
        w.refresh()
        return w

    def cw8(self, conv):
        w = self._convw8.get(conv)
        if w is None or w.param is not conv.weight:
            w = ops.Fp8Weights(conv.weight, conv.kernel_size, conv.stride, conv.padding)
            self._convw8[conv] = w
        return w

    def refresh_all_weights_fp8(self):
        """e4m3 shadows of every convolution with Cin % 64 == 0 (the stem and DenseNet's odd widths stay bf16)."""
        ws = [self.cw8(m) for m in self.modules() if isinstance(m, ConvParam) and m.in_channels % 64 == 0]
        ops.refresh_many_fp8(ws)

    def refresh_all_weights(self):
        """bf16 shadow copies of every convolution weight, refreshed in one multi-tensor launch."""
        ws = []
        for m in self.modules():
            if isinstance(m, ConvParam):
                w = self._convw.get(m)
                if w is None or w.param is not m.weight:
                    w = ops.ConvWeights(m.weight, m.kernel_size, m.stride, m.padding, stem=(m.in_channels == 3))
                    self._convw[m] = w
                ws.append(w)
        ops.refresh_many(ws)

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.Yv1Error("yolo_v1_amd backbones run on the GPU only; there is no CPU fallback")
        params = [p for p in self.parameters()]
        return BackboneFn.apply(self, x, *params)

    def _bump_counters(self, bns):
        # num_batches_tracked += 1 for every BatchNorm that ran in training mode
        torch._foreach_add_([b.num_batches_tracked for b in bns], 1)
