"""Thin host wrappers over the C-ABI kernels (include/yv1.h).

An activation is an NHWC bf16 torch tensor plus a channel window (``Act``): kernels take a base
pointer and a pixel stride, so a DenseNet block's growing feature map is ONE buffer and every
layer writes its 32-channel slice in place -- ``torch.cat`` (OriginDenseNet.py:36) disappears.
torch only owns the memory and the stream.
"""
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class Act:
    """NHWC bf16 activation window: channels [c0, c0+C) of tensor ``t`` [N,H,W,Ctot]."""
    __slots__ = ("t", "c0", "C", "N", "H", "W", "ld", "p")

    def __init__(self, t, c0=0, C=None):
        self.t = t
        self.N, self.H, self.W, self.ld = t.shape
        self.c0 = c0
        self.C = self.ld - c0 if C is None else C
        self.p = t.data_ptr() + 2 * c0

    @property
    def npix(self):
        return self.N * self.H * self.W

    def window(self, c0, C):
        return Act(self.t, self.c0 + c0, C)


def new_act(N, H, W, C, device):
    return Act(torch.empty((N, H, W, C), dtype=torch.bfloat16, device=device))


def _f32(n, device):
    return torch.empty(n, dtype=torch.float32, device=device)


# ------------------------------------------------------------------ weights
_WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    """Tell the bf16 shadow copies that parameters were modified by a raw kernel (fused optimizer,
    graph replay) that did not go through torch's version counter."""
    _WEIGHT_EPOCH[0] += 1


def _new_tr(Ipad, taps, Opad, dev):
    """Transposed weight copy [Ipad][taps][Opad]; pointwise weights get their rows padded to a multiple of 128 with zeros
    (never rewritten: the prep kernels fill the first Ipad rows) so that a data-gradient column tile may reach past Ipad
    (conv_dgrad_bn_deferred)."""
    if taps != 1 or Ipad % 128 == 0:
        return torch.empty((Ipad, taps, Opad), dtype=torch.bfloat16, device=dev)
    return torch.zeros(((Ipad + 127) // 128 * 128, taps, Opad), dtype=torch.bfloat16, device=dev)


class ConvWeights:
    """bf16 shadow copies of one fp32 OIHW parameter, refreshed when the parameter changes:
    ``fwd`` [Opad][taps][Ipad] and ``tr`` [Ipad][taps][Opad] (dgrad operand)."""

    def __init__(self, param, k, stride, pad, need_dgrad=True, stem=False):
        self.param = param
        self.O, self.I = param.shape[0], param.shape[1]
        self.k, self.stride, self.pad = k, stride, pad
        self.stem = stem
        self.Opad = (self.O + 31) // 32 * 32
        self.Ipad = 32 if stem else self.I
        self.need_dgrad = need_dgrad and not stem
        self.version = -1
        self.fwd = None
        self.tr = None

    def refresh(self):
        p = self.param
        ver = (p._version, _WEIGHT_EPOCH[0])
        if self.fwd is not None and ver == self.version and self.fwd.device == p.device:
            return
        dev = p.device
        s = stream_ptr(dev)
        src = p.detach()
        so, si, sh, sw = src.stride()
        if self.stem:
            if self.fwd is None or self.fwd.device != dev:
                self.fwd = torch.empty((self.O, 7, 32), dtype=torch.bfloat16, device=dev)
            check(lib().yv1_prep_stem_weights(ptr(src), so, si, sh, sw, self.O, ptr(self.fwd), s), "yv1_prep_stem_weights")
        else:
            taps = self.k * self.k
            if self.fwd is None or self.fwd.device != dev:
                self.fwd = torch.empty((self.Opad, taps, self.Ipad), dtype=torch.bfloat16, device=dev)
                self.tr = _new_tr(self.Ipad, taps, self.Opad, dev) if self.need_dgrad else None
            check(lib().yv1_prep_weights(ptr(src), so, si, sh, sw, self.O, self.I, self.k, self.k, self.Opad, self.Ipad,
                                         ptr(self.fwd), ptr(self.tr), s), "yv1_prep_weights")
        self.version = ver


def refresh_many(weights):
    """Refreshes every stale ConvWeights of ``weights`` with as few launches as possible (one multi-tensor
    launch per 40 weights instead of one launch per weight)."""
    import ctypes
    stale = []
    for w in weights:
        p = w.param
        ver = (p._version, _WEIGHT_EPOCH[0])
        if w.fwd is not None and ver == w.version and w.fwd.device == p.device:
            continue
        if w.stem:
            w.refresh()
            continue
        dev = p.device
        taps = w.k * w.k
        if w.fwd is None or w.fwd.device != dev:
            w.fwd = torch.empty((w.Opad, taps, w.Ipad), dtype=torch.bfloat16, device=dev)
            w.tr = _new_tr(w.Ipad, taps, w.Opad, dev) if w.need_dgrad else None
        stale.append((w, ver))
    if not stale:
        return
    L = lib()
    maxn = L.yv1_prep_weights_max_tensors()
    dev = stale[0][0].param.device
    for i in range(0, len(stale), maxn):
        chunk = stale[i:i + maxn]
        n = len(chunk)
        PA, IA, LA = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_longlong * (4 * n)
        strides = []
        for w, _ in chunk:
            strides.extend(w.param.stride())
        check(L.yv1_prep_weights_multi(PA(*[w.param.data_ptr() for w, _ in chunk]), LA(*strides),
                                       IA(*[w.O for w, _ in chunk]), IA(*[w.I for w, _ in chunk]),
                                       IA(*[w.k for w, _ in chunk]), IA(*[w.Opad for w, _ in chunk]),
                                       IA(*[w.Ipad for w, _ in chunk]), PA(*[w.fwd.data_ptr() for w, _ in chunk]),
                                       PA(*[(w.tr.data_ptr() if w.tr is not None else None) for w, _ in chunk]), n,
                                       stream_ptr(dev)), "yv1_prep_weights_multi")
    for w, ver in stale:
        w.version = ver


# ------------------------------------------------------------------ convolution
def conv_out_hw(H, W, k, stride, pad):
    return (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1


def conv_fwd(x, w, y, want_stats=True):
    """y = conv(x, w); returns BN-statistic partials [rows][2][Cout_pad] (or None)."""
    dev = x.t.device
    M = y.npix
    stats = None
    if want_stats:
        rows = lib().yv1_conv2d_stats_rows(M, w.Opad, w.Ipad, w.k, w.stride, w.pad)
        stats = _f32(rows * 2 * w.Opad, dev).view(rows, 2, w.Opad)
    check(lib().yv1_conv2d_fwd_nhwc_bf16(x.p, ptr(w.fwd), y.p, x.N, x.H, x.W, x.ld, w.Ipad, w.Opad, y.ld, w.k, w.stride,
                                         w.pad, ptr(stats), stream_ptr(dev)), "yv1_conv2d_fwd_nhwc_bf16")
    return stats


def conv_fwd_bn_act(x, w, y, st, relu=True, residual=None):
    """Inference: y = relu?(bf16(conv(x, w) * scale + shift) + residual), ``st`` an eval-mode BNState."""
    dev = x.t.device
    check(lib().yv1_conv2d_fwd_bn_act_nhwc_bf16(x.p, ptr(w.fwd), y.p, x.N, x.H, x.W, x.ld, w.Ipad, w.Opad, y.ld, w.k, w.stride,
                                                w.pad, ptr(st.scale), ptr(st.shift),
                                                residual.p if residual is not None else None,
                                                residual.ld if residual is not None else 0, 1 if relu else 0,
                                                stream_ptr(dev)), "yv1_conv2d_fwd_bn_act_nhwc_bf16")


def stem_fwd_bn_act(xp, w, y, H, W, st, relu=True):
    dev = xp.device
    check(lib().yv1_conv2d_stem_fwd_bn_act_bf16(ptr(xp), ptr(w.fwd), y.p, y.N, H, W, w.O, y.ld, ptr(st.scale), ptr(st.shift),
                                                1 if relu else 0, stream_ptr(dev)), "yv1_conv2d_stem_fwd_bn_act_bf16")


# ------------------------------------------------------------------ fp8 forward GEMMs (training, BASELINE config 5)
class Fp8Act:
    """NHWC e4m3 activation [N,H,W,C] held as uint8 (scale 1)."""
    __slots__ = ("t", "N", "H", "W", "C", "ld", "p")

    def __init__(self, N, H, W, C, device):
        self.t = torch.empty((N, H, W, C), dtype=torch.uint8, device=device)
        self.N, self.H, self.W, self.C, self.ld = N, H, W, C, C
        self.p = self.t.data_ptr()


class Fp8Weights:
    """e4m3 shadow of one conv weight for the forward GEMM: [Opad][taps][I] bytes + alpha = 1/q per output channel
    (q: the power of two each output channel is scaled by before rounding).  Refreshed with the bf16 shadows."""

    def __init__(self, param, k, stride, pad):
        self.param = param
        self.O, self.I = param.shape[0], param.shape[1]
        self.k, self.stride, self.pad = k, stride, pad
        self.Opad = (self.O + 63) // 64 * 64
        self.version = -1
        self.w8 = self.alpha = None


_ZERO_BETA = {}


def _zero_beta(n, device):
    key = (torch.device(device).index, n)
    if key not in _ZERO_BETA:
        _ZERO_BETA[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return _ZERO_BETA[key]


def refresh_many_fp8(weights):
    """Re-quantises every stale Fp8Weights (multi-tensor launches of yv1_prep_weights_fp8_multi)."""
    import ctypes
    stale = []
    for w in weights:
        p = w.param
        ver = (p._version, _WEIGHT_EPOCH[0])
        if w.w8 is not None and ver == w.version and w.w8.device == p.device:
            continue
        if w.w8 is None or w.w8.device != p.device:
            w.w8 = torch.empty((w.Opad, w.k * w.k, w.I), dtype=torch.uint8, device=p.device)
            w.alpha = torch.empty(w.Opad, dtype=torch.float32, device=p.device)
        stale.append((w, ver))
    if not stale:
        return
    L = lib()
    maxn = L.yv1_prep_weights_fp8_max_tensors()
    dev = stale[0][0].param.device
    for i in range(0, len(stale), maxn):
        chunk = [w for w, _ in stale[i:i + maxn]]
        n = len(chunk)
        PA, IA, LA = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_longlong * (4 * n)
        strides = []
        for w in chunk:
            strides.extend(w.param.stride())
        check(L.yv1_prep_weights_fp8_multi(PA(*[w.param.data_ptr() for w in chunk]), LA(*strides), IA(*[w.O for w in chunk]),
                                           IA(*[w.I for w in chunk]), IA(*[w.k for w in chunk]),
                                           PA(*[w.w8.data_ptr() for w in chunk]), PA(*[w.alpha.data_ptr() for w in chunk]), n,
                                           stream_ptr(dev)), "yv1_prep_weights_fp8_multi")
    for w, ver in stale:
        w.version = ver


def quantize_fp8(x, out=None):
    """bf16 Act -> e4m3 Fp8Act (saturating at +-448)."""
    dev = x.t.device
    out = out or Fp8Act(x.N, x.H, x.W, x.C, dev)
    check(lib().yv1_quantize_bf16_to_fp8(x.p, x.ld, out.p, out.ld, x.npix, x.C, stream_ptr(dev)), "yv1_quantize_bf16_to_fp8")
    return out


def conv_fwd_fp8(x8, w8, y, want_stats=True):
    """y (bf16 Act) = conv(x8, w8) on the fp8 MFMA, dequantised; returns the BN-statistic partials like conv_fwd."""
    dev = x8.t.device
    L = lib()
    stats = None
    if want_stats:
        rows = L.yv1_conv2d_fp8_stats_rows(y.npix, w8.Opad)
        stats = _f32(rows * 2 * w8.Opad, dev).view(rows, 2, w8.Opad)
    check(L.yv1_conv2d_fwd_stats_nhwc_fp8(x8.p, ptr(w8.w8), ptr(w8.alpha), ptr(_zero_beta(w8.Opad, dev)), y.p, y.ld, ptr(stats),
                                          x8.N, x8.H, x8.W, x8.ld, w8.I, w8.Opad, w8.k, w8.stride, w8.pad, stream_ptr(dev)),
          "yv1_conv2d_fwd_stats_nhwc_fp8")
    return stats


def pack_input(images):
    """NCHW fp32 -> zero-padded NHWC4 bf16 [N][H+6][W+6][4]."""
    N, C, H, W = images.shape
    if C != 3:
        raise _lib.Yv1Error("the stem expects 3-channel images, got %d" % C)
    images = images.to(dtype=torch.float32).contiguous()
    xp = torch.empty((N, H + 6, W + 6, 4), dtype=torch.bfloat16, device=images.device)
    check(lib().yv1_pack_input_nhwc4(ptr(images), ptr(xp), N, H, W, stream_ptr(images.device)), "yv1_pack_input_nhwc4")
    return xp


def stem_fwd(xp, w, y, H, W):
    dev = xp.device
    rows = (y.npix + 127) // 128
    stats = _f32(rows * 2 * w.O, dev).view(rows, 2, w.O)
    check(lib().yv1_conv2d_stem_fwd_bf16(ptr(xp), ptr(w.fwd), y.p, y.N, H, W, w.O, y.ld, ptr(stats), stream_ptr(dev)),
          "yv1_conv2d_stem_fwd_bf16")
    return stats


def conv_dgrad(dy, w, dx, accumulate=False):
    """dx (+)= conv_transpose(dy, w).  dx has the forward input's geometry."""
    dev = dy.t.device
    check(lib().yv1_conv2d_dgrad_nhwc_bf16(dy.p, ptr(w.tr), dx.p, dx.N, dx.H, dx.W, dx.ld, w.Ipad, w.Opad, dy.ld, w.k,
                                           w.stride, w.pad, 1 if accumulate else 0, stream_ptr(dev)),
          "yv1_conv2d_dgrad_nhwc_bf16")


def conv_dgrad_add_masked(dy, w, dx, g, mask):
    """dx = conv_transpose(dy, w) + (mask ? g : 0) for the 1x1 stride-1 first convolution of an identity-shortcut
    block: the shortcut's gradient (block-output gradient ``g`` gated by the block's ReLU ``mask``) is added in the
    GEMM epilogue instead of being materialised and read-modify-written."""
    if w.k != 1 or w.stride != 1 or w.pad != 0:
        raise ValueError("conv_dgrad_add_masked: 1x1 stride-1 convolution only")
    if (g.N, g.H, g.W, g.C) != (dx.N, dx.H, dx.W, dx.C):
        raise ValueError("conv_dgrad_add_masked: g must have dx's geometry")
    dev = dy.t.device
    check(lib().yv1_conv2d_dgrad_add_masked_nhwc_bf16(dy.p, ptr(w.tr), dx.p, dx.N, dx.H, dx.W, dx.ld, w.Ipad, w.Opad, dy.ld,
                                                      g.p, g.ld, mask.p, mask.ld, stream_ptr(dev)),
          "yv1_conv2d_dgrad_add_masked_nhwc_bf16")


def conv_dgrad_add_masked_out(dy, w, dx, g, mask, out_mask=None, want_sum=False):
    """conv_dgrad_add_masked with two additions for the block BELOW (dx is its output gradient): ``out_mask`` -- that block's
    ReluMask: dx is stored already masked; ``want_sum`` -- per-tile column sums of the stored values (their sum is that
    block's bn3 dbeta).  ``mask`` None: ``g`` is added as it is (an already masked gradient).  Returns the partial rows
    [rows][C] (fp32) or None."""
    if w.k != 1 or w.stride != 1 or w.pad != 0:
        raise ValueError("conv_dgrad_add_masked_out: 1x1 stride-1 convolution only")
    if (g.N, g.H, g.W, g.C) != (dx.N, dx.H, dx.W, dx.C):
        raise ValueError("conv_dgrad_add_masked_out: g must have dx's geometry")
    dev = dy.t.device
    L = lib()
    gsum = None
    if want_sum:
        rows = L.yv1_conv2d_dgrad_gsum_rows(dx.npix, w.Ipad, w.Opad)
        gsum = _f32(rows * dx.C, dev).view(rows, dx.C)
    check(L.yv1_conv2d_dgrad_add_masked_out_nhwc_bf16(dy.p, ptr(w.tr), dx.p, dx.N, dx.H, dx.W, dx.ld, w.Ipad, w.Opad, dy.ld,
                                                      g.p, g.ld, mask.p if mask is not None else None,
                                                      mask.ld if mask is not None else 0,
                                                      out_mask.p if out_mask is not None else None,
                                                      out_mask.ld if out_mask is not None else 0, ptr(gsum), stream_ptr(dev)),
          "yv1_conv2d_dgrad_add_masked_out_nhwc_bf16")
    return gsum


def dgrad_gsum_rows(dy, w):
    """Rows of column-sum partials the 1x1 data gradient of ``dy`` through ``w`` reports (one per pixel tile)."""
    return lib().yv1_conv2d_dgrad_gsum_rows(dy.npix, w.Ipad, w.Opad)


def conv_dgrad_out(dy, w, dx, accumulate, out_mask, want_sum=True, gsum=None):
    """conv_dgrad for a 1x1 pad-0 convolution (stride 1 | 2) whose result is the output gradient of the block below: stored
    masked by that block's ReluMask, with the per-tile column sums of what this launch added (see
    conv_dgrad_add_masked_out).  Returns the partial rows or None.  ``gsum``: a [dgrad_gsum_rows(dy, w)][dx.C] slice of a
    caller-owned table to write the partials into (two launches into one block input: one table, no concatenation)."""
    if w.k != 1 or w.pad != 0:
        raise ValueError("conv_dgrad_out: 1x1 pad-0 convolution only")
    dev = dy.t.device
    L = lib()
    if gsum is not None:
        rows = L.yv1_conv2d_dgrad_gsum_rows(dy.npix, w.Ipad, w.Opad)
        if tuple(gsum.shape) != (rows, dx.C) or not gsum.is_contiguous() or gsum.dtype != torch.float32:
            raise ValueError("conv_dgrad_out: gsum must be a contiguous fp32 [%d][%d] table" % (rows, dx.C))
    elif want_sum:
        rows = L.yv1_conv2d_dgrad_gsum_rows(dy.npix, w.Ipad, w.Opad)
        gsum = _f32(rows * dx.C, dev).view(rows, dx.C)
    check(L.yv1_conv2d_dgrad_out_nhwc_bf16(dy.p, ptr(w.tr), dx.p, dx.N, dx.H, dx.W, dx.ld, w.Ipad, w.Opad, dy.ld, w.stride,
                                           1 if accumulate else 0, out_mask.p if out_mask is not None else None,
                                           out_mask.ld if out_mask is not None else 0, ptr(gsum), stream_ptr(dev)),
          "yv1_conv2d_dgrad_out_nhwc_bf16")
    return gsum


def wgrad_raw_buffers(x, dy):
    """(out, workspace, workspace bytes) of wgrad_raw(x, dy): allocated by the caller when the launch itself happens on
    another stream (SideStream: buffers belong to the main stream)."""
    dev = x.t.device
    out = torch.empty((dy.C, 1, x.C), dtype=torch.float32, device=dev)
    wsb = lib().yv1_conv2d_wgrad_workspace_bytes(x.N, dy.H, dy.W, x.C, dy.C, 1)
    return out, torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev), wsb


def wgrad_raw(x, dy, shared=False, buffers=None):
    """fp32 [Cdy][1][Cx] = dy^T x over all pixels (the 1x1 weight-gradient GEMM on arbitrary operands: T = gm^T z2 and
    G = z2^T z2 of the bn3 algebra).  ``shared``: the split-K width for launches that run beside another stream."""
    dev = x.t.device
    L = lib()
    out, ws, wsb = buffers if buffers is not None else wgrad_raw_buffers(x, dy)
    fn = L.yv1_conv2d_wgrad_shared_nhwc_bf16 if shared else L.yv1_conv2d_wgrad_nhwc_bf16
    check(fn(x.p, dy.p, ptr(out), x.N, x.H, x.W, x.ld, x.C, dy.C, dy.ld, 1, 1, 0, ptr(ws), wsb, stream_ptr(dev)),
          "yv1_conv2d_wgrad_nhwc_bf16")
    return out, ws


_ONES = {}


def _ones(n, device):
    key = (torch.device(device).index, n)
    if key not in _ONES:
        _ONES[key] = torch.ones(n, dtype=torch.float32, device=device)
    return _ONES[key]


import os as _os2
# "bn3's backward as algebra" (csrc/bn3alg.hip): identity-shortcut Bottlenecks up to this many planes
# run their BatchNorm-3 backward inside the 1x1 GEMMs next to it (0 = off)
# Measured per stage (interleaved A/B of the step): off 2938-2946 img/s, layer1 only (64) 2957, layer1-2 (128) 2974-2975,
# layer1-3 (256) 2926-2930 -- at 28x28 the two passes it removes (34 + 57 us) no longer cover T on the main stream, the
# 256 x 256 x 1024 operand build and the 25 % longer data gradient.
BN3_ALGEBRA_MAX_P = int(_os2.environ.get("YV1_BN3_ALGEBRA", "128"))
# projection Bottlenecks too (bn3 and the downsample BatchNorm, both on the same masked gradient).  Parity 3.4e-3 against the
# dual passes.  First measurement: LEVEL (3038-3051 vs 3036-3039 img/s) -- two T GEMMs in front of the data gradients, and each
# algebra call paid 227 us in k_bn3_dw's serial row loop.  After that loop was parallelised and the epilogue operands of the
# masked-output data gradients were hoisted: 2984 / 2991 -> 3047 / 3053 img/s (+2.1 %, same box, interleaved): ON.
BN3_ALGEBRA_PROJ = _os2.environ.get("YV1_BN3_ALGEBRA_PROJ", "1") != "0"


def _gsum_rows_le32(gsum, dev):
    """Pre-reduce the per-tile column sums to at most 32 rows (coalesced): the coefficient kernel reads them per channel."""
    rows, C = gsum.shape
    if rows <= 64:
        return gsum, rows
    RB = (rows + 31) // 32
    r2 = (rows + RB - 1) // RB
    out = _f32(r2 * C, dev)
    check(lib().yv1_reduce_rows(ptr(gsum), ptr(out), rows, C, RB, stream_ptr(dev)), "yv1_reduce_rows")
    return out.view(r2, C), r2


def bn3_algebra_backward(gm, gsum, z2, w3, st3, bn3, conv3_param, dz2, side, stride=1, accumulate=False, out_mask=None):
    """BatchNorm + pointwise-convolution backward as algebra (csrc/bn3alg.hip) from the MASKED block-output gradient ``gm``
    (Act, 4p channels) and its per-tile column sums ``gsum``.  ``z2``: the convolution's DENSE input [N,h,w,p] (conv3: z2;
    a projection shortcut: the block input at the pixels the strided convolution reads), ``w3`` / ``st3`` / ``bn3``: the
    convolution's ConvWeights and its BatchNorm; ``dz2``: where the input gradient goes (``stride`` 2 + ``accumulate``: the
    shortcut's scatter-accumulate into the block-input gradient; ``out_mask``: dz2 is the output gradient of the block below,
    stored masked by its ReluMask, and a fourth return value holds the column sums of what was added).  Returns (dgamma,
    dbeta, dW view[, sums]).  The stand-alone reduce /
    apply passes over gm and the convolution output do not run; the output gradient of the convolution is never formed."""
    dev = gm.t.device
    L = lib()
    s = stream_ptr(dev)
    p, C4 = z2.C, gm.C
    M = gm.npix
    # T = gm^T z on the MAIN stream: the coefficients, and with them the data gradient, wait for it
    T, ws_t = wgrad_raw(z2, gm, shared=False)
    gsum, rows = _gsum_rows_le32(gsum, dev)
    kk = torch.empty((3, C4), dtype=torch.float32, device=dev)        # k1, k2, k3*invstd
    if _ARENA[0] is not None:
        dgam, dbet = _grad_buf(bn3.weight, (C4,)), _grad_buf(bn3.bias, (C4,))
    else:
        gb = torch.empty((2, C4), dtype=torch.float32, device=dev)
        dgam, dbet = gb[0], gb[1]
    check(L.yv1_bn3_coeffs(ptr(gsum), rows, ptr(T), ptr(w3.fwd), p, C4, ptr(st3.mean), ptr(st3.invstd), ptr(bn3.weight),
                           float(M), ptr(dgam), ptr(dbet), ptr(kk[0]), ptr(kk[1]), ptr(kk[2]), s), "yv1_bn3_coeffs")
    wcat = torch.empty((p, C4 + p), dtype=torch.bfloat16, device=dev)
    bias = torch.empty(p, dtype=torch.float32, device=dev)
    check(L.yv1_bn3_build(ptr(w3.fwd), p, C4, ptr(kk[0]), ptr(kk[1]), ptr(kk[2]), ptr(st3.mean), ptr(wcat), ptr(bias), s),
          "yv1_bn3_build")
    mk = side.mark()
    osum = None
    if out_mask is not None:
        orows = L.yv1_conv2d_dgrad_gsum_rows(gm.npix, p, C4 + p)
        osum = _f32(orows * p, dev).view(orows, p)
    check(L.yv1_conv2d_dgrad_cat_bias_nhwc_bf16(gm.p, gm.ld, C4, z2.p, z2.ld, p, ptr(wcat), ptr(_ones(p, dev)), ptr(bias),
                                                dz2.p, dz2.ld, p, gm.N, gm.H, gm.W, stride, 1 if accumulate else 0,
                                                out_mask.p if out_mask is not None else None,
                                                out_mask.ld if out_mask is not None else 0, ptr(osum), s),
          "yv1_conv2d_dgrad_cat_bias_nhwc_bf16")
    dW = _grad_buf(conv3_param, (C4, 1, p))

    # buffers of the side-stream launches are allocated HERE, on the main stream (SideStream: kept alive until join())
    gbuf = wgrad_raw_buffers(z2, z2)
    srows = L.yv1_bn_reduce_rows(z2.npix, p)
    szp = _f32(srows * 2 * p, dev).view(srows, 2, p)
    # long partial tables are pre-reduced to <= 32 rows (coalesced, parallel): the dW kernel sums them per 32x32 tile
    RB = (srows + 31) // 32
    srows2 = (srows + RB - 1) // RB if srows > 64 else srows
    szp2 = _f32(srows2 * 2 * p, dev) if srows > 64 else szp

    def weight_side():
        overl = side.side is not None and not getattr(side, "wide", False)
        G, _ = wgrad_raw(z2, z2, shared=overl, buffers=gbuf)
        check(L.yv1_bn_stats(z2.p, z2.ld, z2.npix, p, ptr(szp), stream_ptr(dev)), "yv1_bn_stats")
        if szp2 is not szp:
            check(L.yv1_reduce_rows(ptr(szp), ptr(szp2), srows, 2 * p, RB, stream_ptr(dev)), "yv1_reduce_rows")
        check(L.yv1_bn3_dw(ptr(T), ptr(G), ptr(szp2), srows2, ptr(w3.fwd), p, C4, ptr(kk[0]), ptr(kk[1]), ptr(kk[2]),
                           ptr(st3.mean), ptr(dW), stream_ptr(dev)), "yv1_bn3_dw")
    side.run(weight_side, z2.t, T, ws_t, kk, wcat, bias, dW, gbuf[0], gbuf[1], szp, szp2, after=mk)
    if out_mask is not None:
        return dgam, dbet, dW.view(C4, 1, 1, p).permute(0, 3, 1, 2), osum
    return dgam, dbet, dW.view(C4, 1, 1, p).permute(0, 3, 1, 2)


def subsample2(x):
    """The pixels (2h, 2w) of an NHWC bf16 activation, dense: what a stride-2 1x1 convolution reads."""
    y = new_act(x.N, x.H // 2, x.W // 2, x.C, x.t.device)
    check(lib().yv1_subsample2_nhwc_bf16(x.p, x.ld, y.p, y.ld, x.N, x.H, x.W, x.C, stream_ptr(x.t.device)),
          "yv1_subsample2_nhwc_bf16")
    return y


class SideStream:
    """Runs leaf work (weight gradients: nothing else in the backward consumes them) on a second HIP stream so it
    overlaps with the dgrad / BatchNorm chain on the main stream -- eagerly and inside a captured hipGraph
    (fork/join).  Buffers are allocated on the main stream and kept alive until ``join()``, so the caching
    allocator can never hand memory the side stream still reads to later main-stream work."""

    def __init__(self, device, enabled=True):
        self.main = torch.cuda.current_stream(device)
        self.side = _side_stream_for(device) if enabled else None
        self.keep = []
        self.wide = False                    # conv_wgrad: use the split-K width tuned for kernels that run alone

    def mark(self):
        """Event at the current end of the main stream: ``run(..., after=mark)`` orders side work after THIS point, so
        main-stream kernels launched between mark() and run() (the data gradient -- the critical path) are enqueued
        ahead of the side work without the side work having to wait for them."""
        if self.side is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self.main)
        self._mark_launches = (ev, _lib.LAUNCHES[0])
        return ev

    def run(self, fn, *keep_alive, after=None):
        """fn() launches kernels; they are ordered after ``after`` (a mark()) or, without it, after everything issued
        on the main stream so far."""
        if self.side is None:
            return fn()
        if after is not None:
            # Inside a capture the HIP graph executor gives a node's FIRST captured successor the node's own hardware queue
            # and pushes later successors to other queues: side work captured before the next main-chain kernel moves the
            # main chain off its queue (after four such forks it shared a queue with the weight gradients: -3 % at S=7,
            # -10 % at S=14).  A mark must therefore be followed by a main-stream launch first.
            m = getattr(self, "_mark_launches", None)
            if m is not None and m[0] is after and m[1] == _lib.LAUNCHES[0] and torch.cuda.is_current_stream_capturing():
                raise _lib.Yv1Error("SideStream.run(after=mark) captured before any main-stream launch since mark(): "
                                    "enqueue the main chain's next kernel first (hipGraph queue assignment)")
            self.side.wait_event(after)
        else:
            self.side.wait_stream(self.main)
        with torch.cuda.stream(self.side):
            out = fn()
        self.keep.extend(keep_alive)
        return out

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)
        self.keep = []


_SIDE_STREAMS = {}


def _side_stream_for(device):
    key = torch.device(device).index
    if key not in _SIDE_STREAMS:
        import os
        prio = int(os.environ.get("YV1_SIDE_PRIORITY", "0"))      # tuning: -1 = high priority for the weight-gradient stream
        _SIDE_STREAMS[key] = torch.cuda.Stream(device, priority=prio)
    return _SIDE_STREAMS[key]


class GradArena:
    """One flat fp32 buffer for every parameter gradient of a backbone.  Slices are handed out in first-use order -- the
    order the backward executor produces the gradients, which is the order ``distributed.GradSync`` reduces them in --
    and the same parameter gets the same slice on every later backward.  The weight-gradient / BatchNorm-backward
    kernels write straight into their slice, ``param.grad`` is a view of it, and a bucket of consecutive parameters is
    one contiguous range: the all-reduce runs in place, with no flatten pass before it and no copy-back after it
    (GradSync without an arena concatenates and scatters: two passes over 165 MB per step for ResNet-50)."""

    def __init__(self, net, device):
        # sized from what the gradient-producing ops will actually request: a convolution weight asks for its PADDED
        # shape (conv_wgrad: Opad x taps x Ipad -- output channels rounded up to 32, e.g. the 30-channel head; the stem
        # asks for O x 7 x 7 x 3), every other parameter for its own element count
        from .engine import ConvParam
        conv_params = {}
        for m in net.modules():
            if isinstance(m, ConvParam):
                O, I = m.weight.shape[0], m.weight.shape[1]
                opad = (O + 31) // 32 * 32
                conv_params[id(m.weight)] = m.weight.numel() if I == 3 else opad * m.kernel_size * m.kernel_size * I
        total = sum((conv_params.get(id(p), p.numel()) + 3) // 4 * 4 for p in net.parameters())
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.ranges = {}                   # id(param) -> (offset, numel)
        self.top = 0

    def get(self, param, numel):
        r = self.ranges.get(id(param))
        if r is None:
            n4 = (numel + 3) // 4 * 4      # 16-byte aligned slices
            if self.top + n4 > self.flat.numel():
                raise _lib.Yv1Error("gradient arena exhausted")
            r = (self.top, numel)
            self.ranges[id(param)] = r
            self.top += n4
        elif r[1] != numel:
            raise _lib.Yv1Error("gradient arena: a parameter asked for %d elements, had %d" % (numel, r[1]))
        return self.flat[r[0]:r[0] + numel]

    def span(self, params):
        """[lo, hi) of the arena covered by ``params`` (must already have slices); hi is 16-byte rounded."""
        rs = [self.ranges[id(p)] for p in params]
        lo = min(r[0] for r in rs)
        hi = max(r[0] + (r[1] + 3) // 4 * 4 for r in rs)
        return lo, hi


_ARENA = [None]


def set_grad_arena(arena):
    """Activates (or, with None, deactivates) a GradArena for the gradient-producing ops below."""
    _ARENA[0] = arena


def _grad_buf(param, shape):
    """fp32 tensor of ``shape`` for the gradient of ``param``: its arena slice when an arena is active, else fresh."""
    n = 1
    for d in shape:
        n *= d
    a = _ARENA[0]
    if a is not None and param is not None:
        return a.get(param, n).view(shape)
    return torch.empty(shape, dtype=torch.float32, device=param.device if param is not None else None)


def conv_wgrad(x, dy, w, side=None, after=None):
    """Returns the fp32 gradient as an OIHW view whose storage is [O][kh][kw][I] (channels_last).
    With ``side`` (a SideStream) the kernels run on the side stream, ordered after ``after`` (SideStream.mark())."""
    dev = x.t.device
    L = lib()
    taps = w.k * w.k
    g = _grad_buf(w.param, (w.Opad, taps, w.Ipad))
    wsb = L.yv1_conv2d_wgrad_workspace_bytes(x.N, dy.H, dy.W, w.Ipad, w.Opad, w.k)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)

    # beside the main stream's kernels a narrower split-K wins, alone on the device the wider one (see yv1.h); ``side.wide``:
    # this stretch of the side stream runs after the main chain has ended (the tail of the backward) -- alone on the device
    overlapped = side is not None and side.side is not None and not getattr(side, "wide", False)
    fn = L.yv1_conv2d_wgrad_shared_nhwc_bf16 if overlapped else L.yv1_conv2d_wgrad_nhwc_bf16

    def launch():
        check(fn(x.p, dy.p, ptr(g), x.N, x.H, x.W, x.ld, w.Ipad, w.Opad, dy.ld, w.k, w.stride, w.pad, ptr(ws), wsb,
                 stream_ptr(dev)), "yv1_conv2d_wgrad_nhwc_bf16")
    if side is None:
        launch()
    else:
        side.run(launch, x.t, dy.t, ws, g, after=after)
    return g[:w.O].view(w.O, w.k, w.k, w.Ipad).permute(0, 3, 1, 2)


def stem_wgrad(xp, dy, w, H, W, side=None):
    dev = xp.device
    L = lib()
    g = torch.empty((w.O, 7, 32), dtype=torch.float32, device=dev)
    wsb = L.yv1_conv2d_stem_wgrad_workspace_bytes(dy.N, H, W, w.O)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    out = _grad_buf(w.param, (w.O, 7, 7, 3)).permute(0, 3, 1, 2)
    so, si, sh, sw = out.stride()

    def launch():
        check(L.yv1_conv2d_stem_wgrad_bf16(ptr(xp), dy.p, ptr(g), dy.N, H, W, w.O, dy.ld, ptr(ws), wsb, stream_ptr(dev)),
              "yv1_conv2d_stem_wgrad_bf16")
        check(L.yv1_unpack_stem_grad(ptr(g), ptr(out), so, si, sh, sw, w.O, stream_ptr(dev)), "yv1_unpack_stem_grad")
    if side is None:
        launch()
    else:
        side.run(launch, xp, dy.t, ws, g, out)
    return out


# ------------------------------------------------------------------ batch norm
class BNState:
    """Per-forward state of one BatchNorm: [mean, invstd, scale, shift] rows of one fp32 buffer."""
    __slots__ = ("buf", "C", "count")

    def __init__(self, C, device):
        self.buf = torch.empty((4, C), dtype=torch.float32, device=device)
        self.C = C
        self.count = 0

    mean = property(lambda s: s.buf[0])
    invstd = property(lambda s: s.buf[1])
    scale = property(lambda s: s.buf[2])
    shift = property(lambda s: s.buf[3])


def _shrink_partials(part, rows, W, dev):
    """Pre-reduction of very long partial tables (the finalize kernels sum up to ~2k rows themselves)."""
    if rows <= 2048:
        return part, rows
    RB = (rows + 255) // 256                    # 256 x (W/128) workgroups: enough to stream the table at HBM rate
    r2 = (rows + RB - 1) // RB
    out = _f32(r2 * W, dev)
    check(lib().yv1_reduce_rows(ptr(part), ptr(out), rows, W, RB, stream_ptr(dev)), "yv1_reduce_rows")
    return out, r2


def bn_finalize(stats, count, bn, C=None, training=True, seg=None):
    """stats [rows][2][ld] partials -> BNState (and running-stat update of module ``bn``).
    ``seg`` = (part [rows][2][Cseg], c0) with a one-row ``stats`` table (DenseNet): the table's columns [c0, c0+Cseg) are
    still the partial rows of the convolution that produced those features -- merged into the table by the same launch."""
    dev = stats.device
    rows, _, ld = stats.shape
    C = C or bn.num_features
    st = BNState(C, dev)
    st.count = count
    if seg is not None:
        part, c0 = seg
        prow, _, cseg = part.shape
        if rows != 1 or c0 + cseg > C:
            raise ValueError("bn_finalize: a merged segment needs the one-row table and must lie inside the finalized channels")
        part, prow = _shrink_partials(part, prow, 2 * cseg, dev)
        check(lib().yv1_bn_finalize_merged(ptr(stats), C, ld, float(count), ptr(bn.weight), ptr(bn.bias), BN_EPS, BN_MOMENTUM,
                                           ptr(bn.running_mean) if training else None,
                                           ptr(bn.running_var) if training else None, ptr(st.mean), ptr(st.invstd),
                                           ptr(st.scale), ptr(st.shift), ptr(part), prow, c0, cseg, stream_ptr(dev)),
              "yv1_bn_finalize_merged")
        return st
    part, rows = _shrink_partials(stats, rows, 2 * ld, dev)
    check(lib().yv1_bn_finalize(ptr(part), rows, C, ld, float(count), ptr(bn.weight), ptr(bn.bias), BN_EPS, BN_MOMENTUM,
                                ptr(bn.running_mean) if training else None, ptr(bn.running_var) if training else None,
                                ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), stream_ptr(dev)),
          "yv1_bn_finalize")
    return st


# ---- finalize fused into the consuming launch (csrc/elementwise.hip, "finalize fused into the consuming launch")
import os as _os
# Measured (tools/bench_bn_fused.py, profiles/r03_bn_fused_table.txt): the fused form is SLOWER on every shape -- +14 us on
# the small tensors, +50 us at C = 2048 -- and the step loses 13 % with it.  A dependent launch inside a captured graph
# costs ~1-2 us and a finalize kernel ~3 us (the 5-7 us / 4.5 us gap figures of round 2 were rocprofv3's), while
# publishing through memory and polling across the XCDs costs 13-18 us per hand-off (the guide's "barrier-counter" row
# says 7-26).  Off by default; kept as the record of the experiment, bit-identical to the launch pairs (tests).
BN_FUSED = _os.environ.get("YV1_BN_FUSED", "0") == "1"
_SYNC = {}
_SYNC_SLOTS = 4096


def _sync_slot(device):
    """(sync pointer, fault pointer): a self-resetting counter pair for ONE fused launch and the device's fault word.
    Slots are handed out round-robin from one zero-initialised buffer per device; a slot comes round again 4096 fused
    launches later, long after the launch that used it has finished (the kernel leaves the pair zeroed)."""
    key = torch.device(device).index
    ent = _SYNC.get(key)
    if ent is None:
        ent = [torch.zeros((_SYNC_SLOTS + 1) * 2, dtype=torch.int32, device=device), 0]
        _SYNC[key] = ent
    buf = ent[0]
    ent[1] = ent[1] % _SYNC_SLOTS + 1                     # slots 1 .. 4096; words 0/1 of the buffer: fault word + spare
    base = buf.data_ptr()
    return _lib.c_p(base + 8 * ent[1]), _lib.c_p(base)


def fused_sync_fault(device=None):
    """True when a consumer workgroup of a fused finalize+apply launch ever gave up waiting for its producers (never
    observed; the spin's exit condition).  Host sync.  Tests and bench.py assert it is False."""
    out = False
    for key, (buf, _) in _SYNC.items():
        if device is None or torch.device(device).index == key:
            out = out or bool(int(buf[0].item()) != 0)
    return out


def bn_finalize_apply(stats, count, bn, y, z, relu=True, residual=None, res_stats=None, res_bn=None, res_state=None,
                      want_mask=False, z8=None, C=None):
    """bn_finalize + bn_apply in ONE launch (training mode): ``stats`` are the partial rows of the convolution that produced
    ``y``; with ``res_stats`` / ``res_bn`` the residual is a raw projection-shortcut output whose BatchNorm is finalized by
    the same launch (else ``res_state``: an already finalized BNState, or None for a plain residual).
    Returns (BNState of bn, ReluMask or None, BNState of res_bn or None)."""
    dev = y.t.device
    rows, _, ld = stats.shape
    C = C or bn.num_features
    st = BNState(C, dev)
    st.count = count
    part, rows = _shrink_partials(stats, rows, 2 * ld, dev)
    rst, rpart, rrows, rld = res_state, None, 0, 0
    if res_stats is not None:
        rrows, _, rld = res_stats.shape
        rst = BNState(C, dev)
        rst.count = count
        rpart, rrows = _shrink_partials(res_stats, rrows, 2 * rld, dev)
    mask = ReluMask(y.npix, y.C, dev) if (want_mask and relu) else None
    sync, fault = _sync_slot(dev)
    check(lib().yv1_bn_finalize_apply(
        ptr(part), rows, ld, float(count), ptr(bn.weight), ptr(bn.bias), BN_EPS, BN_MOMENTUM, ptr(bn.running_mean),
        ptr(bn.running_var), ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift),
        ptr(rpart), rrows, rld, ptr(res_bn.weight) if rpart is not None else None, ptr(res_bn.bias) if rpart is not None else None,
        ptr(res_bn.running_mean) if rpart is not None else None, ptr(res_bn.running_var) if rpart is not None else None,
        ptr(rst.mean) if rpart is not None else None, ptr(rst.invstd) if rpart is not None else None,
        ptr(rst.scale) if rst is not None else None, ptr(rst.shift) if rst is not None else None,
        y.p, y.ld, z.p, z.ld, residual.p if residual is not None else None, residual.ld if residual is not None else 0,
        y.npix, y.C, 1 if relu else 0, mask.p if mask is not None else None, z8.p if z8 is not None else None,
        z8.ld if z8 is not None else 0, sync, fault, stream_ptr(dev)), "yv1_bn_finalize_apply")
    return st, mask, (rst if res_stats is not None else None)


def bn_finalize_merged_apply(table, count, bn, C, x, z, seg=None, relu=True):
    """DenseNet: bn_finalize(table, ..., seg=seg) + bn_apply(x, st, z) in one launch.  ``table`` [1][2][ld]."""
    dev = x.t.device
    rows, _, ld = table.shape
    if rows != 1:
        raise ValueError("bn_finalize_merged_apply needs the one-row table")
    st = BNState(C, dev)
    st.count = count
    part = prow = None
    c0 = cseg = 0
    if seg is not None:
        part, c0 = seg
        prow, _, cseg = part.shape
        if c0 + cseg > C:
            raise ValueError("bn_finalize_merged_apply: the merged segment must lie inside the finalized channels")
        part, prow = _shrink_partials(part, prow, 2 * cseg, dev)
    sync, fault = _sync_slot(dev)
    check(lib().yv1_bn_finalize_merged_apply(ptr(table), C, ld, float(count), ptr(bn.weight), ptr(bn.bias), BN_EPS, BN_MOMENTUM,
                                             ptr(bn.running_mean), ptr(bn.running_var), ptr(st.mean), ptr(st.invstd),
                                             ptr(st.scale), ptr(st.shift), ptr(part), prow or 0, c0, cseg, x.p, x.ld, z.p, z.ld,
                                             x.npix, 1 if relu else 0, sync, fault, stream_ptr(dev)),
          "yv1_bn_finalize_merged_apply")
    return st


def bn_eval_state(bn):
    dev = bn.weight.device
    st = BNState(bn.num_features, dev)
    check(lib().yv1_bn_eval_coeffs(st.C, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var), BN_EPS,
                                   ptr(st.scale), ptr(st.shift), stream_ptr(dev)), "yv1_bn_eval_coeffs")
    return st


def stats_merge(part, table, c0):
    """part [rows][2][Cseg] -> table[2][ld] columns [c0, c0+Cseg)."""
    rows, _, cseg = part.shape
    dev = part.device
    part2, rows = _shrink_partials(part, rows, 2 * cseg, dev)
    check(lib().yv1_stats_merge(ptr(part2), rows, cseg, ptr(table), table.shape[1], c0, stream_ptr(dev)), "yv1_stats_merge")


def bn_stats(x):
    """Stand-alone batch statistics of an activation window -> partials [rows][2][C]."""
    dev = x.t.device
    rows = lib().yv1_bn_reduce_rows(x.npix, x.C)
    part = _f32(rows * 2 * x.C, dev).view(rows, 2, x.C)
    check(lib().yv1_bn_stats(x.p, x.ld, x.npix, x.C, ptr(part), stream_ptr(dev)), "yv1_bn_stats")
    return part


class ReluMask:
    """1 bit per element sign mask written by bn_apply (uint8 [npix][C/8]); stands in for the activation
    itself in bn_backward(mask_mode=3) -- 16x fewer bytes than re-reading the bf16 output."""
    __slots__ = ("t", "p", "ld")

    def __init__(self, npix, C, device):
        self.t = torch.empty((npix, C // 8), dtype=torch.uint8, device=device)
        self.p = self.t.data_ptr()
        self.ld = C // 8


def bn_apply(y, st, z, relu=True, residual=None, res_state=None, want_mask=False, z8=None):
    """z = relu?(scale*y + shift [+ residual | + BN'd projection]); ``z8`` (Fp8Act): an e4m3 copy of z written by the same
    launch (operand of the next fp8 convolution)."""
    dev = y.t.device
    mask = ReluMask(y.npix, y.C, dev) if (want_mask and relu) else None
    args = (y.p, y.ld, z.p, z.ld, residual.p if residual is not None else None,
            residual.ld if residual is not None else 0, ptr(st.scale), ptr(st.shift),
            ptr(res_state.scale) if res_state is not None else None,
            ptr(res_state.shift) if res_state is not None else None, y.npix, y.C, 1 if relu else 0,
            mask.p if mask is not None else None)
    if z8 is None:
        check(lib().yv1_bn_apply(*args, stream_ptr(dev)), "yv1_bn_apply")
    else:
        check(lib().yv1_bn_apply_q8(*args, z8.p, z8.ld, stream_ptr(dev)), "yv1_bn_apply_q8")
    return mask


def bn_backward(dz, y, st, bn, dy, mask_mode, z=None, dres=None, accumulate=False, pool_idx=None):
    """BN (+ReLU) backward.  mask_mode 0: no ReLU; 1: mask from ``z`` > 0; 2: mask from scale*y+shift > 0;
    3: ``z`` is the ReluMask bn_apply wrote.
    Writes dy (grad wrt the raw conv output) and optionally dres (masked dz, the identity-shortcut
    gradient).  Returns (dgamma, dbeta).
    With ``pool_idx`` (the index tensor of ``maxpool_fwd``) ``dz`` is the gradient of the 3x3/2 max pool that follows
    the BN(+ReLU): the pool's backward is gathered inside both BatchNorm-backward kernels (stems)."""
    dev = y.t.device
    L = lib()
    C = y.C
    rows = L.yv1_bn_reduce_rows(y.npix, C)
    part = _f32(rows * 2 * C, dev)
    s = stream_ptr(dev)
    zp, zld = (z.p, z.ld) if z is not None else (None, 0)
    if pool_idx is not None:
        if z is not None or dres is not None or accumulate:
            raise ValueError("bn_backward: the pooled form takes mask_mode 0/2 only, no dres, no accumulate")
        check(L.yv1_bn_bwd_reduce_pooled(dz.p, dz.ld, ptr(pool_idx), y.p, y.ld, ptr(st.mean), ptr(st.invstd), ptr(st.scale),
                                         ptr(st.shift), y.N, y.H, y.W, C, mask_mode, ptr(part), s), "yv1_bn_bwd_reduce_pooled")
    else:
        check(L.yv1_bn_bwd_reduce(dz.p, dz.ld, zp, zld, y.p, y.ld, ptr(st.mean), ptr(st.invstd), ptr(st.scale),
                                  ptr(st.shift), y.npix, C, mask_mode, ptr(part), s), "yv1_bn_bwd_reduce")
    part, rows = _shrink_partials(part, rows, 2 * C, dev)
    gb = torch.empty((5, C), dtype=torch.float32, device=dev)     # dgamma, dbeta, k1, k2, k3
    dgam, dbet = gb[0], gb[1]
    if _ARENA[0] is not None and bn is not None and bn.weight.numel() == C:
        dgam, dbet = _grad_buf(bn.weight, (C,)), _grad_buf(bn.bias, (C,))
    if BN_FUSED and pool_idx is None:
        sync, fault = _sync_slot(dev)
        check(L.yv1_bn_bwd_finalize_apply(ptr(part), rows, float(y.npix), ptr(bn.weight) if bn is not None else None, ptr(dgam),
                                          ptr(dbet), ptr(gb[2]), ptr(gb[3]), ptr(gb[4]), dz.p, dz.ld, zp, zld, y.p, y.ld,
                                          ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), y.npix, C, mask_mode,
                                          dy.p, dy.ld, dres.p if dres is not None else None, dres.ld if dres is not None else 0,
                                          1 if accumulate else 0, sync, fault, s), "yv1_bn_bwd_finalize_apply")
        return dgam, dbet
    check(L.yv1_bn_bwd_finalize(ptr(part), rows, C, float(y.npix), ptr(bn.weight) if bn is not None else None,
                                ptr(st.invstd), ptr(dgam), ptr(dbet), ptr(gb[2]), ptr(gb[3]), ptr(gb[4]), s),
          "yv1_bn_bwd_finalize")
    if pool_idx is not None:
        check(L.yv1_bn_bwd_apply_pooled(dz.p, dz.ld, ptr(pool_idx), y.p, y.ld, ptr(st.mean), ptr(st.invstd), ptr(st.scale),
                                        ptr(st.shift), ptr(gb[2]), ptr(gb[3]), ptr(gb[4]), y.N, y.H, y.W, C, mask_mode,
                                        dy.p, dy.ld, s), "yv1_bn_bwd_apply_pooled")
        return dgam, dbet
    check(L.yv1_bn_bwd_apply(dz.p, dz.ld, zp, zld, y.p, y.ld, ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift),
                             ptr(gb[2]), ptr(gb[3]), ptr(gb[4]), y.npix, C, mask_mode, dy.p, dy.ld,
                             dres.p if dres is not None else None, dres.ld if dres is not None else 0,
                             1 if accumulate else 0, s),
          "yv1_bn_bwd_apply")
    return dgam, dbet


# "deferred BatchNorm backward" (csrc/bn_deferred.hip): DenseNet's norm1 / transition norm in front of
# a pointwise convolution -- the reduction-free term rides in the data gradient's epilogue, the rest is summed as per-channel
# coefficients and subtracted once per channel (0 = the dgrad + reduce + finalize + apply sequence)
BN_DEFERRED = _os.environ.get("YV1_BN_DEFERRED", "1") != "0"
# corrections applied by the NEXT data gradient into the buffer (one uncorrected term at a time) instead of summed up and
# applied once per channel slice (0: the first form of the round -- same launches, lower precision)
BN_DEFERRED_PENDING = _os.environ.get("YV1_BN_DEFERRED_PENDING", "1") != "0"


def conv_dgrad_bn_deferred(dy, w, dx, x, st, accumulate=True, pending=None):
    """dx (+)= scale * mask * conv_transpose(dy, w) for the 1x1 stride-1 convolution ``w`` whose input was relu(bn(x)) with the
    TRAINING-mode BNState ``st`` (mask = scale*x + shift > 0); returns the partial sums [rows][2][C] of the masked gradient
    that bn_bwd_finalize_deferred turns into dgamma / dbeta and the correction coefficients.  ``pending`` ([2][C] rows KA, KB):
    the correction the PREVIOUS launch into ``dx`` still owes these channels, subtracted in the same pass."""
    if w.k != 1 or w.stride != 1 or w.pad != 0:
        raise ValueError("conv_dgrad_bn_deferred: 1x1 stride-1 pad-0 convolution only")
    if (x.N, x.H, x.W, x.C) != (dx.N, dx.H, dx.W, dx.C) or x.C != w.Ipad:
        raise ValueError("conv_dgrad_bn_deferred: x and dx must be the convolution input's window")
    dev = dy.t.device
    L = lib()
    wt_rows = w.tr.shape[0]                      # ConvWeights pads the transposed copy of 1x1 weights to 128 rows (zeros)
    rows = L.yv1_conv2d_dgrad_bn_deferred_rows(dx.npix, w.Ipad, w.Opad, wt_rows)
    part = _f32(rows * 2 * dx.C, dev)
    check(L.yv1_conv2d_dgrad_bn_deferred_nhwc_bf16(dy.p, ptr(w.tr), dx.p, dx.N, dx.H, dx.W, dx.ld, w.Ipad, w.Opad, dy.ld, x.p,
                                                   x.ld, ptr(st.scale), ptr(st.shift), ptr(st.mean), 1 if accumulate else 0,
                                                   ptr(part), wt_rows, ptr(pending[0]) if pending is not None else None,
                                                   ptr(pending[1]) if pending is not None else None, stream_ptr(dev)),
          "yv1_conv2d_dgrad_bn_deferred_nhwc_bf16")
    return part.view(rows, 2, dx.C)


def bn_bwd_finalize_deferred(part, count, bn, st, K, accumulate=True):
    """Partial sums of conv_dgrad_bn_deferred -> (dgamma, dbeta) of module ``bn``; the affine correction this BatchNorm owes
    the gradient of its input channels is added to (``accumulate``) or stored in K = [2][C] (rows KA, KB)."""
    dev = part.device
    rows, _, C = part.shape
    flat, rows = _shrink_partials(part.reshape(-1), rows, 2 * C, dev)
    if _ARENA[0] is not None:
        dgam, dbet = _grad_buf(bn.weight, (C,)), _grad_buf(bn.bias, (C,))
    else:
        gb = torch.empty((2, C), dtype=torch.float32, device=dev)
        dgam, dbet = gb[0], gb[1]
    check(lib().yv1_bn_bwd_finalize_deferred(ptr(flat), rows, C, float(count), ptr(bn.weight), ptr(st.mean), ptr(st.invstd),
                                             ptr(dgam), ptr(dbet), ptr(K[0]), ptr(K[1]), 1 if accumulate else 0,
                                             stream_ptr(dev)), "yv1_bn_bwd_finalize_deferred")
    return dgam, dbet


def bn_deferred_fix(g, x, K):
    """g -= KA + KB * x over the channel window ``g`` / ``x`` share (K = [2][C] rows for exactly that window)."""
    if (g.N, g.H, g.W, g.C) != (x.N, x.H, x.W, x.C) or K.shape[1] != g.C:
        raise ValueError("bn_deferred_fix: g, x and K must cover the same channel window")
    check(lib().yv1_bn_deferred_fix(g.p, g.ld, x.p, x.ld, ptr(K[0]), ptr(K[1]), g.npix, g.C, stream_ptr(g.t.device)),
          "yv1_bn_deferred_fix")


# the reduction pass of a BatchNorm(+ReLU) backward inside the data gradient that produces its input gradient (0 = separate
# yv1_bn_bwd_reduce pass)
BN_SUMS_IN_DGRAD = _os.environ.get("YV1_BN_SUMS_IN_DGRAD", "1") != "0"


def conv_dgrad_bn_sums(dy, w, dx, y, st):
    """dx = mask * conv_transpose(dy, w) with mask = (scale*y + shift > 0) of the TRAINING-mode BatchNorm ``st`` over ``y``
    (the convolution's input was relu(bn(y))), + the BatchNorm-backward sums of dx per pixel tile.  Returns the partial rows
    [rows][2][C] for bn_backward_from_sums, or None when this shape has no such kernel (dx untouched: run conv_dgrad +
    bn_backward instead)."""
    if not BN_SUMS_IN_DGRAD or w.stride != 1 or 2 * w.pad != w.k - 1:
        return None
    if (y.N, y.H, y.W, y.C) != (dx.N, dx.H, dx.W, dx.C):
        raise ValueError("conv_dgrad_bn_sums: y and dx must have the convolution input's geometry")
    dev = dy.t.device
    L = lib()
    rows = L.yv1_conv2d_dgrad_bn_sums_rows(dx.npix, w.Ipad, w.Opad, w.k, w.pad)
    if rows <= 0:
        return None
    part = _f32(rows * 2 * dx.C, dev)
    check(L.yv1_conv2d_dgrad_bn_sums_nhwc_bf16(dy.p, ptr(w.tr), dx.p, dx.N, dx.H, dx.W, dx.ld, w.Ipad, w.Opad, dy.ld, w.k, w.pad,
                                               y.p, y.ld, ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), ptr(part),
                                               stream_ptr(dev)), "yv1_conv2d_dgrad_bn_sums_nhwc_bf16")
    return part.view(rows, 2, dx.C)


def bn_backward_from_sums(d, y, st, bn, dy, part):
    """BatchNorm(+ReLU) backward from the MASKED gradient ``d`` and its partial sums (conv_dgrad_bn_sums): finalize + apply,
    no reduction pass.  Returns (dgamma, dbeta)."""
    dev = y.t.device
    L = lib()
    C = y.C
    rows = part.shape[0]
    flat, rows = _shrink_partials(part.reshape(-1), rows, 2 * C, dev)
    gb = torch.empty((5, C), dtype=torch.float32, device=dev)     # dgamma, dbeta, k1, k2, k3
    dgam, dbet = gb[0], gb[1]
    if _ARENA[0] is not None and bn is not None and bn.weight.numel() == C:
        dgam, dbet = _grad_buf(bn.weight, (C,)), _grad_buf(bn.bias, (C,))
    s = stream_ptr(dev)
    check(L.yv1_bn_bwd_finalize(ptr(flat), rows, C, float(y.npix), ptr(bn.weight) if bn is not None else None,
                                ptr(st.invstd), ptr(dgam), ptr(dbet), ptr(gb[2]), ptr(gb[3]), ptr(gb[4]), s),
          "yv1_bn_bwd_finalize")
    check(L.yv1_bn_bwd_apply(d.p, d.ld, None, 0, y.p, y.ld, ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift),
                             ptr(gb[2]), ptr(gb[3]), ptr(gb[4]), y.npix, C, 0, dy.p, dy.ld, None, 0, 0, s),
          "yv1_bn_bwd_apply")
    return dgam, dbet


def bn_backward_dual(dz, mask, a, b):
    """BatchNorm backward of TWO BatchNorms that receive the same ReLU-masked gradient (a projection Bottleneck's bn3 and
    downsample BatchNorm, OriginResNet.py:100-105): ``a`` / ``b`` = (y, BNState, bn module, dy).  One reduction pass and
    one apply pass read ``dz`` and the 1-bit ``mask`` once for both.  Returns ((dgamma_a, dbeta_a), (dgamma_b, dbeta_b))."""
    (ya, sta, bna, dya), (yb, stb, bnb, dyb) = a, b
    dev = ya.t.device
    L = lib()
    C = ya.C
    s = stream_ptr(dev)
    rows = L.yv1_bn_reduce_rows(ya.npix, C)
    pa, pb = _f32(rows * 2 * C, dev), _f32(rows * 2 * C, dev)
    check(L.yv1_bn_bwd_reduce_dual(dz.p, dz.ld, mask.p, mask.ld, ya.p, ya.ld, ptr(sta.mean), ptr(sta.invstd), yb.p, yb.ld,
                                   ptr(stb.mean), ptr(stb.invstd), ya.npix, C, 3, ptr(pa), ptr(pb), s), "yv1_bn_bwd_reduce_dual")
    if BN_FUSED and rows <= 2048:
        k6 = torch.empty((6, C), dtype=torch.float32, device=dev)
        gr = []
        for bn in (bna, bnb):
            if _ARENA[0] is not None:
                gr.append((_grad_buf(bn.weight, (C,)), _grad_buf(bn.bias, (C,))))
            else:
                g2 = torch.empty((2, C), dtype=torch.float32, device=dev)
                gr.append((g2[0], g2[1]))
        sync, fault = _sync_slot(dev)
        check(L.yv1_bn_bwd_finalize_apply_dual(ptr(pa), ptr(pb), rows, float(ya.npix), ptr(bna.weight), ptr(bnb.weight),
                                               ptr(gr[0][0]), ptr(gr[0][1]), ptr(gr[1][0]), ptr(gr[1][1]), ptr(k6), dz.p, dz.ld,
                                               mask.p, mask.ld, ya.p, ya.ld, ptr(sta.mean), ptr(sta.invstd), dya.p, dya.ld,
                                               yb.p, yb.ld, ptr(stb.mean), ptr(stb.invstd), dyb.p, dyb.ld, ya.npix, C, 3,
                                               sync, fault, s), "yv1_bn_bwd_finalize_apply_dual")
        return gr[0], gr[1]
    out, ks = [], []
    for part, st, bn in ((pa, sta, bna), (pb, stb, bnb)):
        part, r = _shrink_partials(part, rows, 2 * C, dev)
        gb = torch.empty((5, C), dtype=torch.float32, device=dev)
        dgam, dbet = gb[0], gb[1]
        if _ARENA[0] is not None:
            dgam, dbet = _grad_buf(bn.weight, (C,)), _grad_buf(bn.bias, (C,))
        check(L.yv1_bn_bwd_finalize(ptr(part), r, C, float(ya.npix), ptr(bn.weight), ptr(st.invstd), ptr(dgam), ptr(dbet),
                                    ptr(gb[2]), ptr(gb[3]), ptr(gb[4]), s), "yv1_bn_bwd_finalize")
        out.append((dgam, dbet))
        ks.append(gb)
    check(L.yv1_bn_bwd_apply_dual(dz.p, dz.ld, mask.p, mask.ld, ya.p, ya.ld, ptr(sta.mean), ptr(sta.invstd), ptr(ks[0][2]),
                                  ptr(ks[0][3]), ptr(ks[0][4]), dya.p, dya.ld, yb.p, yb.ld, ptr(stb.mean), ptr(stb.invstd),
                                  ptr(ks[1][2]), ptr(ks[1][3]), ptr(ks[1][4]), dyb.p, dyb.ld, ya.npix, C, 3, s),
          "yv1_bn_bwd_apply_dual")
    return out[0], out[1]


# ------------------------------------------------------------------ pooling / head
def maxpool_fwd(x, y, want_index=False):
    """3x3/2 max pool; with want_index returns the uint8 first-argmax tensor the backward consumes."""
    idx = torch.empty((y.N, y.H, y.W, x.C), dtype=torch.uint8, device=x.t.device) if want_index else None
    check(lib().yv1_maxpool3x3s2_fwd(x.p, x.ld, y.p, y.ld, ptr(idx), x.N, x.H, x.W, x.C, stream_ptr(x.t.device)),
          "yv1_maxpool3x3s2_fwd")
    return idx


def bn_act_maxpool_fwd(y, st, out, relu=True, want_index=False):
    """out = maxpool3x3s2(bf16(relu?(BN(y)))) without materialising the BatchNorm output (stems); returns the argmax
    index tensor ``bn_backward(..., pool_idx=)`` consumes when ``want_index``."""
    idx = torch.empty((out.N, out.H, out.W, y.C), dtype=torch.uint8, device=y.t.device) if want_index else None
    check(lib().yv1_bn_act_maxpool3x3s2_fwd(y.p, y.ld, ptr(st.scale), ptr(st.shift), 1 if relu else 0, out.p, out.ld, ptr(idx),
                                            y.N, y.H, y.W, y.C, stream_ptr(y.t.device)), "yv1_bn_act_maxpool3x3s2_fwd")
    return idx


def maxpool_bwd(x, dy, dx, idx=None):
    """x: the forward input (used when idx is None); dx has x's geometry."""
    check(lib().yv1_maxpool3x3s2_bwd(x.p if x is not None else None, x.ld if x is not None else 0, ptr(idx), dy.p, dy.ld,
                                     dx.p, dx.ld, dx.N, dx.H, dx.W, dx.C, stream_ptr(dx.t.device)), "yv1_maxpool3x3s2_bwd")


def avgpool_fwd(x, y):
    check(lib().yv1_avgpool2_fwd(x.p, x.ld, y.p, y.ld, x.N, x.H, x.W, x.C, stream_ptr(x.t.device)), "yv1_avgpool2_fwd")


def avgpool_bwd(dy, dx):
    check(lib().yv1_avgpool2_bwd(dy.p, dy.ld, dx.p, dx.ld, dx.N, dx.H, dx.W, dx.C, stream_ptr(dx.t.device)), "yv1_avgpool2_bwd")


def head_fwd(y, st, C):
    """sigmoid(bn_end(y)) as fp32 [N,H,W,C] (the NHWC tensor the reference gets from permute)."""
    dev = y.t.device
    out = torch.empty((y.N, y.H, y.W, C), dtype=torch.float32, device=dev)
    check(lib().yv1_head_sigmoid_fwd(y.p, y.ld, ptr(st.scale), ptr(st.shift), ptr(out), y.npix, C, stream_ptr(dev)),
          "yv1_head_sigmoid_fwd")
    return out


def head_bwd(dout, out, y, st, bn, dy):
    dev = y.t.device
    C = out.shape[-1]
    if _ARENA[0] is not None and bn.weight.numel() == C:
        dgam, dbet = _grad_buf(bn.weight, (C,)), _grad_buf(bn.bias, (C,))
    else:
        gb = torch.empty((2, C), dtype=torch.float32, device=dev)
        dgam, dbet = gb[0], gb[1]
    dout = dout.to(dtype=torch.float32).contiguous()
    check(lib().yv1_head_sigmoid_bwd(ptr(dout), ptr(out), y.p, y.ld, ptr(bn.weight), ptr(st.mean), ptr(st.invstd), dy.p,
                                     dy.ld, ptr(dgam), ptr(dbet), y.npix, C, stream_ptr(dev)), "yv1_head_sigmoid_bwd")
    return dgam, dbet
