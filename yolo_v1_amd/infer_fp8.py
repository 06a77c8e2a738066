"""FP8 inference executor for the ResNet-50 backbone -- BASELINE config 5 (S=14, fp8 MFMA conv, batched eval NMS).

Eval-mode forward of ``backbones/OriginResNet.py:173-195`` with every Bottleneck convolution
(:87-107) on the block-scaled fp8 MFMA: one launch per convolution does conv + BatchNorm (running
statistics, folded into per-channel alpha/beta) + residual add + ReLU (``yv1_conv2d_fwd_nhwc_fp8``).
What is stored where:
  * activations that only feed convolutions: e4m3, scale 1 (post-ReLU BatchNorm outputs);
  * the residual stream (block outputs, projection-shortcut outputs): bf16, plus an e4m3 copy of each
    block output for the next block's convolutions -- both written by the same epilogue;
  * weights: e4m3 with one power-of-two scale per output channel, folded into alpha;
  * the 7x7 stem (3 input channels) and the 30-channel head BatchNorm + sigmoid stay on the bf16 / fp32
    kernels of the training path.
The result feeds ``utils.utils.decode_batch`` (batched decoder + NMS) exactly like the bf16 eval path.
No autograd: this is an inference executor.
"""
import torch

from . import ops
from ._lib import check, lib, ptr, require_cuda, stream_ptr
from .engine import ConvParam


class Act8:
    """NHWC e4m3 activation [N,H,W,C] held as uint8."""
    __slots__ = ("t", "N", "H", "W", "C", "ld", "p")

    def __init__(self, N, H, W, C, device, tensor=None):
        self.t = torch.empty((N, H, W, C), dtype=torch.uint8, device=device) if tensor is None else tensor
        self.N, self.H, self.W, self.C, self.ld = N, H, W, C, C
        self.p = self.t.data_ptr()

    @classmethod
    def from_tensor(cls, t):
        """t: contiguous uint8 [N,H,W,C] device tensor holding e4m3 bytes."""
        if t.dtype != torch.uint8 or not t.is_contiguous() or t.dim() != 4:
            raise ValueError("Act8.from_tensor: contiguous uint8 [N,H,W,C]")
        return cls(t.shape[0], t.shape[1], t.shape[2], t.shape[3], t.device, t)

    @property
    def npix(self):
        return self.N * self.H * self.W


class Fp8Conv:
    """e4m3 weights [Opad][taps][I] + folded (alpha, beta) of one convolution and the BatchNorm behind it."""

    def __init__(self, conv, bn=None, fold_bn=True):
        assert isinstance(conv, ConvParam)
        w = conv.weight.detach()
        dev = w.device
        self.k, self.stride, self.pad = conv.kernel_size, conv.stride, conv.padding
        self.O, self.I = w.shape[0], w.shape[1]
        self.Opad = (self.O + 63) // 64 * 64
        if self.I % 64:
            raise ValueError("fp8 convolution needs Cin % 64 == 0 (got %d)" % self.I)
        s = stream_ptr(dev)
        L = lib()
        self.w8 = torch.empty((self.Opad, self.k * self.k, self.I), dtype=torch.uint8, device=dev)
        self.q = torch.empty(self.Opad, dtype=torch.float32, device=dev)
        so, si, sh, sw = w.stride()
        check(L.yv1_prep_weights_fp8(ptr(w), so, si, sh, sw, self.O, self.I, self.k, self.Opad, self.I, ptr(self.w8),
                                     ptr(self.q), s), "yv1_prep_weights_fp8")
        self.alpha = torch.empty(self.Opad, dtype=torch.float32, device=dev)
        self.beta = torch.empty(self.Opad, dtype=torch.float32, device=dev)
        st = ops.bn_eval_state(bn) if (bn is not None and fold_bn) else None
        check(L.yv1_fp8_fold_bn(ptr(st.scale) if st is not None else None, ptr(st.shift) if st is not None else None,
                                ptr(self.q), self.O, self.Opad, ptr(self.alpha), ptr(self.beta), s), "yv1_fp8_fold_bn")


def quantize(x, out=None):
    """bf16 Act -> e4m3 Act8 (saturating)."""
    dev = x.t.device
    out = out or Act8(x.N, x.H, x.W, x.C, dev)
    check(lib().yv1_quantize_bf16_to_fp8(x.p, x.ld, out.p, out.ld, x.npix, x.C, stream_ptr(dev)), "yv1_quantize_bf16_to_fp8")
    return out


def conv8(x8, w, relu, out16=None, out8=None, residual=None):
    """One fused launch; ``out16`` (bf16 Act) and/or ``out8`` (Act8) receive the result."""
    dev = x8.t.device
    if x8.C != w.I:
        raise ValueError("conv8: input has %d channels, the weights expect %d" % (x8.C, w.I))
    check(lib().yv1_conv2d_fwd_nhwc_fp8(x8.p, ptr(w.w8), ptr(w.alpha), ptr(w.beta),
                                        residual.p if residual is not None else None,
                                        residual.ld if residual is not None else 0,
                                        out16.p if out16 is not None else None, out16.ld if out16 is not None else 0,
                                        out8.p if out8 is not None else None, out8.ld if out8 is not None else 0,
                                        x8.N, x8.H, x8.W, x8.ld, w.I, w.Opad, w.k, w.stride, w.pad, 1 if relu else 0,
                                        stream_ptr(dev)), "yv1_conv2d_fwd_nhwc_fp8")


class ResNetFp8:
    """``ResNetFp8(net)(images)`` -> pred [N,S,S,B*5+C] fp32, ``net`` a yolo_v1_amd ResNet on the GPU.
    Weights are quantised when the executor is built; call ``refresh()`` after the parameters change."""

    def __init__(self, net):
        self.net = net
        self.trace = None        # set to a list to collect (label, tensor) per block (debugging / layer-wise tests)
        self.refresh()

    def refresh(self):
        net = self.net
        self.blocks = []
        for blk in net._blocks():
            ds = blk.downsample
            self.blocks.append((blk, Fp8Conv(blk.conv1, blk.bn1), Fp8Conv(blk.conv2, blk.bn2), Fp8Conv(blk.conv3, blk.bn3),
                                Fp8Conv(ds[0], ds[1]) if ds is not None else None))
        self.head = Fp8Conv(net.layer6, None)         # bn_end + sigmoid run in the fp32 head kernel
        self.stem_bn = ops.bn_eval_state(net.bn1)
        self.head_bn = ops.bn_eval_state(net.bn_end)

    def run_block(self, bi, x8, x16):
        """One Bottleneck: (e4m3 input, bf16 copy of it or None) -> (e4m3 output, bf16 copy when the next block's
        shortcut is an identity).  Four (three without projection) fused launches."""
        blk, c1, c2, c3, cd = self.blocks[bi]
        dev = x8.t.device
        N = x8.N
        planes = blk.conv1.out_channels
        z1 = Act8(N, x8.H, x8.W, planes, dev)
        conv8(x8, c1, True, out8=z1)
        h2, w2 = ops.conv_out_hw(x8.H, x8.W, 3, blk.stride, 1)
        z2 = Act8(N, h2, w2, planes, dev)
        conv8(z1, c2, True, out8=z2)
        cout = c3.O
        if cd is not None:
            res = ops.new_act(N, h2, w2, cout, dev)
            conv8(x8, cd, False, out16=res)
        else:
            if x16 is None:
                raise ValueError("block %d has an identity shortcut and needs the bf16 copy of its input" % bi)
            res = x16
        nxt_identity = bi + 1 < len(self.blocks) and self.blocks[bi + 1][4] is None
        out8 = Act8(N, h2, w2, cout, dev)
        out16 = ops.new_act(N, h2, w2, cout, dev) if nxt_identity else None
        conv8(z2, c3, True, out16=out16, out8=out8, residual=res)
        return out8, out16

    @torch.no_grad()
    def __call__(self, images):
        net = self.net
        require_cuda(images)
        dev = images.device
        N, _, H, W = images.shape
        if H % 64 or W % 64:
            raise ValueError("image sides must be multiples of 64")
        # stem on the bf16 path (3 input channels): conv 7x7/2 -> BN -> ReLU -> maxpool, then one quantisation
        w0 = net.cw(net.conv1, stem=True)
        xp = ops.pack_input(images)
        y0 = ops.new_act(N, H // 2, W // 2, 64, dev)
        ops.stem_fwd_bn_act(xp, w0, y0, H, W, self.stem_bn, relu=True)      # BatchNorm + ReLU in the stem's epilogue
        pooled = ops.new_act(N, H // 4, W // 4, 64, dev)
        ops.maxpool_fwd(y0, pooled)
        x8 = quantize(pooled)
        if self.trace is not None:
            self.trace.append(("stem", x8.t, None))
        x16 = None                                     # bf16 copy of the block input, kept when it is an identity shortcut
        for bi in range(len(self.blocks)):
            x8, x16 = self.run_block(bi, x8, x16)
            if self.trace is not None:
                self.trace.append(("block%d" % bi, x8.t, x16.t if x16 is not None else None))
        yh = ops.new_act(N, x8.H, x8.W, self.head.Opad, dev)
        conv8(x8, self.head, False, out16=yh)
        return ops.head_fwd(yh, self.head_bn, net.out_channels)
