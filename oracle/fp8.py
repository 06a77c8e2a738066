"""Oracle (test infrastructure only): fp8 (OCP e4m3) inference emulation of the ResNet-50 YOLO backbone.

CPU restatement of the eval-mode forward of backbones/OriginResNet.py:87-107, :173-195 with the storage and
rounding points of the HIP fp8 executor (yolo_v1_amd/infer_fp8.py) made explicit with torch's own
``float8_e4m3fn`` / ``bfloat16`` casts:
  * conv operands: e4m3 activations (scale 1, saturating at +-448) x e4m3 weights pre-multiplied per output
    channel by q = 2^floor(log2(448/amax)); fp32 accumulation (F.conv2d on the dequantised values: every product is
    exact in fp32).  The hardware's adder is NOT that of fp32: inside a group of 8 products the fp8 MFMA truncates every
    product at 2^-13 of the group's largest one (tools/fp8_accum_probe.hip, measured on MI355X); this oracle keeps the
    exact fp32 sum as the reference value and the GPU tests bound the difference by 2^-13 x sum|products|
    (tests/test_gpu_bench_configs_fp8.py);
  * t = bf16(acc * (gamma*rsqrt(var+eps)/q) + (beta - mean*gamma*rsqrt(var+eps))), out = relu(t + residual),
    stored as bf16 and as e4m3(bf16(out));
  * stem and head as in the bf16 path (bf16 storage emulation).
The conv/BN arithmetic itself is PyTorch ATen's (not in the reference tree): parity for this configuration is
pinned against torch 2.10 CPU, like oracle/backbones.py; the reference publishes no fp8 numbers ("parity
unpinned upstream").
"""
import torch
import torch.nn.functional as F

EPS = 1e-5


def bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def e4m3(t):
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


def quantize_weight(w):
    """OIHW fp32 -> (e4m3 values of w*q as fp32, q [O]) with q the largest power of two with amax*q <= 448."""
    amax = w.abs().amax(dim=(1, 2, 3))
    ratio = torch.tensor(448.0) / amax
    _, e = torch.frexp(ratio)                          # ratio = m * 2^e, m in [0.5, 1)
    q = torch.where((amax > 0) & torch.isfinite(ratio), torch.ldexp(torch.ones_like(amax), (e - 1).clamp(-100, 100)),
                    torch.ones_like(amax))
    return e4m3(w * q.view(-1, 1, 1, 1)), q


def fp8_forward_ste(conv_fn, x, w_master, w_stored):
    """``qconv`` hook for ``backbones.resnet50_forward``: training with fp8 forward GEMMs and a bf16 backward.
    Forward VALUE: the convolution of the e4m3 copy (scale 1) of the stored bf16 activation with the per-output-channel
    scaled e4m3 weights (quantised from the fp32 master weight, as the HIP path does).  GRADIENTS: those of the
    convolution of the stored bf16 operands -- the HIP backward reads the bf16 activations and weights."""
    y = conv_fn(x, w_stored)
    with torch.no_grad():
        w8, q = quantize_weight(w_master)
        yq = conv_fn(e4m3(bf16(x)), w8 / q.view(-1, 1, 1, 1))
    return y + (yq - y).detach()


def bn_coeffs(P, name):
    invstd = torch.rsqrt(P[name + ".running_var"] + EPS)
    scale = P[name + ".weight"] * invstd
    shift = P[name + ".bias"] - P[name + ".running_mean"] * P[name + ".weight"] * invstd
    return scale, shift


def conv_fused(x8, w, scale=None, shift=None, residual=None, relu=True, stride=1, padding=0):
    """x8: fp32 tensor holding e4m3 values (NCHW).  Returns (out_bf16, out_e4m3) as fp32 tensors."""
    w8, q = quantize_weight(w)
    acc = F.conv2d(x8, w8, stride=stride, padding=padding)
    alpha = (scale if scale is not None else torch.ones_like(q)) / q
    beta = shift if shift is not None else torch.zeros_like(q)
    t = bf16(acc * alpha.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1))
    if residual is not None:
        t = t + residual
    if relu:
        t = F.relu(t)
    out16 = bf16(t)
    return out16, e4m3(out16)


def resnet50_eval_fp8(x, P, S=7, trace=None):
    """x [N,3,H,W] fp32 -> pred [N,H/32 (S=14) or H/64 (S=7), ., B*5+C].  ``trace`` (a list) collects the e4m3
    block outputs as (label, e4m3 values, bf16 values), NCHW fp32 tensors."""
    y0 = F.conv2d(bf16(x), bf16(P["conv1.weight"]), stride=2, padding=3)      # BatchNorm + ReLU ride in the stem's epilogue:
    s0, b0 = bn_coeffs(P, "bn1")                                                 # one bf16 rounding, after the ReLU
    z0 = bf16(F.relu(y0 * s0.view(1, -1, 1, 1) + b0.view(1, -1, 1, 1)))
    x16 = F.max_pool2d(z0, 3, 2, 1)
    x8 = e4m3(x16)
    if trace is not None:
        trace.append(("stem", x8, None))
    stages = [("layer1", 3, 1), ("layer2", 4, 2), ("layer3", 6, 2), ("layer4", 3, 2)]
    if S == 7:
        stages.append(("layer5", 3, 2))
    for name, blocks, stride in stages:
        for i in range(blocks):
            p = "%s.%d" % (name, i)
            st = stride if i == 0 else 1
            _, z1 = conv_fused(x8, P[p + ".conv1.weight"], *bn_coeffs(P, p + ".bn1"))
            _, z2 = conv_fused(z1, P[p + ".conv2.weight"], *bn_coeffs(P, p + ".bn2"), stride=st, padding=1)
            if (p + ".downsample.0.weight") in P:
                res, _ = conv_fused(x8, P[p + ".downsample.0.weight"], *bn_coeffs(P, p + ".downsample.1"), relu=False,
                                    stride=st)
            else:
                res = x16
            x16, x8 = conv_fused(z2, P[p + ".conv3.weight"], *bn_coeffs(P, p + ".bn3"), residual=res)
            if trace is not None:
                trace.append((p, x8, x16))
    yh, _ = conv_fused(x8, P["layer6.weight"], relu=False)
    sh, bh = bn_coeffs(P, "bn_end")
    return torch.sigmoid(yh * sh.view(1, -1, 1, 1) + bh.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
