"""GPU parity of the fp8 (e4m3) inference convolution path -- BASELINE config 5 -- against oracle/fp8.py.

Tolerances (stated here as the task asks): weight / activation quantisation is pure elementwise arithmetic on identical
inputs and must be BIT-EXACT; the fused convolution differs from the oracle only in fp32 summation order (and one FMA),
so a bf16 output may differ by one bf16 ulp (2^-8 relative) on a small fraction of elements and an e4m3 output by one
e4m3 ulp (2^-3 relative) on a smaller one; the whole network (54 quantised layers) is checked on the sigmoid outputs.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def _as_f32(u8):
    return u8.view(torch.float8_e4m3fn).to(torch.float32)


def test_quantisers_bit_exact(dev):
    from oracle import fp8 as o8
    from yolo_v1_amd import infer_fp8, ops
    from yolo_v1_amd.engine import ConvParam
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 5, 7, 64, generator=g) * torch.tensor([1e-3, 0.1, 1.0, 30.0, 600.0]).view(1, 5, 1, 1)
    x[0, 0, 0, :8] = torch.tensor([0.0, -0.0, 448.0, 464.0, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, -500.0])
    xa = ops.Act(x.to(torch.bfloat16).to(dev))
    got = infer_fp8.quantize(xa)
    want = o8.e4m3(x.to(torch.bfloat16).to(torch.float32))
    assert torch.equal(_as_f32(got.t.cpu()), want)
    for (O, I, k) in [(64, 64, 1), (128, 64, 3), (30, 2048, 1), (256, 128, 3)]:
        conv = ConvParam(I, O, k, 1, k // 2)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(O, I, k, k, generator=g) * (torch.rand(O, 1, 1, 1, generator=g) * 0.2 + 1e-3))
            conv.weight[0].zero_()                       # an all-zero filter: q = 1, zeros
        conv = conv.to(dev)
        fw = infer_fp8.Fp8Conv(conv, None)
        w8, q = o8.quantize_weight(conv.weight.detach().cpu())
        assert torch.equal(fw.q[:O].cpu(), q)
        got_w = _as_f32(fw.w8.cpu())[:O].view(O, k, k, I).permute(0, 3, 1, 2)
        assert torch.equal(got_w, w8)
        assert float(_as_f32(fw.w8.cpu())[O:].abs().sum()) == 0.0


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, residual, relu, dual
    (2, 16, 16, 64, 64, 1, 1, False, True, False),
    (2, 16, 16, 64, 64, 3, 1, False, True, False),
    (2, 16, 16, 64, 256, 1, 1, True, True, True),
    (3, 14, 14, 256, 128, 1, 1, False, True, False),
    (2, 28, 28, 128, 128, 3, 2, False, True, False),
    (2, 28, 28, 256, 512, 1, 2, False, False, True),
    (2, 14, 14, 512, 2048, 1, 1, True, True, True),
    (5, 7, 7, 2048, 30, 1, 1, False, False, True),
    (1, 9, 11, 128, 192, 3, 1, True, False, True),          # ragged M, Cout that is only a multiple of 64
    (70, 14, 14, 128, 128, 3, 1, False, True, False),       # enough tiles for the 128x128 configuration
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_fused_fp8_conv_matches_oracle(dev, case):
    from oracle import fp8 as o8
    from yolo_v1_amd import infer_fp8, ops
    from yolo_v1_amd.engine import ConvParam
    N, H, W, Cin, Cout, k, stride, use_res, relu, dual = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    pad = k // 2
    conv = ConvParam(Cin, Cout, k, stride, pad)
    bn = torch.nn.BatchNorm2d(Cout)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
        bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cout, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    conv, bn = conv.to(dev), bn.to(dev)
    x = torch.relu(torch.randn(N, H, W, Cin, generator=g)) * 1.5
    x8 = infer_fp8.quantize(ops.Act(x.to(torch.bfloat16).to(dev)))
    fw = infer_fp8.Fp8Conv(conv, bn)
    Ho, Wo = ops.conv_out_hw(H, W, k, stride, pad)
    res = torch.randn(N, Ho, Wo, fw.Opad, generator=g).to(torch.bfloat16) if use_res else None
    res_act = ops.Act(res.to(dev)) if use_res else None
    out16 = ops.new_act(N, Ho, Wo, fw.Opad, dev)
    out8 = infer_fp8.Act8(N, Ho, Wo, fw.Opad, dev) if dual else None
    infer_fp8.conv8(x8, fw, relu, out16=out16, out8=out8, residual=res_act)
    P = {"bn." + n: t.detach().cpu() for n, t in list(bn.named_parameters()) + list(bn.named_buffers())}
    scale, shift = o8.bn_coeffs(P, "bn")
    xin = _as_f32(x8.t.cpu()).permute(0, 3, 1, 2)
    rin = res.to(torch.float32)[..., :Cout].permute(0, 3, 1, 2) if use_res else None
    w16, w8 = o8.conv_fused(xin, conv.weight.detach().cpu(), scale, shift, residual=rin, relu=relu, stride=stride, padding=pad)
    got16 = out16.t.cpu().to(torch.float32)[..., :Cout].permute(0, 3, 1, 2)
    diff = (got16 - w16).abs()
    # Tolerance: one bf16 ulp of the result (<= 2^-7 relative), plus fp32 summation-order noise, which scales with the
    # sum of |terms| (not with the possibly cancelled result): 8e-6 * sum|x*w| * |alpha| covers both the MFMA's and
    # F.conv2d's accumulation; with a residual the intermediate bf16(t) may flip too: one more ulp of |t| <= |out|+|res|.
    w8q, qv = o8.quantize_weight(conv.weight.detach().cpu())
    mag = torch.nn.functional.conv2d(xin.abs(), w8q.abs(), stride=stride, padding=pad) * (scale / qv).abs().view(1, -1, 1, 1)
    ulp = w16.abs() * 2.0 ** -7 + 8e-6 * mag + 1e-7
    if use_res:
        ulp = ulp + (w16.abs() + rin.abs()) * 2.0 ** -7
    assert bool((diff <= ulp).all()), "bf16 output off by more than one ulp: max %g" % float((diff / ulp).max())
    assert float((diff > 0).float().mean()) < 0.02
    if Cout < fw.Opad:
        pad_part = out16.t.cpu().to(torch.float32)[..., Cout:]
        expect_pad = res.to(torch.float32)[..., Cout:] if use_res else torch.zeros_like(pad_part)
        if relu:
            expect_pad = expect_pad.clamp_min(0)
        assert torch.equal(pad_part, expect_pad.to(torch.bfloat16).to(torch.float32))
    if dual:
        got8 = _as_f32(out8.t.cpu())[..., :Cout].permute(0, 3, 1, 2)
        d8 = (got8 - w8).abs()
        # e4m3 of two bf16 values within `ulp` of each other: at most one e4m3 step apart (2^-3 relative, 2^-9 absolute
        # in the denormal range) unless the bf16 values straddle more, which `ulp` bounds
        assert bool((d8 <= torch.maximum(w8.abs() * 0.126, torch.tensor(2.0 ** -9 * 1.01)) + ulp).all())
        assert float((d8 > 0).float().mean()) < 0.01
        # and the e4m3 output is exactly the quantisation of the kernel's own bf16 output
        assert torch.equal(_as_f32(out8.t.cpu()), o8.e4m3(out16.t.cpu().to(torch.float32)))


def _prepared_net(S, dev):
    """Random-init ResNet whose running statistics come from one training-mode pass of the fp32 oracle (momentum 1), so
    the eval-mode activations are scaled the way a trained checkpoint's are; bn3.weight x0.2 keeps the residual stream
    contractive (same device as tests/test_gpu_resnet.py)."""
    from oracle import backbones as ob
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    torch.manual_seed(0)
    net = resnet50(S=S)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("bn3.weight"):
                p.mul_(0.2)
    P = {k: v.detach().clone() for k, v in net.state_dict().items()}
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(7))
    old = ob._bn.__defaults__
    ob._bn.__defaults__ = (1.0, 1e-5)                      # momentum 1: running stats := this batch's statistics
    try:
        with torch.no_grad():
            ob.resnet50_forward(x, P, S=S, training=True)
    finally:
        ob._bn.__defaults__ = old
    net.load_state_dict(P)
    return net.to(dev).eval(), P, x


@pytest.mark.parametrize("S", [14, 7])
def test_resnet50_fp8_blocks_teacher_forced(dev, S):
    """Every Bottleneck of the fp8 executor fed the ORACLE's block input: with identical inputs only the rounding flips
    of one block remain (fp32 summation order), so the e4m3 outputs must agree on all but a few percent of the elements (three
    quantised convolutions per block, K up to 4608) and differ by one e4m3 step where they do not."""
    from oracle import fp8 as o8
    from yolo_v1_amd import ops
    from yolo_v1_amd.infer_fp8 import Act8, ResNetFp8
    net, P, x = _prepared_net(S, dev)
    trace = []
    with torch.no_grad():
        o8.resnet50_eval_fp8(x, P, S, trace=trace)
    eng = ResNetFp8(net)
    eng.trace = []
    eng(x.to(dev))
    stem_d = (_as_f32(eng.trace[0][1].cpu()).permute(0, 3, 1, 2) - trace[0][1]).abs()
    assert float((stem_d > 0).float().mean()) < 2e-3            # bf16 stem (summation-order flips only) + quantiser
    to_u8 = lambda t: t.permute(0, 2, 3, 1).contiguous().to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
    for bi in range(len(eng.blocks)):
        _, in8, in16 = trace[bi]
        _, want8, want16 = trace[bi + 1]
        x16 = ops.Act(in16.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)) if in16 is not None else None
        out8, out16 = eng.run_block(bi, Act8.from_tensor(to_u8(in8)), x16)
        got8 = _as_f32(out8.t.cpu()).permute(0, 3, 1, 2)
        d = (got8 - want8).abs()
        frac = float((d > 0).float().mean())
        assert frac < 0.05, "block %d: %.4f of the e4m3 outputs differ" % (bi, frac)
        step = torch.maximum(want8.abs() * 0.126, torch.tensor(2.0 ** -9 * 1.01))
        # flips are single steps, except where a flipped z1/z2 element inside the block moved several outputs a little
        assert float((d > 2 * step).float().mean()) < 5e-3, "block %d" % bi
        if out16 is not None:
            got16 = out16.t.cpu().to(torch.float32).permute(0, 3, 1, 2)
            assert float((got16 - want16).abs().mean()) < 1e-2 * float(want16.abs().mean()), "block %d" % bi


@pytest.mark.parametrize("S", [14, 7])
def test_resnet50_fp8_inference_end_to_end(dev, S):
    """Whole eval-mode network.  Run end to end the rounding flips of ~50 quantised layers compound (each flip moves an
    activation by 6-12 %), so executor and oracle are compared in distribution: both must sit at the same distance
    from the fp32 network, and no further from each other than from it."""
    from oracle import backbones as ob, fp8 as o8
    from yolo_v1_amd.infer_fp8 import ResNetFp8
    net, P, x = _prepared_net(S, dev)
    with torch.no_grad():
        want8 = o8.resnet50_eval_fp8(x, P, S)
        want32 = ob.resnet50_forward(x, P, S=S, training=False)
    eng = ResNetFp8(net)
    got = eng(x.to(dev)).cpu()
    assert got.shape == want8.shape and bool(torch.isfinite(got).all())
    e_kernel = float((got - want32).abs().mean())
    e_oracle = float((want8 - want32).abs().mean())
    assert abs(e_kernel - e_oracle) < 0.25 * e_oracle + 2e-3, (e_kernel, e_oracle)
    assert float((got - want8).abs().mean()) < 1.2 * e_oracle + 2e-3
    assert e_kernel < 0.12                                  # what e4m3 storage costs on a random-weight network
    from yolo_v1_amd.utils.utils import decode_batch
    boxes, cls, scores, keep, counts, ncand = decode_batch(got.to(dev), grid_num=got.shape[1], B=2, thresh=0.1, nms_th=0.5)
    assert counts.shape[0] == 2


# ------------------------------------------------------------------ training with fp8 forward GEMMs
@pytest.mark.parametrize("N,H,Wd,Cin,Cout,k,stride", [(2, 16, 16, 64, 64, 1, 1), (2, 16, 16, 64, 256, 1, 1), (9, 28, 28, 128, 128, 3, 1),
                                                      (3, 28, 28, 256, 512, 1, 2), (2, 14, 14, 256, 256, 3, 2), (70, 14, 14, 512, 128, 1, 1)])
def test_fp8_training_conv_output_and_batch_statistics(dev, N, H, Wd, Cin, Cout, k, stride):
    """Training form of the fp8 convolution: y = bf16(acc / q) and the BatchNorm statistic partials, against fp32 math on the
    same quantised operands; the multi-tensor weight quantiser bit-exact against the oracle."""
    from oracle import fp8 as o8
    from yolo_v1_amd import ops
    from yolo_v1_amd.engine import ConvParam
    g = torch.Generator().manual_seed(N + Cin + Cout + k)
    pad = k // 2
    conv = ConvParam(Cin, Cout, k, stride, pad)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
    conv = conv.to(dev)
    w8 = ops.Fp8Weights(conv.weight, k, stride, pad)
    ops.refresh_many_fp8([w8])
    want_w, q = o8.quantize_weight(conv.weight.detach().cpu())
    assert torch.equal(_as_f32(w8.w8.cpu())[:Cout].view(Cout, k, k, Cin).permute(0, 3, 1, 2), want_w)
    assert torch.equal(w8.alpha[:Cout].cpu(), 1.0 / q)
    x = torch.relu(torch.randn(N, H, Wd, Cin, generator=g)) * 1.5
    x8 = ops.quantize_fp8(ops.Act(x.to(torch.bfloat16).to(dev)))
    Ho, Wo = ops.conv_out_hw(H, Wd, k, stride, pad)
    y = ops.new_act(N, Ho, Wo, w8.Opad, dev)
    stats = ops.conv_fwd_fp8(x8, w8, y, True)
    xin = _as_f32(x8.t.cpu()).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xin, want_w / q.view(-1, 1, 1, 1), stride=stride, padding=pad)
    got = y.t.cpu().to(torch.float32)[..., :Cout].permute(0, 3, 1, 2)
    mag = torch.nn.functional.conv2d(xin.abs(), (want_w / q.view(-1, 1, 1, 1)).abs(), stride=stride, padding=pad)
    refb = o8.bf16(ref)                       # rounded vs rounded: a rounding flip is exactly one bf16 ulp (<= 2^-7 relative)
    # the block-scaled MFMA's accumulation noise, measured: up to ~2e-5 of sum|x*w| (the bf16 MFMA stays below 8e-6)
    assert bool(((got - refb).abs() <= refb.abs() * 2.0 ** -7 + 4e-5 * mag + 1e-7).all())
    s = stats.sum(0).cpu()
    np.testing.assert_allclose(s[0, :Cout].numpy(), ref.sum((0, 2, 3)).numpy(), rtol=2e-3, atol=2e-2 * float(ref.abs().max()))
    np.testing.assert_allclose(s[1, :Cout].numpy(), (ref * ref).sum((0, 2, 3)).numpy(), rtol=2e-3, atol=1e-2)


def test_resnet50_training_step_with_fp8_forward(dev):
    """Whole network, training mode, fp8 forward GEMMs / bf16 backward: outputs against the oracle with the same operand
    quantisation (straight-through backward), and the parameter gradients by direction and norm.  Same contractive
    weight regime and the same kind of bounds as the bf16 whole-network test (tests/test_gpu_resnet.py)."""
    from oracle import backbones as ob, fp8 as o8
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    S, N, hw = 14, 4, 192
    P = ob.init_params(ob.resnet50_param_shapes(S), "resnet", seed=11)
    for k in P:
        if k.endswith("bn3.weight"):
            P[k] = P[k] * 0.2
    net = resnet50(S=S)
    net.load_state_dict(P, strict=True)
    net = net.to(dev).train()
    net.fp8_forward = True
    x = torch.randn(N, 3, hw, hw, generator=torch.Generator().manual_seed(1))
    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    ref = ob.resnet50_forward(x, P, S, training=True, q=ob.bf16_ste, qconv=o8.fp8_forward_ste)
    gup = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3)) * 1e-2
    ref.backward(gup)
    pred = net(x.to(dev))
    d = (pred.detach().cpu() - ref.detach()).abs()
    # e4m3 rounding flips (6-12 % of an activation each) random-walk through 53 quantised convolutions and as many
    # small-batch BatchNorms: measured mean 0.04 / max 0.21 (bf16 whole-network test: 0.007 / 0.04)
    assert d.max().item() <= 0.35 and d.mean().item() <= 6e-2, (d.max().item(), d.mean().item())
    pred.backward(gup.to(dev))
    sd = dict(net.named_parameters())
    cos = lambda a, b: float((a.double().flatten() @ b.double().flatten()) / (a.double().norm() * b.double().norm() + 1e-30))
    for k in ("layer4.2.conv3.weight", "layer4.0.conv2.weight", "layer3.3.conv1.weight", "layer2.1.conv2.weight", "layer6.weight"):
        a, b = sd[k].grad.detach().cpu(), P[k].grad
        assert cos(a, b) >= 0.6, (k, cos(a, b))      # measured 0.67-0.99: the two forwards have drifted apart by then
        assert 0.7 <= float(a.norm() / b.norm()) <= 1.4, (k, float(a.norm() / b.norm()))
    # and a bf16 step of the same network is untouched by the flag being off
    net.fp8_forward = False
    assert torch.isfinite(net(x.to(dev))).all()
