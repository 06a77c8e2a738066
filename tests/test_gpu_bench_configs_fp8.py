"""The fp8 (e4m3 MFMA) convolution kernels at the shapes BASELINE config 5 runs them ("the fp8 kernels are not
element-wise verified at the shapes the bench runs").

`bench.py`'s S=14 fp8 entries run ResNet-50 (no layer5: OriginResNet.py:131-132) at 448x448 with a batch of 64:
  * training with fp8 forward GEMMs  -> `yv1_conv2d_fwd_stats_nhwc_fp8` (raw bf16 output + BatchNorm statistic partials)
  * fp8 inference executor           -> `yv1_conv2d_fwd_nhwc_fp8`       (folded BatchNorm + residual + ReLU, bf16 / e4m3 out)
Every distinct Bottleneck convolution of that network + the head is launched here at batch 64 through the C ABI and
compared, EVERY output element, with fp32 `F.conv2d` on the same e4m3-quantised operands (oracle/fp8.py restates the
quantisers with torch's own float8_e4m3fn casts; products of two e4m3 values are exact in fp32, so only the summation
order differs).  The template each call launched is read back (`yv1_last_config`) and checked against the committed
checklist tests/golden/bench_kernel_templates.json (`fp8_bench_dispatched` = the k_conv_fp8 rows of
profiles/*_resnet50_S14_fp8_forward_kernel_summary_last_step.txt).

Tolerances: training form -- bf16 output vs fp32 math on identical operands: rtol 1e-2, atol 1e-2 x max|ref| (final bf16
rounding + summation order); statistic partials rtol 2e-3 (fp32 summation order over up to 802 816 pixels).  Inference
form -- one bf16 ulp of the result plus what the fp8 MFMA's own adder costs, e4m3 output == e4m3 of the kernel's own bf16
output, bit for bit.

The fp8 MFMA does NOT add its 64 products like fp32 (tools/fp8_accum_probe.hip, run on MI355X: one product 2^8 and 63
products 2^-s -- exact for s <= 5, from s = 6 on the 7 small products that share an adder group of 8 with the large one are
lost): inside a group of 8 products every product is TRUNCATED at 2^-13 of the group's largest one before the group sums
are added (those, and the accumulator input, exactly).  A product can therefore lose up to 2^-13 x (largest product of its
group) <= 2^-13 x sum|products|: that is the HARD bound used below (2^-13 x sum|x||w| x |alpha|).  It is rarely approached:
at batch 64 (5e7 - 2e8 outputs per layer) 2 - 700 outputs per layer exceed the fp32-re-association bound 8e-6 x sum|x||w| of
tests/test_gpu_fp8.py, the worst by a factor 3.3 -- so the DISTRIBUTION is held to that tight bound as well: at most 1e-4
of the outputs may exceed it.
"""
import json
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BATCH = 64

# (Cin, Cout, k, stride, Hin): the distinct Bottleneck convolutions of ResNet-50 S=14 at 448x448 + head
RESNET_S14 = [(64, 64, 1, 1, 112), (64, 64, 3, 1, 112), (64, 256, 1, 1, 112), (256, 64, 1, 1, 112), (256, 128, 1, 1, 112),
              (128, 128, 3, 2, 112), (128, 512, 1, 1, 56), (256, 512, 1, 2, 112), (512, 128, 1, 1, 56), (128, 128, 3, 1, 56),
              (512, 256, 1, 1, 56), (256, 256, 3, 2, 56), (256, 1024, 1, 1, 28), (512, 1024, 1, 2, 56), (1024, 256, 1, 1, 28),
              (256, 256, 3, 1, 28), (1024, 512, 1, 1, 28), (512, 512, 3, 2, 28), (512, 2048, 1, 1, 14), (1024, 2048, 1, 2, 28),
              (2048, 512, 1, 1, 14), (512, 512, 3, 1, 14), (2048, 30, 1, 1, 14)]
SEEN = {}


def _note(kind, shape, cfgs):
    for c in cfgs:
        SEEN.setdefault(c, []).append("%s %s" % (kind, shape))


def _f32(u8):
    return u8.view(torch.float8_e4m3fn).to(torch.float32)


def _operands(Cin, Cout, k, H):
    """Post-ReLU-like bf16 activations (what the e4m3 copies are taken of) and kaiming-scaled weights."""
    g = torch.Generator().manual_seed(Cin * 5 + Cout * 11 + k + H)
    x = (torch.relu(torch.randn(BATCH, H, H, Cin, generator=g)) * 1.5).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    return g, x, w


def _cmp(got_dev, ref_cpu, rtol, atol_scale, what):
    ref = ref_cpu.to(DEV)
    atol = atol_scale * float(ref.abs().max()) + 1e-12
    err = (got_dev.float() - ref).abs()
    nbad = int((err > atol + rtol * ref.abs()).sum())
    assert nbad == 0, "%s: %d of %d elements off (max |err| %g, atol %g)" % (what, nbad, err.numel(), float(err.max()), atol)


@pytest.mark.parametrize("Cin,Cout,k,stride,H", RESNET_S14[:-1])       # the 30-channel head stays bf16 in training
def test_fp8_training_forward_batch64_layer(Cin, Cout, k, stride, H):
    """yv1_conv2d_fwd_stats_nhwc_fp8: e4m3 activations x per-channel-scaled e4m3 weights -> raw bf16 y + statistic partials."""
    from oracle import fp8 as o8
    from yolo_v1_amd import _lib, ops
    g, x, w = _operands(Cin, Cout, k, H)
    pad = 1 if k == 3 else 0
    shape = "%d->%d k%d s%d @%d" % (Cin, Cout, k, stride, H)
    param = torch.nn.Parameter(w.to(DEV).contiguous(memory_format=torch.channels_last))
    w8 = ops.Fp8Weights(param, k, stride, pad)
    ops.refresh_many_fp8([w8])
    x8 = ops.quantize_fp8(ops.Act(x.to(DEV)))
    # the quantisers are elementwise: bit-exact against torch's own casts
    xq = o8.e4m3(x.to(torch.float32))
    assert torch.equal(_f32(x8.t.cpu()), xq)
    wq, q = o8.quantize_weight(w)
    assert torch.equal(_f32(w8.w8.cpu())[:Cout].view(Cout, k, k, Cin).permute(0, 3, 1, 2), wq)
    assert torch.equal(w8.alpha[:Cout].cpu(), 1.0 / q)
    ref = F.conv2d(xq.permute(0, 3, 1, 2), wq, stride=stride, padding=pad) / q.view(1, -1, 1, 1)
    OH = ref.shape[2]
    ya = ops.new_act(BATCH, OH, OH, w8.Opad, DEV)
    stats = ops.conv_fwd_fp8(x8, w8, ya, True)
    cf = _lib.last_config()
    torch.cuda.synchronize()
    assert len(cf) == 1 and cf[0].startswith("k_conv_fp8<"), cf
    _note("fwd+stats", shape, cf)
    _cmp(ya.t[..., :Cout], ref.permute(0, 2, 3, 1), 1e-2, 1e-2, shape + " fp8 forward " + cf[0])
    if w8.Opad > Cout:
        assert float(ya.t[..., Cout:].float().abs().max()) == 0.0
    s = stats.sum(0)
    _cmp(s[0, :Cout], ref.sum((0, 2, 3)), 2e-3, 2e-3 * (BATCH * OH * OH) ** 0.5, shape + " stats sum")
    _cmp(s[1, :Cout], (ref * ref).sum((0, 2, 3)), 2e-3, 1e-4, shape + " stats sumsq")


@pytest.mark.parametrize("Cin,Cout,k,stride,H", RESNET_S14)
def test_fp8_inference_conv_batch64_layer(Cin, Cout, k, stride, H):
    """yv1_conv2d_fwd_nhwc_fp8: conv + folded eval BatchNorm (+ bf16 residual for the block-closing 1x1s) + ReLU."""
    from oracle import fp8 as o8
    from yolo_v1_amd import _lib, infer_fp8, ops
    from yolo_v1_amd.engine import ConvParam
    g, x, w = _operands(Cin, Cout, k, H)
    pad = 1 if k == 3 else 0
    shape = "%d->%d k%d s%d @%d" % (Cin, Cout, k, stride, H)
    head = Cout == 30
    closing = k == 1 and stride == 1 and Cout == 4 * Cin          # conv3 of a Bottleneck: residual + ReLU, both outputs
    conv = ConvParam(Cin, Cout, k, stride, pad)
    bn = torch.nn.BatchNorm2d(Cout)
    with torch.no_grad():
        conv.weight.copy_(w)
        bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cout, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    conv, bn = conv.to(DEV), bn.to(DEV)
    fw = infer_fp8.Fp8Conv(conv, None if head else bn)
    x8 = infer_fp8.quantize(ops.Act(x.to(DEV)))
    OH = ops.conv_out_hw(H, H, k, stride, pad)[0]
    res = torch.randn(BATCH, OH, OH, fw.Opad, generator=g).to(torch.bfloat16) if closing else None
    out16 = ops.new_act(BATCH, OH, OH, fw.Opad, DEV)
    out8 = infer_fp8.Act8(BATCH, OH, OH, fw.Opad, DEV) if (closing or not head) else None
    relu = not head
    infer_fp8.conv8(x8, fw, relu, out16=out16, out8=out8, residual=ops.Act(res.to(DEV)) if closing else None)
    cf = _lib.last_config()
    torch.cuda.synchronize()
    assert len(cf) == 1 and cf[0].startswith("k_conv_fp8<"), cf
    _note("infer", shape, cf)
    if head:
        scale = shift = None
    else:
        P = {"bn." + n: t.detach().cpu() for n, t in list(bn.named_parameters()) + list(bn.named_buffers())}
        scale, shift = o8.bn_coeffs(P, "bn")
    xin = _f32(x8.t.cpu()).permute(0, 3, 1, 2)
    rin = res.to(torch.float32)[..., :Cout].permute(0, 3, 1, 2) if closing else None
    w16, _ = o8.conv_fused(xin, w, scale, shift, residual=rin, relu=relu, stride=stride, padding=pad)
    got16 = out16.t[..., :Cout].permute(0, 3, 1, 2).float()
    wq, qv = o8.quantize_weight(w)
    alpha = (scale if scale is not None else torch.ones_like(qv)) / qv
    mag = F.conv2d(xin.abs(), wq.abs(), stride=stride, padding=pad) * alpha.abs().view(1, -1, 1, 1)
    ulp = w16.abs() * 2.0 ** -7 + 8e-6 * mag + 1e-7
    if closing:
        ulp = ulp + (w16.abs() + rin.abs()) * 2.0 ** -7
    diff = (got16 - w16.to(DEV)).abs()
    ulp, mag = ulp.to(DEV), mag.to(DEV)
    hard = ulp + 2.0 ** -13 * mag                       # + the fp8 MFMA's in-group truncation (module docstring)
    nbad = int((diff > hard).sum())
    assert nbad == 0, "%s %s: %d of %d bf16 outputs beyond one ulp + MFMA truncation (max ratio %g)" % (
        shape, cf[0], nbad, diff.numel(), float((diff / hard).max()))
    tail = float((diff > ulp).float().mean())
    assert tail <= 1e-4, "%s %s: %.3g of the outputs beyond one bf16 ulp + fp32 re-association" % (shape, cf[0], tail)
    assert float((diff > 0).float().mean()) < 0.02
    if out8 is not None:
        assert torch.equal(_f32(out8.t.cpu()), o8.e4m3(out16.t.cpu().to(torch.float32)))


def test_zz_every_fp8_bench_template_was_covered():
    """Runs last (file order): the k_conv_fp8 templates the batch-64 S=14 shapes dispatched, and the checklist."""
    from conftest import ROOT
    if not SEEN:
        pytest.skip("run the whole file: the coverage table is filled by the layer tests")
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_kernel_templates.json")))
    lines = ["%-40s %s" % (k, SEEN[k][0] + (" (+%d more)" % (len(SEEN[k]) - 1) if len(SEEN[k]) > 1 else "")) for k in sorted(SEEN)]
    print("\nfp8 kernel template                      covered by\n" + "\n".join(lines))
    if os.environ.get("YV1_DUMP_TEMPLATES"):
        json.dump({k: sorted(set(v.split(" ")[0] for v in SEEN[k])) for k in sorted(SEEN)},
                  open(os.environ["YV1_DUMP_TEMPLATES"] + ".fp8", "w"), indent=1)
    missing = [t for t in want["fp8_bench_dispatched"] if t not in SEEN]
    assert not missing, "fp8 templates the bench dispatches without a batch-64 element-wise test: %s" % missing
    unknown = [t for t in SEEN if t not in want["fp8_bench_dispatched"] and t not in want["fp8_other_known"]]
    assert not unknown, "fp8 templates dispatched at batch 64 that the checklist does not list: %s" % unknown
