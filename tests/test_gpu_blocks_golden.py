"""The reference's OWN block tensors against the HIP executors.

tests/golden/block_cases.npz was written by oracle/gen_golden.py from the reference's imported classes:
`Bottleneck` with a stride-1 / stride-2 projection shortcut and with the identity shortcut
(backbones/OriginResNet.py:69-107), `_DenseLayer` and `_Transition` (backbones/OriginDenseNet.py:19-54) --
input, output, upstream gradient, input gradient, every parameter gradient and the BatchNorm running
statistics after the step, all fp32.  Here the same inputs / weights / upstream gradients go through the
production per-block executors (`ResNet.block_forward/block_backward`, `DenseNet.layer_forward/...`) in
their real sequencing: shortcut gradient folded into conv1's dgrad epilogue, 1-bit ReLU masks, parity-
decomposed stride-2 dgrad, side-stream weight gradients and their join, in-place concat buffers.

The kernels need channel counts that are multiples of 32; the fixtures' 16-wide `planes` are embedded in
32 channels with zero weights (BatchNorm gamma 1 / beta 0 on the padding): the extra channels carry exact
zeros forward and backward, the real channels see the same arithmetic.

Tolerance (stated per SURVEY 8d: bf16 storage, fp32 accumulate): the HIP path rounds inputs, weights and
every stored activation / gradient to bf16 (2^-9 relative each), the fixture is pure fp32.
  * relative L2 error  ||got - ref|| / ||ref||  <= 2e-2 for y, gx and every parameter gradient
    (cosine >= 0.9998; the whole-net tests could only ask for 0.90);
  * element-wise |got - ref| <= 2e-2 * max|ref| on >= 97 % of the elements and <= 0.25 * max|ref| on all:
    a pre-activation that rounds across zero flips one ReLU gate, which moves the handful of gradient
    elements behind it by their own magnitude -- a property of bf16 storage, not of the sequencing;
  * running statistics: rtol 1e-2, atol 1e-3.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(prefix):
    z = np.load(os.path.join(GOLDEN, "block_cases.npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        if k.startswith(prefix + "/"):
            out[k[len(prefix) + 1:]] = torch.from_numpy(z[k])
    return out


def _pad_to(t, shape, fill=0.0):
    out = torch.full(shape, fill, dtype=t.dtype)
    out[tuple(slice(0, n) for n in t.shape)] = t
    return out


def _act(x_nchw):
    from yolo_v1_amd import ops
    return ops.Act(x_nchw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV))


def _nchw(act, C=None):
    t = act.t.float().cpu()
    if C is not None:
        t = t[..., :C]
    return t.permute(0, 3, 1, 2).contiguous()


def _check(name, got, ref, report):
    got, ref = got.double().cpu(), ref.double()
    assert tuple(got.shape) == tuple(ref.shape), (name, got.shape, ref.shape)
    rel = float((got - ref).norm() / (ref.norm() + 1e-30))
    mx = float(ref.abs().max()) + 1e-30
    err = (got - ref).abs() / mx
    frac_ok = float((err <= 2e-2).double().mean())
    report.append("%-28s relL2 %.2e  within 2e-2*max: %.1f%%  worst %.3f*max" % (name, rel, 100 * frac_ok, float(err.max())))
    assert rel <= 2e-2, "%s: relative L2 error %g" % (name, rel)
    assert frac_ok >= 0.97 and float(err.max()) <= 0.25, "%s: %.1f%% within tolerance, worst %.3f*max" % (
        name, 100 * frac_ok, float(err.max()))


def _fwd_helpers(host):
    from yolo_v1_amd import ops
    norm = lambda stats, count, bn, C=None: ops.bn_finalize(stats, count, bn, C)
    conv = lambda xa, x8, cp, ya: ops.conv_fwd(xa, host.cw(cp), ya, True)
    return norm, conv, (lambda a: None)


@pytest.mark.parametrize("case,inpl,planes,stride,project", [
    ("bneck_s1_ds", 32, 16, 1, True), ("bneck_s2_ds", 64, 32, 2, True), ("bneck_plain", 64, 16, 1, False)])
def test_bottleneck_fixture_forward_backward(case, inpl, planes, stride, project):
    from yolo_v1_amd import ops
    from yolo_v1_amd.backbones.OriginResNet import Bottleneck, ResNet
    from yolo_v1_amd.engine import HipBackbone
    fx = _load(case)
    pp = max(32, planes)                                  # padded width of the two inner convolutions
    blk = Bottleneck(inpl, pp, stride, project=project)
    # the block's own expansion is 4*pp; the fixture's is 4*planes: rebuild conv3/bn3(/downsample) at the fixture width
    from yolo_v1_amd.engine import ConvParam, make_bn
    cout = planes * 4
    blk.conv3 = ConvParam(pp, cout, 1)
    blk.bn3 = make_bn(cout)
    if project:
        blk.downsample = torch.nn.Sequential(ConvParam(inpl, cout, 1, stride), make_bn(cout))
    sd = {}
    for k, v in fx.items():
        if not k.startswith("p/"):
            continue
        k = k[2:]
        tgt = tuple(blk.state_dict()[k].shape)
        if k.endswith("running_var") or (k.endswith(".weight") and v.dim() == 1):
            sd[k] = _pad_to(v, tgt, 1.0)
        else:
            sd[k] = _pad_to(v, tgt, 0.0) if v.dim() > 0 else v
    blk.load_state_dict(sd)

    class Host(HipBackbone):
        pass
    host = Host()
    host.blk = blk
    host = host.to(DEV).train()
    blk = host.blk
    norm, conv, q8 = _fwd_helpers(host)
    x = _act(fx["x"])
    out, _, brec = ResNet.block_forward(host, blk, x, None, norm, conv, q8, True)
    grads = {}
    side = ops.SideStream(torch.device(DEV), enabled=True)
    g = _act(fx["gy"])
    g_in = ResNet.block_backward(host, brec, g, grads, side)
    side.join()
    torch.cuda.synchronize()
    report = []
    _check("y", _nchw(out), fx["y"], report)
    _check("gx", _nchw(g_in), fx["gx"], report)
    named = dict(blk.named_parameters())
    for k, ref in fx.items():
        if not k.startswith("g/"):
            continue
        got = grads[named[k[2:]]].float().cpu()
        got = got[tuple(slice(0, n) for n in ref.shape)]
        _check(k, got, ref, report)
    bufs = dict(blk.named_buffers())
    for k, ref in fx.items():
        if k.startswith("after/"):
            got = bufs[k[6:]].float().cpu()[:ref.shape[0]]
            np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-2, atol=1e-3, err_msg=k)
    print("\n" + case + "\n" + "\n".join(report))


def _dense_host():
    from yolo_v1_amd.backbones.OriginDenseNet import DenseNet
    from yolo_v1_amd.engine import HipBackbone
    host = DenseNet.__new__(DenseNet)
    HipBackbone.__init__(host)
    host.growth = 32
    return host


def test_dense_layer_fixture_forward_backward():
    from yolo_v1_amd import ops
    from yolo_v1_amd.backbones.OriginDenseNet import DenseNet, _DenseLayer
    fx = _load("dense_layer")
    layer = _DenseLayer(64, 32, 4)
    layer.load_state_dict({k[2:]: v for k, v in fx.items() if k.startswith("p/")})
    host = _dense_host()
    host.layer = layer
    host = host.to(DEV).train()
    layer = host.layer
    norm, _, _ = _fwd_helpers(host)
    N, _, H, W = fx["x"].shape
    buf = ops.new_act(N, H, W, 96, DEV)
    buf.t.zero_()
    buf.t[..., :64] = fx["x"].permute(0, 2, 3, 1).to(torch.bfloat16).to(DEV)
    table = torch.empty((1, 2, 96), dtype=torch.float32, device=DEV)
    ops.stats_merge(ops.bn_stats(buf.window(0, 64)), table[0], 0)
    lrec = DenseNet.layer_forward(host, layer, buf, table, 64, norm, True)
    G = _act(fx["gy"])
    grads = {}
    side = ops.SideStream(torch.device(DEV), enabled=True)
    DenseNet.layer_backward(host, lrec, buf, G, grads, side)
    side.join()
    torch.cuda.synchronize()
    report = []
    _check("y (concat buffer)", _nchw(buf), fx["y"], report)
    _check("gx", _nchw(G, 64), fx["gx"], report)
    named = dict(layer.named_parameters())
    for k, ref in fx.items():
        if k.startswith("g/"):
            _check(k, grads[named[k[2:]]].float().cpu(), ref, report)
    bufs = dict(layer.named_buffers())
    for k, ref in fx.items():
        if k.startswith("after/"):
            np.testing.assert_allclose(bufs[k[6:]].float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=1e-3, err_msg=k)
    print("\ndense_layer\n" + "\n".join(report))


def test_transition_fixture_forward_backward():
    from yolo_v1_amd import ops
    from yolo_v1_amd.backbones.OriginDenseNet import DenseNet, _Transition
    fx = _load("transition")
    tr = _Transition(64, 32)
    tr.load_state_dict({k[2:]: v for k, v in fx.items() if k.startswith("p/")})
    host = _dense_host()
    host.tr = tr
    host = host.to(DEV).train()
    tr = host.tr
    norm, _, _ = _fwd_helpers(host)
    buf = _act(fx["x"])
    N, H, W = buf.N, buf.H, buf.W
    table = torch.empty((1, 2, 64), dtype=torch.float32, device=DEV)
    ops.stats_merge(ops.bn_stats(buf), table[0], 0)
    trec = DenseNet.transition_forward(host, tr, buf, table, norm)
    pooled = ops.new_act(N, H // 2, W // 2, 32, DEV)
    ops.avgpool_fwd(trec[5], pooled)                       # OriginDenseNet.py:54
    grads = {}
    side = ops.SideStream(torch.device(DEV), enabled=True)
    G = DenseNet.transition_backward(host, trec, _act(fx["gy"]), grads, side)
    side.join()
    torch.cuda.synchronize()
    report = []
    _check("y", _nchw(pooled), fx["y"], report)
    _check("gx", _nchw(G), fx["gx"], report)
    named = dict(tr.named_parameters())
    for k, ref in fx.items():
        if k.startswith("g/"):
            _check(k, grads[named[k[2:]]].float().cpu(), ref, report)
    bufs = dict(tr.named_buffers())
    for k, ref in fx.items():
        if k.startswith("after/"):
            np.testing.assert_allclose(bufs[k[6:]].float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=1e-3, err_msg=k)
    print("\ntransition\n" + "\n".join(report))
