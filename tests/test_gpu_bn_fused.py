"""BatchNorm finalize fused into the consuming launch (round 3; csrc/elementwise.hip "finalize fused into the consuming
launch"): `yv1_bn_finalize_apply`, `yv1_bn_bwd_finalize_apply`, `yv1_bn_bwd_finalize_apply_dual` against the launch pairs
they replace.  The fused kernels' producer workgroups sum the partial rows in the SAME order as the stand-alone finalize
kernels (64 row lanes, two accumulators per lane, fixed combine order), so the bar is BIT-exact: coefficients, running
statistics, outputs, masks, parameter gradients -- and a whole training step is bitwise the same with and without the
fusion (train-mode nn.BatchNorm2d of OriginResNet.py:90-105 and its autograd backward either way).  The fault word of the
in-kernel wait (its exit condition) must stay clear.

The fusion is OFF by default (ops.BN_FUSED, YV1_BN_FUSED=1 turns it on): measured on MI355X it is slower than the launch
pairs on every shape (tools/bench_bn_fused.py, profiles/r03_bn_fused_table.txt: +14 ... +50 us per BatchNorm, the step
-13 %) -- a hand-off through memory between workgroups of different XCDs costs 13-18 us, a dependent launch inside a
captured graph 1-2 us.  These tests keep the experiment honest: it computes exactly what the default path computes.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
import os
DEFAULT = os.environ.get("YV1_BN_FUSED", "0") == "1"          # ops.BN_FUSED's default: off (measured slower, see ops.py)


def _bn(C, g):
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    return bn.to(DEV)


def _partials(y, rows, g):
    """Partial (sum, sumsq) rows [rows][2][C] whose column sums are the true batch sums of y (split at random)."""
    yf = y.t.float().reshape(-1, y.C)
    s, q = yf.sum(0), (yf * yf).sum(0)
    w = torch.rand(rows, 1, generator=g).to(DEV) + 0.1
    w = w / w.sum()
    return torch.stack([w * s, w * q], 1).contiguous()


CASES = [  # N, H, W, C, rows, residual kind (None | "plain" | "bn"), relu, mask
    (4, 28, 28, 64, 37, None, True, False),
    (4, 28, 28, 256, 784, "plain", True, True),
    (2, 14, 14, 1024, 2048, "bn", True, True),
    (2, 7, 7, 2048, 5000, "bn", True, True),          # > 2048 rows: pre-reduced by yv1_reduce_rows in both forms
    (3, 9, 11, 72, 5, None, False, False),             # C/8 = 9: the non-FIXED_C path
    (1, 4, 4, 8, 1, "plain", True, True),
]


@pytest.mark.parametrize("case", CASES)
def test_fused_forward_is_bitwise_the_two_launch_form(case):
    from yolo_v1_amd import ops
    N, H, W, C, rows, rkind, relu, want_mask = case
    g = torch.Generator().manual_seed(hash(case) % 9973)
    y = ops.Act((torch.randn(N, H, W, C, generator=g) * 2 + 0.3).to(torch.bfloat16).to(DEV))
    res = ops.Act(torch.randn(N, H, W, C, generator=g).to(torch.bfloat16).to(DEV)) if rkind else None
    part = _partials(y, rows, g)
    rpart = _partials(res, max(1, rows // 2), g) if rkind == "bn" else None
    outs = []
    for fused in (False, True):
        bn, rbn = _bn(C, torch.Generator().manual_seed(1)), _bn(C, torch.Generator().manual_seed(2))
        z = ops.new_act(N, H, W, C, DEV)
        z.t.fill_(-7.0)
        if fused:
            st, mask, rst = ops.bn_finalize_apply(part, y.npix, bn, y, z, relu=relu, residual=res, res_stats=rpart,
                                                  res_bn=rbn if rkind == "bn" else None, want_mask=want_mask)
        else:
            st = ops.bn_finalize(part, y.npix, bn)
            rst = ops.bn_finalize(rpart, y.npix, rbn) if rkind == "bn" else None
            mask = ops.bn_apply(y, st, z, relu=relu, residual=res, res_state=rst, want_mask=want_mask)
        torch.cuda.synchronize()
        outs.append((z.t.clone(), st.buf.clone(), rst.buf.clone() if rst is not None else None,
                     mask.t.clone() if mask is not None else None, bn.running_mean.clone(), bn.running_var.clone(),
                     rbn.running_mean.clone(), rbn.running_var.clone()))
    for a, b in zip(*outs):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    # and against nn.BatchNorm2d semantics on the true batch statistics (fp32 reference, bf16 output rounding)
    yf = y.t.float()
    mean, var = yf.mean((0, 1, 2)), yf.var((0, 1, 2), unbiased=False)
    bn = _bn(C, torch.Generator().manual_seed(1))
    ref = (yf - mean) * torch.rsqrt(var + 1e-5) * bn.weight + bn.bias
    if rkind == "plain":
        ref = ref + res.t.float()
    if rkind != "bn":
        if relu:
            ref = ref.clamp_min(0)
        assert float((outs[1][0].float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max()) + 1e-3
    assert not ops.fused_sync_fault()


BWD_CASES = [  # N, H, W, C, mask_mode, with dres, accumulate
    (4, 28, 28, 64, 2, False, False),
    (4, 28, 28, 256, 3, False, False),
    (2, 14, 14, 1024, 1, True, False),
    (3, 9, 11, 72, 0, False, True),
    (8, 56, 56, 128, 2, False, False),
]


@pytest.mark.parametrize("case", BWD_CASES)
def test_fused_backward_is_bitwise_the_two_launch_form(case):
    from yolo_v1_amd import ops
    N, H, W, C, mode, with_dres, acc = case
    g = torch.Generator().manual_seed(hash(case) % 9973)
    y = ops.Act((torch.randn(N, H, W, C, generator=g) * 2 + 0.3).to(torch.bfloat16).to(DEV))
    dz = ops.Act(torch.randn(N, H, W, C, generator=g).to(torch.bfloat16).to(DEV))
    bn = _bn(C, torch.Generator().manual_seed(1))
    st = ops.bn_finalize(ops.bn_stats(y), y.npix, bn)
    zact = ops.new_act(N, H, W, C, DEV)
    mask = ops.bn_apply(y, st, zact, relu=True, want_mask=True)
    zarg = {0: None, 1: zact, 2: None, 3: mask}[mode]
    old = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16).to(DEV)
    outs = []
    for fused in (False, True):
        ops.BN_FUSED = fused
        try:
            dy = ops.Act(old.clone())
            dres = ops.new_act(N, H, W, C, DEV) if with_dres else None
            dg, db = ops.bn_backward(dz, y, st, bn, dy, mode, z=zarg, dres=dres, accumulate=acc)
            torch.cuda.synchronize()
            outs.append((dy.t.clone(), dg.clone(), db.clone(), dres.t.clone() if dres is not None else None))
        finally:
            ops.BN_FUSED = DEFAULT
    for a, b in zip(*outs):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    assert not ops.fused_sync_fault()


def test_fused_dual_backward_is_bitwise_the_two_launch_form():
    from yolo_v1_amd import ops
    N, H, W, C = 4, 28, 28, 512
    g = torch.Generator().manual_seed(5)
    ya = ops.Act((torch.randn(N, H, W, C, generator=g) * 2).to(torch.bfloat16).to(DEV))
    yb = ops.Act((torch.randn(N, H, W, C, generator=g) + 0.5).to(torch.bfloat16).to(DEV))
    dz = ops.Act(torch.randn(N, H, W, C, generator=g).to(torch.bfloat16).to(DEV))
    bna, bnb = _bn(C, torch.Generator().manual_seed(1)), _bn(C, torch.Generator().manual_seed(2))
    sta, stb = ops.bn_finalize(ops.bn_stats(ya), ya.npix, bna), ops.bn_finalize(ops.bn_stats(yb), yb.npix, bnb)
    out = ops.new_act(N, H, W, C, DEV)
    mask = ops.bn_apply(ya, sta, out, relu=True, residual=yb, res_state=stb, want_mask=True)
    outs = []
    for fused in (False, True):
        ops.BN_FUSED = fused
        try:
            dya, dyb = ops.new_act(N, H, W, C, DEV), ops.new_act(N, H, W, C, DEV)
            (ga, ba), (gb, bb) = ops.bn_backward_dual(dz, mask, (ya, sta, bna, dya), (yb, stb, bnb, dyb))
            torch.cuda.synchronize()
            outs.append((dya.t.clone(), dyb.t.clone(), ga.clone(), ba.clone(), gb.clone(), bb.clone()))
        finally:
            ops.BN_FUSED = DEFAULT
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert not ops.fused_sync_fault()


def test_training_steps_are_bitwise_the_same_with_and_without_the_fusion():
    from yolo_v1_amd import ops
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.train import GraphedStep, train_step
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    from yolo_v1_amd.v1Loss import YOLOLossV1
    images, target = synthetic_batch(4, 2, hw=128, device=DEV)
    runs = []
    for fused in (False, True, "graph"):
        ops.BN_FUSED = bool(fused)
        try:
            torch.manual_seed(3)
            net = resnet50(S=7).to(DEV).train()
            opt = FusedSGD(net.parameters(), lr=0.0, momentum=0.99)
            crit = YOLOLossV1(4, 2, 2, 20, _quiet=True)
            if fused == "graph":                            # the fused launches inside a captured, replayed step
                with GraphedStep(net, crit, opt, images, target, warmup=1, preserve_state=True) as gs:
                    losses = [float(gs(1e-3).item()) for _ in range(3)]
            else:
                losses = [float(train_step(net, crit, opt, images, target, 1e-3).item()) for _ in range(3)]
            torch.cuda.synchronize()
            runs.append((losses, {k: v.detach().clone() for k, v in net.state_dict().items()}))
        finally:
            ops.BN_FUSED = DEFAULT
    for losses, sd in runs[1:]:
        assert losses == runs[0][0], (losses, runs[0][0])
        for k, v in sd.items():
            assert torch.equal(v, runs[0][1][k]), k
    assert not ops.fused_sync_fault()
