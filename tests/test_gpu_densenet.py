"""GPU parity of the DenseNet-121 backbone (HIP, bf16, concat-free block buffers) against the CPU
oracle (fp32 math with bf16-storage emulation), forward, backward and eval mode.  Tolerances as in
test_gpu_resnet.py; the kernels themselves are pinned element-wise in test_gpu_ops.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("S,N,hw", [(7, 4, 192), (14, 4, 128)])
def test_densenet121_forward_backward_vs_oracle(S, N, hw):
    from oracle import backbones as ob
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121
    P = ob.init_params(ob.densenet121_param_shapes(S), "densenet", seed=5)
    net = densenet121(S=S)
    net.load_state_dict(P, strict=True)
    net = net.to(DEV).train()
    x = torch.randn(N, 3, hw, hw, generator=torch.Generator().manual_seed(2))
    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    ref = ob.densenet121_forward(x, P, S, training=True, q=ob.bf16_ste)
    gup = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3)) * 0.1
    ref.backward(gup)
    pred = net(x.to(DEV))
    grid = hw // (64 if S == 7 else 32)
    assert tuple(pred.shape) == (N, grid, grid, 30)
    d = (pred.detach().cpu() - ref.detach()).abs()
    # 121 bf16 layers, no residual damping, BatchNorm over <= 36 samples in the last block: measured mean 2.3e-2,
    # max 1.2e-1 with every layer within 4e-3 rel of fp32 math when teacher-forced (test below)
    assert d.max().item() <= 2e-1 and d.mean().item() <= 4e-2, "max %g mean %g" % (d.max().item(), d.mean().item())
    pred.backward(gup.to(DEV))
    sd = dict(net.named_parameters())
    bad = []
    for k, v in P.items():
        if not v.requires_grad:
            continue
        gg = sd[k].grad
        assert gg is not None and tuple(gg.shape) == tuple(v.shape), k
        c = _cos(gg.cpu(), v.grad)
        ratio = float(gg.norm().cpu() / (v.grad.norm() + 1e-30))
        # 121 bf16 layers of discontinuous ReLU / max-pool routing between two noisy forwards: a sanity bound only
        # (measured: 0.70 at conv0 .. 0.98 at the head).  features.norm0 is skipped: every consumer of the pooled
        # stem output is a per-channel BatchNorm (norm1 of each layer, the transition norm), so the loss is invariant
        # to a per-channel rescaling of it and the exact d/d(gamma0,beta0) nearly cancels to zero -- what is left is
        # rounding noise (cosine 0.3-0.7, norm ratio ~2).  The same kernel chain is pinned on identical inputs in
        # test_gpu_ops.py::test_stem_chain_backward.
        if k.startswith("features.norm0"):
            continue
        elif v.dim() != 4:                      # BatchNorm gamma/beta: short, strongly cancelling sums -> direction only
            if c < 0.5:
                bad.append((k, round(c, 4), round(ratio, 3)))
        elif not (c >= 0.50 and 0.8 <= ratio <= 1.25):
            bad.append((k, round(c, 4), round(ratio, 3)))
    assert not bad, bad[:10]
    got = net.state_dict()
    np.testing.assert_allclose(got["features.denseblock2.denselayer3.norm1.running_mean"].cpu().numpy(),
                               P["features.denseblock2.denselayer3.norm1.running_mean"].numpy(), rtol=3e-2, atol=3e-3)
    assert int(got["features.norm5.num_batches_tracked"]) == 1


def test_densenet121_eval_mode_and_train_steps():
    from oracle import backbones as ob
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    from yolo_v1_amd.v1Loss import YOLOLossV1
    S = 14
    P = ob.init_params(ob.densenet121_param_shapes(S), "densenet", seed=6)
    for k in P:
        if k.endswith("running_var"):
            P[k] = P[k] * 1.3 + 0.1
    net = densenet121(S=S)
    net.load_state_dict(P)
    net = net.to(DEV).eval()
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        a = net(x.to(DEV)).cpu()
        ref = ob.densenet121_forward(x, P, S, training=False, q=ob.bf16_ste)
    assert (a - ref).abs().max().item() <= 3e-2
    with torch.no_grad():                                # norm2 + ReLU in the conv epilogue vs the separate BN-apply pass
        net.fused_eval = False
        b = net(x.to(DEV)).cpu()
        net.fused_eval = True
    assert (a - b).abs().max().item() <= 2e-2 and (b - ref).abs().max().item() <= 3e-2
    net.train()
    images, target = synthetic_batch(8, 4, hw=128, device=DEV)
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
    crit = YOLOLossV1(8, 4, 2, 20, _quiet=True)
    losses = []
    for _ in range(10):
        loss = crit(net(images), target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_densenet121_every_layer_teacher_forced():
    """Every tensor the HIP forward produces against fp32 torch math applied to the HIP path's own (bf16)
    input of that layer -- includes the in-place channel-slice writes that replace torch.cat and the
    statistics table that replaces per-BatchNorm statistic passes."""
    import torch.nn.functional as F
    from oracle import backbones as ob
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121
    S, N, hw = 7, 4, 192
    P = ob.init_params(ob.densenet121_param_shapes(S), "densenet", seed=5)
    net = densenet121(S=S)
    net.load_state_dict(P)
    net = net.to(DEV).train()
    x = torch.randn(N, 3, hw, hw, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        pred, rec = net._run_forward(x.to(DEV), True, True)
    torch.cuda.synchronize()
    bfw = lambda k: P[k].to(torch.bfloat16).float()

    def nchw(a, c0=0, C=None):
        t = a.t.float().cpu()
        C = C if C is not None else a.C
        return t[..., a.c0 + c0: a.c0 + c0 + C].permute(0, 3, 1, 2).contiguous()

    def chk(name, got, want):
        rel = float((got - want).abs().max() / (want.abs().max() + 1e-9))
        assert rel <= 1e-2, "%s: rel err %.3g" % (name, rel)

    bnf = lambda t, k: F.batch_norm(t, None, None, P[k + ".weight"], P[k + ".bias"], True)
    xp, y0, s0, _, H, W = rec["stem"][:6]
    chk("stem conv", nchw(y0), F.conv2d(x.to(torch.bfloat16).float(), bfw("features.conv0.weight"), stride=2, padding=3))
    z0 = F.relu(bnf(nchw(y0), "features.norm0"))       # norm0 + relu0 + pool0 are one launch: z0 is never stored
    bi, prev_yc = 0, None
    for st in rec["stages"]:
        if st[0] == "block":
            bi += 1
            _, buf, lrecs, nf = st
            first = nchw(buf, 0, nf)
            chk("pool into block %d" % bi, first, F.max_pool2d(z0, 3, 2, 1) if bi == 1 else F.avg_pool2d(prev_yc, 2, 2))
            for li, (layer, cin, st1, t1, y1, st2, t2) in enumerate(lrecs):
                p = "features.denseblock%d.denselayer%d" % (bi, li + 1)
                chk(p + " t1", nchw(t1), F.relu(bnf(nchw(buf, 0, cin), p + ".norm1")))
                chk(p + " y1", nchw(y1), F.conv2d(nchw(t1), bfw(p + ".conv1.weight")))
                chk(p + " t2", nchw(t2), F.relu(bnf(nchw(y1), p + ".norm2")))
                chk(p + " slice", nchw(buf, cin, 32), F.conv2d(nchw(t2), bfw(p + ".conv2.weight"), padding=1))
        else:
            _, tr, buf, stt, t, yc = st
            p = "features.transition%d" % bi
            chk(p + " t", nchw(t), F.relu(bnf(nchw(buf), p + ".norm")))
            chk(p + " yc", nchw(yc), F.conv2d(nchw(t), bfw(p + ".conv.weight")))
            prev_yc = nchw(yc)
    buf, st5, t5, yh, sh, pr = rec["head"]
    chk("norm5", nchw(t5), F.relu(bnf(nchw(buf), "features.norm5")))
    chk("head conv", nchw(yh, 0, 30), F.conv2d(nchw(t5), bfw("layer6.weight")))
    ref = torch.sigmoid(bnf(nchw(yh, 0, 30), "bn_end")).permute(0, 2, 3, 1)
    assert float((pr.cpu() - ref).abs().max()) <= 2e-3


def test_mini_densenet_backward_tight():
    """A 2-block DenseNet (same executor, 4 dense layers + 1 transition) is shallow enough that bf16 noise
    does not swamp the comparison: gradients of every parameter against fp32 autograd, cosine >= 0.99."""
    import torch.nn.functional as F
    from oracle import backbones as ob
    from yolo_v1_amd.backbones.OriginDenseNet import DenseNet
    torch.manual_seed(0)
    net = DenseNet(block_config=(2, 2), S=14)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    P = {k: v.detach().clone().contiguous() for k, v in net.state_dict().items()}
    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    net = net.to(DEV).train()
    N, hw = 6, 64
    x = torch.randn(N, 3, hw, hw, generator=torch.Generator().manual_seed(9))
    q = ob.bf16_ste
    h = q(F.conv2d(q(x), q(P["features.conv0.weight"]), stride=2, padding=3))
    h = q(F.relu(ob._bn(h, P, "features.norm0", True)))
    h = F.max_pool2d(h, 3, 2, 1)
    for li in (1, 2):
        h = ob.dense_layer(h, P, "features.denseblock1.denselayer%d" % li, True, q)
    h = ob.transition(h, P, "features.transition1", True, q)
    for li in (1, 2):
        h = ob.dense_layer(h, P, "features.denseblock2.denselayer%d" % li, True, q)
    h = q(F.relu(ob._bn(h, P, "features.norm5", True)))
    h = q(F.conv2d(h, q(P["layer6.weight"])))
    ref = torch.sigmoid(ob._bn(h, P, "bn_end", True)).permute(0, 2, 3, 1)
    gup = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3))
    ref.backward(gup)
    pred = net(x.to(DEV))
    assert (pred.detach().cpu() - ref.detach()).abs().max().item() <= 2e-2
    pred.backward(gup.to(DEV))
    bad = []
    for k, p in net.named_parameters():
        c = _cos(p.grad.cpu(), P[k].grad)
        ratio = float(p.grad.norm().cpu() / (P[k].grad.norm() + 1e-30))
        if k.startswith("features.norm0"):        # near-zero by symmetry, see the note in the full-size test
            continue
        else:
            ok = c >= 0.97 and 0.92 <= ratio <= 1.08
        if not ok:
            bad.append((k, round(c, 4), round(ratio, 3)))
    assert not bad, bad


def test_densenet_graphed_steps_single_and_data_parallel_are_bitwise_the_eager_step():
    """hipGraph replay of the DenseNet training step (one rank: whole step in one graph; data-parallel form rehearsed with a
    1-rank RCCL group: forward+backward graph, one collective, fused SGD) against the eager step, bit for bit."""
    import torch.distributed as dist
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121
    from yolo_v1_amd.distributed import GradSync
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.train import GraphedStep, train_step
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    from yolo_v1_amd.v1Loss import YOLOLossV1
    images, target = synthetic_batch(4, 4, hw=128, device=DEV)
    lrs = [1e-3, 2e-3, 5e-4]
    torch.manual_seed(0)
    a = densenet121(S=14).to(DEV).train()
    init = {k: v.clone() for k, v in a.state_dict().items()}
    oa = FusedSGD(a.parameters(), lr=0.0, momentum=0.99)
    ref = [train_step(a, YOLOLossV1(4, 4, 2, 20, _quiet=True), oa, images, target, lr).item() for lr in lrs]
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29519", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        for with_sync in (False, True):
            b = densenet121(S=14).to(DEV).train()
            b.load_state_dict(init)
            ob_ = FusedSGD(b.parameters(), lr=0.0, momentum=0.99)
            for grp in ob_.param_groups:
                grp['lr'] = lrs[0]
            gs = GraphedStep(b, YOLOLossV1(4, 4, 2, 20, _quiet=True), ob_, images, target, GradSync(b) if with_sync else None,
                             warmup=1)
            assert not gs.two_phase                      # no phase boundary in this executor: one graph, one collective
            got = [gs(lr).item() for lr in lrs[1:]]
            assert got == ref[1:], (with_sync, got, ref[1:])
            pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
            for k in ("features.conv0.weight", "features.denseblock2.denselayer5.conv2.weight", "features.norm5.weight", "bn_end.bias"):
                assert torch.equal(pa[k], pb[k]), (with_sync, k)
            gs.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_restructured_backward_is_as_close_to_the_oracle_as_the_passes():
    """The deferred norm1 / transition-norm backward and the fused norm2 reduction round at different places than the
    dgrad + reduce + apply sequence.  Both against the oracle's fp32 backward of the same bf16-storage forward, every
    parameter gradient: the restructured path must not sit further away (median and upper-quartile relative error within
    15 % of the pass-based path's; measured: see the printed line)."""
    from oracle import backbones as ob
    from yolo_v1_amd import ops
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121
    S, N, hw = 7, 8, 256
    P = ob.init_params(ob.densenet121_param_shapes(S), "densenet", seed=9)
    x = torch.randn(N, 3, hw, hw, generator=torch.Generator().manual_seed(4))
    gup = None
    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    ref = ob.densenet121_forward(x, {k: (v if "running" not in k else v.clone()) for k, v in P.items()}, S, training=True,
                                 q=ob.bf16_ste)
    gup = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3)) * 0.1 + 0.05      # not mean-free on purpose
    ref.backward(gup)
    errs = {}
    saved = (ops.BN_DEFERRED, ops.BN_SUMS_IN_DGRAD)
    try:
        for name, flags in (("passes", (False, False)), ("restructured", (True, True))):
            ops.BN_DEFERRED, ops.BN_SUMS_IN_DGRAD = flags
            net = densenet121(S=S)
            net.load_state_dict({k: v.detach().clone() for k, v in P.items()}, strict=True)
            net = net.to(DEV).train()
            net(x.to(DEV)).backward(gup.to(DEV))
            sd = dict(net.named_parameters())
            e = {}
            for k, v in P.items():
                if not v.requires_grad or k.startswith("features.norm0"):       # norm0: rounding noise on every path
                    continue
                g = sd[k].grad.detach().cpu().double()
                e[k] = float((g - v.grad.double()).norm() / (v.grad.double().norm() + 1e-30))
            errs[name] = e
    finally:
        ops.BN_DEFERRED, ops.BN_SUMS_IN_DGRAD = saved
    a = np.array([errs["passes"][k] for k in errs["passes"]])
    b = np.array([errs["restructured"][k] for k in errs["passes"]])
    print("\nrelative error of %d parameter gradients against the oracle: passes median %.3g q75 %.3g max %.3g | restructured "
          "median %.3g q75 %.3g max %.3g | restructured worse by > 1.25x on %d, better by > 1.25x on %d" % (
              len(a), np.median(a), np.quantile(a, 0.75), a.max(), np.median(b), np.quantile(b, 0.75), b.max(),
              int((b > 1.25 * a).sum()), int((a > 1.25 * b).sum())))
    assert np.median(b) <= 1.15 * np.median(a) and np.quantile(b, 0.75) <= 1.15 * np.quantile(a, 0.75)
