"""GPU parity of the whole ResNet-50 backbone (HIP, bf16) against the CPU oracle on the same
inputs and weights, forward and backward, plus state_dict compatibility.

With seeded-random (untrained) weights and the tiny batches a CPU oracle can afford, ~60 stacked
small-batch BatchNorms amplify bf16 storage rounding chaotically (a single bf16 flip early on moves
sigmoid outputs by 0.5).  The end-to-end comparison therefore (1) puts the network in the
contractive regime a trained ResNet lives in by setting the residual-branch gain (bn3.weight) to
0.2 -- weights are inputs of the test, both sides get the same -- and (2) runs the oracle with its
bf16-storage emulation hook (``q=bf16_ste``: same fp32 math, tensors rounded to bf16 where the HIP
path stores them).  Every layer is additionally checked teacher-forced against plain fp32 math on
the HIP path's own inputs, with the untouched unit-gain random weights.
Tolerances: per-layer bf16-in/fp32-accumulate outputs <= 1e-2 rel; whole-net sigmoid
outputs mean |err| <= 1e-2 and max |err| <= 5e-2 (SURVEY 8d suggested 2e-2 max; measured here:
mean 5-7e-3, max 3-4e-2 over 3840 outputs, i.e. ~0.3 % rounding per layer random-walking through
~60 layers -- the two bf16 paths round at the same points but not to the same bits); loss <= 2 % rel.  Gradients cross ~60 bf16 layers and as many
discontinuous ReLU masks, so whole-net they are compared by direction and norm (cosine >= 0.90, norm
ratio within 10 %); the element-wise checks of dgrad / wgrad / BN-backward are in test_gpu_ops.py.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("S,N,hw", [(7, 8, 256), (14, 4, 192)])
def test_resnet50_forward_backward_vs_oracle(S, N, hw):
    from oracle import backbones as ob
    from oracle import loss as ol
    from oracle import train_step as ots
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.v1Loss import YOLOLossV1
    P = ob.init_params(ob.resnet50_param_shapes(S), "resnet", seed=11)
    for k in P:
        if k.endswith("bn3.weight"):
            P[k] = P[k] * 0.2
    net = resnet50(S=S)
    net.load_state_dict(P, strict=True)
    net = net.to(DEV).train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, 3, hw, hw, generator=g)
    grid = hw // (64 if S == 7 else 32)
    _, tg = ots.synthetic_batch(N, grid, hw=8)

    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    ref = ob.resnet50_forward(x, P, S, training=True, q=ob.bf16_ste)
    ref.retain_grad()
    ref_loss, _ = ol.yolo_loss(ref, tg, grid, 2, 20, batch_size=N)
    ref_loss.backward()

    pred = net(x.to(DEV))
    assert tuple(pred.shape) == (N, grid, grid, 30) and pred.dtype == torch.float32
    d = (pred.detach().cpu() - ref.detach()).abs()
    err = d.max().item()
    assert err <= 5e-2 and d.mean().item() <= 1e-2, "sigmoid outputs differ by max %g mean %g" % (err, d.mean().item())
    loss = YOLOLossV1(N, grid, 2, 20, _quiet=True)(pred, tg.to(DEV))
    np.testing.assert_allclose(loss.item(), ref_loss.item(), rtol=2e-2)
    # backward: feed both backbones the SAME upstream gradient (the oracle's d loss / d pred), so the
    # comparison is of the backbone backward only; the loss kernel's own gradient is pinned bit-tight
    # in test_gpu_loss_decode.py (the responsible-box argmax makes d loss / d pred discontinuous in pred)
    pred.backward(ref.grad.to(DEV))
    sd = dict(net.named_parameters())
    worst = 1.0
    for k, v in P.items():
        if not v.requires_grad:
            continue
        gg = sd[k].grad
        assert gg is not None and tuple(gg.shape) == tuple(v.shape), k
        c = _cos(gg.cpu(), v.grad)
        ratio = float(gg.norm().cpu() / (v.grad.norm() + 1e-30))
        worst = min(worst, c)
        # ReLU masks are discontinuous: the ~3 % forward deviation between two bf16 paths flips ~2 % of the
        # masks per layer, which alone costs cos ~0.98 per layer (measured: 0.97 at layer5 -> 0.93 at conv1,
        # norm ratio 1.00 +- 0.005).  The backward kernels themselves are pinned tightly, on identical
        # inputs, in test_gpu_ops.py.
        lim = 0.90 if v.dim() == 4 else 0.80     # BatchNorm gamma/beta gradients are 64..2048-element sums: noisier
        assert c >= lim and 0.9 <= ratio <= 1.1, "%s: cosine %.4f norm ratio %.3f" % (k, c, ratio)
    # running statistics and counters updated like nn.BatchNorm2d in train mode
    got = net.state_dict()
    np.testing.assert_allclose(got["bn1.running_mean"].cpu().numpy(), P["bn1.running_mean"].numpy(), rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(got["bn_end.running_var"].cpu().numpy(), P["bn_end.running_var"].numpy(), rtol=5e-2, atol=5e-3)
    assert int(got["layer3.2.bn2.num_batches_tracked"]) == 1


def test_resnet50_state_dict_keys_and_eval_mode():
    from oracle import backbones as ob
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    inv = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["inventory"]
    for S in (7, 14):
        net = resnet50(S=S)
        mine = [[k, list(v.shape)] for k, v in net.state_dict().items()]
        assert mine == inv["resnet_S%d" % S]           # same keys, order and OIHW shapes as the reference
    # eval mode uses running statistics and is deterministic / grad-free
    S = 14
    P = ob.init_params(ob.resnet50_param_shapes(S), "resnet", seed=3)
    for k in P:
        if k.endswith("running_var"):
            P[k] = P[k] * 1.5 + 0.1
        if k.endswith("running_mean"):
            P[k] = P[k] + 0.05
    net = resnet50(S=S)
    net.load_state_dict(P)
    net = net.to(DEV).eval()
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        a = net(x.to(DEV)).cpu()
        ref = ob.resnet50_forward(x, P, S, training=False, q=ob.bf16_ste)
    assert (a - ref).abs().max().item() <= 2e-2
    with pytest.raises(Exception):
        net(x)                                           # CPU tensor: no fallback


def test_train_loop_body_runs_and_loss_falls():
    # the loop body of train.py:155-172 with torch.optim.SGD(momentum=0.99) on the HIP backbone
    from oracle import train_step as ots
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.v1Loss import YOLOLossV1
    torch.manual_seed(0)
    net = resnet50(S=7).to(DEV).train()
    images, target = ots.synthetic_batch(8, 2, hw=128)
    images, target = images.to(DEV), target.to(DEV)
    opt = torch.optim.SGD(net.parameters(), lr=0.0, momentum=0.99)
    crit = YOLOLossV1(8, 2, 2, 20, _quiet=True)
    losses = []
    for it in range(12):
        for gparam in opt.param_groups:
            gparam["lr"] = 1e-3
        loss = crit(net(images), target)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0], losses


def test_training_trajectory_follows_the_oracle_over_several_sgd_steps():
    # train.py:155-172 repeated: the same weights, batch and learning-rate policy through the oracle (fp32 math, bf16
    # storage hook) and through the product step (HIP backbone + loss kernel + fused SGD, momentum 0.99).  The loss
    # sequence, the accumulated weight change and the BatchNorm running statistics must track each other.
    # LR 1e-4 is the reference's second-phase rate (train.py lr map {75: 1e-4}).  Measured: losses within 0.5 %,
    # weight-change cosine 0.91 (conv1) .. 0.99 (layer6), norm ratio 1.00 +- 0.006 after six steps; a single step
    # already sits at cosine 0.93 for conv1 (ReLU-mask flips between two bf16 paths, see the module docstring).  At
    # 1e-3 on these random weights the loss drops 25 -> 11 in six steps and the two bf16 trajectories, while keeping
    # losses within 3 % and norms within 1 %, decorrelate in direction (conv1 cosine 0.64) -- chaos, not a defect, so
    # the direction check is made where the problem is smooth.
    from oracle import backbones as ob
    from oracle import train_step as ots
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.train import learning_rate_policy, train_step
    from yolo_v1_amd.v1Loss import YOLOLossV1
    S, N, hw, steps = 7, 8, 256, 6
    grid = hw // 64
    P = ob.init_params(ob.resnet50_param_shapes(S), "resnet", seed=5)
    for k in P:
        if k.endswith("bn3.weight"):
            P[k] = P[k] * 0.2                      # contractive regime, see the module docstring
    P0 = {k: v.clone() for k, v in P.items()}
    net = resnet50(S=S)
    net.load_state_dict(P0, strict=True)
    net = net.to(DEV).train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, 3, hw, hw, generator=g)
    _, tg = ots.synthetic_batch(N, grid, hw=8)
    lr_map = {0: 1e-4}

    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    ref = ots.train_steps(P, x, tg, S, steps, "resnet", batch_size=N, lr_map=lr_map, fwd_kwargs={"q": ob.bf16_ste},
                          grid=grid)

    opt = FusedSGD(net.parameters(), lr=0.0, momentum=0.99)
    crit = YOLOLossV1(N, grid, 2, 20, _quiet=True)
    xd, td = x.to(DEV), tg.to(DEV)
    lr, got = 0.0, []
    for it in range(1, steps + 1):
        lr = learning_rate_policy(it, 0, lr, lr_map)
        got.append(float(train_step(net, crit, opt, xd, td, lr).item()))
    want = [r["loss"] for r in ref]
    assert all(abs(r["lr"] - 1e-4) < 1e-12 for r in ref)
    assert want[-1] < want[0]                       # the oracle's own trajectory descends ...
    # ... and the HIP path follows it step by step.  4 %: the third step sits on a sensitive point of this trajectory -- two
    # bit-different but equally valid roundings of the BatchNorm-backward formula gave 14.13 and 13.78 against the oracle's
    # 14.07 there, and both rejoin it afterwards (8.35 / 8.30 against 8.32 at step six)
    np.testing.assert_allclose(got, want, rtol=4e-2)
    sd = net.state_dict()
    for k in ("conv1.weight", "layer1.0.conv2.weight", "layer3.2.conv1.weight", "layer5.2.conv2.weight", "layer6.weight",
              "layer4.1.bn2.weight", "bn_end.bias"):
        d_ref = P[k].detach() - P0[k]
        d_got = sd[k].detach().cpu().float() - P0[k]
        c = _cos(d_got, d_ref)
        ratio = float(d_got.norm() / (d_ref.norm() + 1e-30))
        assert c >= (0.85 if d_ref.dim() == 4 else 0.8) and 0.9 <= ratio <= 1.1, "%s: cosine %.4f ratio %.3f" % (k, c, ratio)
    np.testing.assert_allclose(sd["bn1.running_mean"].cpu().numpy(), P["bn1.running_mean"].numpy(), rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(sd["layer2.1.bn1.running_var"].cpu().numpy(), P["layer2.1.bn1.running_var"].numpy(),
                               rtol=5e-2, atol=5e-3)
    assert int(sd["layer3.2.bn2.num_batches_tracked"]) == steps


def test_training_steps_are_bitwise_reproducible_run_to_run():
    # No atomics, fixed-order split-K and partial-row reductions: two independent runs from the same initial state must
    # agree bit for bit (loss sequence and every parameter).  Guards the raw-barrier DMA loops too: a missing LDS read
    # fence once made the convolution results differ from run to run.
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.train import train_step
    from yolo_v1_amd.v1Loss import YOLOLossV1
    from oracle import train_step as ots
    N, hw = 8, 256
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, 3, hw, hw, generator=g).to(DEV)
    _, tg = ots.synthetic_batch(N, hw // 64, hw=8)
    tg = tg.to(DEV)
    runs = []
    for _ in range(3):
        torch.manual_seed(4)
        net = resnet50(S=7).to(DEV).train()
        opt = FusedSGD(net.parameters(), lr=0.0, momentum=0.99)
        crit = YOLOLossV1(N, hw // 64, 2, 20, _quiet=True)
        losses = [train_step(net, crit, opt, x, tg, 1e-4).item() for _ in range(4)]
        torch.cuda.synchronize()
        runs.append((losses, {k: v.detach().clone() for k, v in net.state_dict().items()}))
    for losses, sd in runs[1:]:
        assert losses == runs[0][0], (losses, runs[0][0])
        for k, v in sd.items():
            assert torch.equal(v, runs[0][1][k]), k


def test_resnet50_every_layer_teacher_forced():
    """Each conv / BN+ReLU / block output of the HIP forward against fp32 torch math applied to the HIP
    path's own (bf16) input of that layer: isolates every kernel launch in the real network shapes."""
    import torch.nn.functional as F
    from oracle import backbones as ob
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    S, N, hw = 7, 4, 192
    P = ob.init_params(ob.resnet50_param_shapes(S), "resnet", seed=11)
    net = resnet50(S=S)
    net.load_state_dict(P)
    net = net.to(DEV).train()
    x = torch.randn(N, 3, hw, hw, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        pred, rec = net._run_forward(x.to(DEV), True, True)
    torch.cuda.synchronize()
    bfw = lambda k: P[k].to(torch.bfloat16).float()

    def nchw(a, C=None):
        t = a.t.float().cpu()
        return (t[..., :C] if C else t).permute(0, 3, 1, 2).contiguous()

    worst = [0.0, ""]

    def chk(name, got, want):
        rel = float((got - want).abs().max() / (want.abs().max() + 1e-9))
        if rel > worst[0]:
            worst[0], worst[1] = rel, name
        assert rel <= 1e-2, "%s: rel err %.3g" % (name, rel)

    bnf = lambda t, k: F.batch_norm(t, None, None, P[k + ".weight"], P[k + ".bias"], True)
    xp, y0, s0, _, H, W = rec["stem"][:6]
    chk("stem conv", nchw(y0), F.conv2d(x.to(torch.bfloat16).float(), bfw("conv1.weight"), stride=2, padding=3))
    # bn1 + relu + maxpool are one launch (the BatchNorm output is never stored)
    chk("stem bn+relu+maxpool", nchw(rec["blocks"][0][1]), F.max_pool2d(F.relu(bnf(nchw(y0), "bn1")), 3, 2, 1))
    names = ["%s.%d" % (st, i) for st in net._stage_names for i in range(len(getattr(net, st)))]
    for name, (blk, xin, y1, s1, z1, y2, s2, z2, y3, s3, yd, sd, out, _m) in zip(names, rec["blocks"]):
        xi = nchw(xin)
        chk(name + " y1", nchw(y1), F.conv2d(xi, bfw(name + ".conv1.weight")))
        chk(name + " z1", nchw(z1), F.relu(bnf(nchw(y1), name + ".bn1")))
        chk(name + " y2", nchw(y2), F.conv2d(nchw(z1), bfw(name + ".conv2.weight"), stride=blk.stride, padding=1))
        chk(name + " z2", nchw(z2), F.relu(bnf(nchw(y2), name + ".bn2")))
        chk(name + " y3", nchw(y3), F.conv2d(nchw(z2), bfw(name + ".conv3.weight")))
        idt = xi
        if yd is not None:
            chk(name + " yd", nchw(yd), F.conv2d(xi, bfw(name + ".downsample.0.weight"), stride=blk.stride))
            idt = bnf(nchw(yd), name + ".downsample.1")
        chk(name + " out", nchw(out), F.relu(bnf(nchw(y3), name + ".bn3") + idt))
    xh, yh, sh, pr = rec["head"]
    chk("head conv", nchw(yh, 30), F.conv2d(nchw(xh), bfw("layer6.weight")))
    ref = torch.sigmoid(bnf(nchw(yh, 30), "bn_end")).permute(0, 2, 3, 1)
    assert float((pr.cpu() - ref).abs().max()) <= 2e-3
    print("worst layer:", worst)


def test_fused_sgd_matches_torch_sgd_on_identical_gradients():
    from yolo_v1_amd.optim import FusedSGD
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 3, 7, 7), (256, 64, 1, 1), (30,), (128, 128, 3, 3), (7,), (2048,)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    pa[1].data = pa[1].data.contiguous(memory_format=torch.channels_last)
    pa[3].data = pa[3].data.contiguous(memory_format=torch.channels_last)
    pb = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in pa]
    oa = torch.optim.SGD(pa, lr=0.1, momentum=0.99)
    ob_ = FusedSGD(pb, lr=0.1, momentum=0.99)
    for step, lr in enumerate([0.1, 0.05, 0.2]):
        for grp in oa.param_groups + ob_.param_groups:
            grp['lr'] = lr
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).to(DEV)
            a.grad = gr.clone().contiguous(memory_format=torch.channels_last) if a.dim() == 4 else gr.clone()
            b.grad = gr.clone() if step == 1 else a.grad.clone()      # step 1: a gradient whose strides differ
        oa.step()
        ob_.step()
        for a, b in zip(pa, pb):
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_graphed_step_is_bitwise_the_eager_step():
    """A hipGraph replay of the training step launches the same kernels in the same order as the eager step:
    losses and weights must be bit-identical (everything in the step is deterministic: no float atomics)."""
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.train import GraphedStep, train_step
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    from yolo_v1_amd.v1Loss import YOLOLossV1
    images, target = synthetic_batch(4, 2, hw=128, device=DEV)
    torch.manual_seed(0)
    a = resnet50(S=7).to(DEV).train()
    b = resnet50(S=7).to(DEV).train()
    b.load_state_dict(a.state_dict())
    oa = FusedSGD(a.parameters(), lr=0.0, momentum=0.99)
    ob_ = FusedSGD(b.parameters(), lr=0.0, momentum=0.99)
    la, lb = YOLOLossV1(4, 2, 2, 20, _quiet=True), YOLOLossV1(4, 2, 2, 20, _quiet=True)
    lrs = [1e-3, 2e-3, 5e-4, 1e-3, 3e-3]
    ref = [train_step(a, la, oa, images, target, lr).item() for lr in lrs]
    for grp in ob_.param_groups:
        grp['lr'] = lrs[0]
    gs = GraphedStep(b, lb, ob_, images, target, warmup=1)      # its eager warm-up step consumes lrs[0]
    got = [gs(lr).item() for lr in lrs[1:]]
    assert got == ref[1:], (got, ref[1:])
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    for k in ("conv1.weight", "layer3.1.conv2.weight", "bn_end.bias"):
        assert torch.equal(pa[k], pb[k]), k
    from yolo_v1_amd.train import GraphedStep as _GS
    n_live = _GS.live_graphs()
    gs.close()                                                  # deterministic teardown: the exec is gone NOW
    assert _GS.live_graphs() == n_live - 1 and gs.graph is None and gs.net is None
    gs.close()                                                  # idempotent
    # an eager forward after graph replays sees the updated weights (bf16 shadow copies refreshed)
    a.eval(); b.eval()
    with torch.no_grad():
        assert torch.equal(a(images), b(images))


def test_data_parallel_graphed_step_two_graphs_is_bitwise_the_eager_step():
    """The multi-rank form of GraphedStep (two hipGraphs cut inside the backward pass, RCCL collective of the first
    phase's gradients issued between them, second collective + optimizer after) rehearsed with a process group of ONE
    rank: averaging over one rank is the identity, so losses and weights must be bit-identical to the eager step; the
    single-graph fallback (YV1_DP_PHASES=1 / executors without a phase boundary) likewise."""
    import torch.distributed as dist
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.distributed import GradSync
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.train import GraphedStep, train_step
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    from yolo_v1_amd.v1Loss import YOLOLossV1
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1,
                                device_id=torch.device(DEV))
    try:
        images, target = synthetic_batch(4, 2, hw=128, device=DEV)
        lrs = [1e-3, 2e-3, 5e-4, 1e-3]
        torch.manual_seed(0)
        a = resnet50(S=7).to(DEV).train()
        init = {k: v.clone() for k, v in a.state_dict().items()}
        oa = FusedSGD(a.parameters(), lr=0.0, momentum=0.99)
        la = YOLOLossV1(4, 2, 2, 20, _quiet=True)
        ref = [train_step(a, la, oa, images, target, lr).item() for lr in lrs]
        for two_phase in (True, False):
            b = resnet50(S=7).to(DEV).train()
            b.load_state_dict(init)
            ob_ = FusedSGD(b.parameters(), lr=0.0, momentum=0.99)
            for grp in ob_.param_groups:
                grp['lr'] = lrs[0]
            sync = GradSync(b)
            gs = GraphedStep(b, YOLOLossV1(4, 2, 2, 20, _quiet=True), ob_, images, target, sync, warmup=1,
                             two_phase=two_phase)
            assert gs.two_phase == two_phase and (gs.phase1 is not None) == two_phase
            if two_phase:      # head + layer5 + layer4: most of the gradient bytes
                early = sum(g.numel() for _, g in gs.phase1)
                assert 0.7 < early / sum(p.numel() for p in b.parameters()) < 0.9
            got = [gs(lr).item() for lr in lrs[1:]]
            assert got == ref[1:], (two_phase, got, ref[1:])
            pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
            for k in ("conv1.weight", "layer1.0.conv1.weight", "layer4.0.downsample.0.weight", "layer5.2.bn3.weight",
                      "bn_end.bias"):
                assert torch.equal(pa[k], pb[k]), (two_phase, k)
            assert len(gs.graphs) == (2 if two_phase else 1) and len(gs.phases) == len(gs.graphs) - 1
            # eager warm-up: one bucket; every replayed step: one collective per phase boundary + one for the rest
            assert sync.buckets_issued == 1 + len(gs.graphs) * (len(lrs) - 1)
            assert sync.in_place_buckets == sync.buckets_issued                            # all inside the gradient arena
            gs.close()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("fp8", [False, True])
def test_batched_eval_loop_matches_per_image_oracle_pipeline(fp8):
    """run_test_mAP (batched forward + batched GPU decoder/NMS + host voc_eval) against the reference's
    per-image pipeline restated with the oracle decoder and oracle voc_eval on the same network outputs -- for the bf16
    eval-mode network and for the fp8 inference executor (BASELINE config 5: fp8 conv + batched eval NMS)."""
    from collections import defaultdict
    from oracle import boxes as obx
    from oracle import voc as ov
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.utils.utils import VOC_CLASSES, run_test_mAP
    from yolo_v1_amd.utils.YOLODataLoader import yoloDataset
    torch.manual_seed(3)
    net = resnet50(S=7).to(DEV).eval()
    if fp8:
        from yolo_v1_amd.infer_fp8 import ResNetFp8
        net = ResNetFp8(net)
    ds = yoloDataset(None, train=False, with_file_path=True, S=7, length=10, image_size=128)
    target = ds.synthetic_ground_truth()

    class Q:
        def info(self, m):
            pass
    got = run_test_mAP(net, {k: [list(b) for b in v] for k, v in target.items()}, ds, len(ds), S=2, device=DEV,
                       logger=Q(), batch_size=4)
    # reference pipeline, one image at a time (utils/utils.py:393-411)
    preds = defaultdict(list)
    with torch.no_grad():
        for i in range(len(ds)):
            img, _, fname = ds[i]
            pred = net(img[None].to(DEV)).cpu().numpy()
            bx, cl, pr, _ = obx.decoder(pred, 2, 2, 0.005, 0.45)
            if len(pr) == 1 and pr[0] == 0:
                continue
            bx = np.clip(bx, 0.0, 1.0)
            for j in range(len(pr)):
                preds[VOC_CLASSES[int(cl[j])]].append([fname.split('.')[0], float(pr[j])] + [int(v * 128) for v in bx[j]])
    want, _ = ov.voc_eval(preds, target, VOC_CLASSES)
    assert abs(got - want) < 1e-12, (got, want)


def test_non_square_images_forward_backward_vs_oracle():
    """H != W (the reference only ever feeds 448x448, but nothing in its modules assumes a square): rows and columns
    must not be mixed up anywhere in the NHWC kernels (tap offsets, pooling windows, parity-decomposed dgrad, halo of
    the multi-tap wgrad).  Training-mode forward + backward of the S=14 network on 128x320 images."""
    from oracle import backbones as ob
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    S, N, H, W = 14, 4, 128, 320
    P = ob.init_params(ob.resnet50_param_shapes(S), "resnet", seed=5)
    for k in P:
        if k.endswith("bn3.weight"):
            P[k] = P[k] * 0.2
    net = resnet50(S=S)
    net.load_state_dict(P, strict=True)
    net = net.to(DEV).train()
    x = torch.randn(N, 3, H, W, generator=torch.Generator().manual_seed(2))
    for k, v in P.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    ref = ob.resnet50_forward(x, P, S, training=True, q=ob.bf16_ste)
    gup = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3)) * 1e-2
    ref.backward(gup)
    pred = net(x.to(DEV))
    assert tuple(pred.shape) == (N, H // 32, W // 32, 30)
    d = (pred.detach().cpu() - ref.detach()).abs()
    assert d.max().item() <= 5e-2 and d.mean().item() <= 1e-2, (d.max().item(), d.mean().item())
    pred.backward(gup.to(DEV))
    sd = dict(net.named_parameters())
    for k in ("conv1.weight", "layer1.0.conv2.weight", "layer2.0.conv2.weight", "layer2.0.downsample.0.weight",
              "layer3.2.conv2.weight", "layer4.1.conv1.weight", "layer6.weight", "bn1.weight"):
        a, b = sd[k].grad.detach().cpu(), P[k].grad
        assert _cos(a, b) >= 0.90, (k, _cos(a, b))
        assert 0.85 <= float(a.norm() / b.norm()) <= 1.15, (k, float(a.norm() / b.norm()))


def test_fused_eval_forward_matches_unfused_and_oracle():
    """eval(): one launch per convolution (BatchNorm folded into the epilogue) against the separate conv / BN-apply
    launches of the same network and against the fp32 oracle."""
    from oracle import backbones as ob
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    S = 7
    P = ob.init_params(ob.resnet50_param_shapes(S), "resnet", seed=4)
    for k in P:
        if k.endswith("running_var"):
            P[k] = P[k] * 1.3 + 0.2
        if k.endswith("running_mean"):
            P[k] = P[k] - 0.03
        if k.endswith("bn3.weight"):
            P[k] = P[k] * 0.3
    net = resnet50(S=S)
    net.load_state_dict(P)
    net = net.to(DEV).eval()
    x = torch.randn(3, 3, 128, 192, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        assert net.fused_eval
        a = net(x.to(DEV)).cpu()
        net.fused_eval = False
        b = net(x.to(DEV)).cpu()
        net.fused_eval = True
        ref = ob.resnet50_forward(x, P, S, training=False)
    assert (a - b).abs().max().item() <= 2e-2 and (a - b).abs().mean().item() <= 3e-3
    assert (a - ref).abs().max().item() <= 2e-2 and (a - ref).abs().mean().item() <= 3e-3
    assert (a - ref).abs().mean().item() <= (b - ref).abs().mean().item() * 1.2 + 1e-4     # one rounding fewer per layer


def test_full_size_properties_batch64_448():
    """BASELINE size (N=64, 448x448, S=7), where the CPU oracle cannot follow: size-independent properties.
    (1) eval-mode forward is per-image: the batch of 64 equals its two halves run separately, bit for bit (every conv
    reduces over K in the same order whatever the tile a pixel lands in); (2) determinism: two training-mode
    forward/backward passes of the same batch give bit-identical outputs and gradients; (3) linearity of the backward in
    the upstream gradient: doubling it doubles every parameter gradient exactly (powers of two commute with rounding)."""
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    torch.manual_seed(1)
    net = resnet50(S=7).to(DEV)
    x = torch.randn(64, 3, 448, 448, generator=torch.Generator().manual_seed(9)).to(DEV)
    net.eval()
    with torch.no_grad():
        full = net(x)
        halves = torch.cat([net(x[:32]), net(x[32:])])
    assert tuple(full.shape) == (64, 7, 7, 30) and torch.equal(full, halves)
    net.train()
    gup = torch.randn(64, 7, 7, 30, generator=torch.Generator().manual_seed(10)).to(DEV) * 1e-2
    rm0 = net.bn1.running_mean.clone()

    def run(scale):
        for p in net.parameters():
            p.grad = None
        with torch.no_grad():
            net.bn1.running_mean.copy_(rm0)
        pred = net(x)
        pred.backward(gup * scale)
        return pred.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()}
    p1, g1 = run(1.0)
    p2, g2 = run(1.0)
    p3, g3 = run(2.0)
    assert torch.equal(p1, p2) and torch.equal(p1, p3)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
        assert torch.isfinite(g1[n]).all(), n
    for n in ("conv1.weight", "layer1.0.conv1.weight", "layer3.4.conv2.weight", "layer5.2.bn3.weight", "layer6.weight", "bn_end.bias"):
        assert torch.equal(g3[n], 2.0 * g1[n]), n
