"""Deferred BatchNorm backward (round 3; csrc/bn_deferred.hip + the `bn-deferred` epilogue of k_conv_dma)
against the sequence it replaces -- autograd of `conv1(relu1(norm1(cat(features))))` in a _DenseLayer (OriginDenseNet.py:
22-27,:32-36) and of `conv(relu(norm(x)))` in a _Transition (:50-52):

    reference path   dt = conv_dgrad(dy, w);  bn_backward(dt, x, st, mask_mode 2, accumulate) -> G, dgamma, dbeta
    deferred path    part = conv_dgrad_bn_deferred(dy, w, G, x, st[, pending])   (G += scale * mask * dgrad - pending
                                                                                   correction, in the epilogue)
                     dgamma, dbeta = bn_bwd_finalize_deferred(part, ...) (correction coefficients into K)
                     bn_deferred_fix(G, x, K)                            (G -= KA + KB * x where no later launch does it)

Both paths mask the SAME bf16-rounded data gradient and sum it in fp32, so dbeta / dgamma agree to summation order (1e-4 of
the largest); G differs by where its bf16 roundings fall (the reference rounds a*d - k2 - k3*xhat + old once, the deferred
path rounds old + a*d, then that - corr): two bf16 ulps of the magnitudes involved.  Then the whole DenseNet-121 training step
with and without it: same loss (the forward is untouched), head / norm5 gradients bit-identical, every other gradient within
rel-L2 8e-2 (measured worst 5.4e-2) / cosine 0.995, and bitwise reproducible run to run.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))


def _setup(N, H, cin, cout, seed):
    from yolo_v1_amd import ops
    g = torch.Generator().manual_seed(seed)
    x = ops.Act((torch.randn(N, H, H, cin, generator=g) * 1.3 + torch.randn(cin, generator=g) * 0.7).to(torch.bfloat16).to(DEV))
    bn = torch.nn.BatchNorm2d(cin).to(DEV)
    with torch.no_grad():
        bn.weight.copy_((torch.rand(cin, generator=g) + 0.5).to(DEV))
        bn.bias.copy_((torch.randn(cin, generator=g) * 0.3).to(DEV))
    st = ops.bn_finalize(ops.bn_stats(x), x.npix, bn)
    param = torch.nn.Parameter((torch.randn(cout, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).to(DEV)
                               .contiguous(memory_format=torch.channels_last))
    w = ops.ConvWeights(param, 1, 1, 0)
    w.refresh()
    dy = ops.Act((torch.randn(N, H, H, cout, generator=g) * 0.05).to(torch.bfloat16).to(DEV))
    old = (torch.randn(N, H, H, cin, generator=g) * 0.02).to(torch.bfloat16).to(DEV)
    return x, bn, st, w, dy, old


@pytest.mark.parametrize("N,H,cin,cout,accumulate", [
    (4, 28, 160, 128, True),       # cin = 64j + 32: 64-wide tiles whose last column tile reaches past cin (zero weight rows)
    (4, 28, 224, 128, True),       # cin % 128 == 96: 128- or 64-wide tiles, last one a quarter empty
    (4, 28, -160, 128, True),      # the same with UNPADDED transposed weights: the 128x32 fallback
    (2, 56, 64, 128, True),        # 64-wide tiles
    (8, 28, 256, 128, True),       # 128x128 tiles
    (3, 14, 320, 128, True),       # 64x64 tiles (few tiles)
    (2, 28, 256, 128, False),      # a transition: first writer of G
    # batch-64 bench shapes (DenseNet-121 at 448x448): one dense layer per block, and the three transitions
    (64, 112, 96, 128, True),      # 6272 partial rows: pre-reduced by yv1_reduce_rows
    (64, 56, 320, 128, True),
    (64, 28, 512, 128, True),
    (64, 14, 800, 128, True),
    (64, 112, 256, 128, False),
    (64, 56, 512, 256, False),
    (64, 28, 1024, 512, False),
])
def test_deferred_data_gradient_against_dgrad_plus_bn_backward(N, H, cin, cout, accumulate):
    from yolo_v1_amd import _lib, ops
    unpadded, cin = cin < 0, abs(cin)
    x, bn, st, w, dy, old = _setup(N, H, cin, cout, 7 * cin + H)
    if unpadded:
        w.tr = w.tr[:w.Ipad].clone()
    # reference
    G_ref = ops.Act(old.clone() if accumulate else torch.empty_like(old))
    dt = ops.new_act(N, H, H, cin, DEV)
    ops.conv_dgrad(dy, w, dt)
    dg_ref, db_ref = ops.bn_backward(dt, x, st, bn, G_ref, 2, accumulate=accumulate)
    # deferred
    G = ops.Act(old.clone() if accumulate else torch.empty_like(old))
    K = torch.zeros((2, cin), dtype=torch.float32, device=DEV)
    part = ops.conv_dgrad_bn_deferred(dy, w, G, x, st, accumulate=accumulate)
    cfg = ";".join(_lib.last_config())
    dg, db = ops.bn_bwd_finalize_deferred(part, x.npix, bn, st, K, accumulate=True)
    ops.bn_deferred_fix(G, x, K)
    torch.cuda.synchronize()
    print("\n%s  rows %d" % (cfg, part.shape[0]))
    assert "bn-deferred" in cfg and "k_conv_dma<" in cfg
    assert ("k_conv_dma<128,32," in cfg) == unpadded, cfg
    sb = float(db_ref.abs().max()) + 1e-12
    sg = float(dg_ref.abs().max()) + 1e-12
    assert float((db - db_ref).abs().max()) <= 1e-4 * sb, float((db - db_ref).abs().max()) / sb
    assert float((dg - dg_ref).abs().max()) <= 1e-3 * sg, float((dg - dg_ref).abs().max()) / sg
    a, b = G.t.float(), G_ref.t.float()
    # two extra bf16 roundings of values of this size: the masked term, the old gradient and the correction
    mag = (dt.t.float().abs() * st.scale.abs() + (old.float().abs() if accumulate else 0.0) + b.abs())
    tol = mag * 2.0 ** -7 + 1e-6
    assert bool(((a - b).abs() <= tol).all()), float(((a - b).abs() / tol).max())
    assert _rel(a, b) <= 5e-3, _rel(a, b)


def test_every_densenet121_layer_shape_at_batch_64():
    """Every norm1 -> conv1 of DenseNet-121 at 448x448, batch 64 (58 dense layers: cin = 64 + 32 i at 112^2, 128 + 32 i at
    56^2, 256 + 32 i at 28^2, 512 + 32 i at 14^2) and its four transitions through the deferred data gradient, against
    dgrad + the three BatchNorm-backward passes.  Device-side random operands (the point is the tile / guard logic of every
    channel count the step runs: 128-, 64- and 32-wide column tiles, partial last tiles, pre-reduced partial tables)."""
    from yolo_v1_amd import _lib, ops
    shapes = []
    for H, nf, nl in ((112, 64, 6), (56, 128, 12), (28, 256, 24), (14, 512, 16)):
        shapes += [(H, nf + 32 * i, 128, True) for i in range(nl)]
        shapes.append((H, nf + 32 * nl, (nf + 32 * nl) // 2, False))
    N = 64
    gen = torch.Generator(device=DEV).manual_seed(1234)
    seen = {}
    for (H, cin, cout, acc) in shapes:
        x = ops.Act((torch.randn(N, H, H, cin, generator=gen, device=DEV) * 1.3 + 0.4).to(torch.bfloat16))
        bn = torch.nn.BatchNorm2d(cin).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(cin, generator=gen, device=DEV) + 0.5)
            bn.bias.copy_(torch.randn(cin, generator=gen, device=DEV) * 0.3)
        st = ops.bn_finalize(ops.bn_stats(x), x.npix, bn)
        param = torch.nn.Parameter((torch.randn(cout, cin, 1, 1, generator=gen, device=DEV) * (2.0 / cin) ** 0.5)
                                   .contiguous(memory_format=torch.channels_last))
        w = ops.ConvWeights(param, 1, 1, 0)
        w.refresh()
        dy = ops.Act((torch.randn(N, H, H, cout, generator=gen, device=DEV) * 0.05).to(torch.bfloat16))
        old = (torch.randn(N, H, H, cin, generator=gen, device=DEV) * 0.02).to(torch.bfloat16)
        G_ref = ops.Act(old.clone() if acc else torch.empty_like(old))
        dt = ops.new_act(N, H, H, cin, DEV)
        ops.conv_dgrad(dy, w, dt)
        dg_ref, db_ref = ops.bn_backward(dt, x, st, bn, G_ref, 2, accumulate=acc)
        G = ops.Act(old.clone() if acc else torch.empty_like(old))
        K = torch.zeros((2, cin), dtype=torch.float32, device=DEV)
        part = ops.conv_dgrad_bn_deferred(dy, w, G, x, st, accumulate=acc)
        cfg = ";".join(_lib.last_config())
        dg, db = ops.bn_bwd_finalize_deferred(part, x.npix, bn, st, K, accumulate=True)
        ops.bn_deferred_fix(G, x, K)
        seen[cfg] = seen.get(cfg, 0) + 1
        sb, sg = float(db_ref.abs().max()) + 1e-12, float(dg_ref.abs().max()) + 1e-12
        assert "bn-deferred" in cfg, (H, cin, cfg)
        assert float((db - db_ref).abs().max()) <= 1e-4 * sb, (H, cin, float((db - db_ref).abs().max()) / sb)
        assert float((dg - dg_ref).abs().max()) <= 1e-3 * sg, (H, cin, float((dg - dg_ref).abs().max()) / sg)
        assert _rel(G.t, G_ref.t) <= 5e-3, (H, cin, _rel(G.t, G_ref.t))
        del x, dy, old, G, G_ref, dt
    print("\ntemplates over the %d shapes: %s" % (len(shapes), seen))


def test_nested_batchnorms_with_pending_corrections():
    """Three BatchNorms over nested channel ranges of ONE feature tensor, in the executor's order (the dense block's
    pattern, backbones/OriginDenseNet.py:layer_backward): each data gradient subtracts what the previous BatchNorm still owes
    the channels it covers (``pending``), the finalize replaces the table with this layer's coefficients, the 32-channel slice
    no later launch touches is fixed separately -- against three reference backward passes."""
    from yolo_v1_amd import ops
    N, H, ctot = 4, 28, 160
    g = torch.Generator().manual_seed(11)
    xt = (torch.randn(N, H, H, ctot, generator=g) * 1.1 + 0.3).to(torch.bfloat16).to(DEV)
    buf = ops.Act(xt)
    G_ref = ops.Act(torch.zeros_like(xt))
    G = ops.Act(torch.zeros_like(xt))
    K = torch.empty((2, ctot), dtype=torch.float32, device=DEV)
    owed, prev = False, None
    for li, cin in enumerate((160, 128, 96)):
        xin = buf.window(0, cin)
        bn = torch.nn.BatchNorm2d(cin).to(DEV)
        with torch.no_grad():
            bn.weight.copy_((torch.rand(cin, generator=g) + 0.5).to(DEV))
            bn.bias.copy_((torch.randn(cin, generator=g) * 0.3).to(DEV))
        st = ops.bn_finalize(ops.bn_stats(xin), xin.npix, bn)
        param = torch.nn.Parameter((torch.randn(128, cin, 1, 1, generator=g) * (2.0 / cin) ** 0.5).to(DEV)
                                   .contiguous(memory_format=torch.channels_last))
        w = ops.ConvWeights(param, 1, 1, 0)
        w.refresh()
        dy = ops.Act(((torch.randn(N, H, H, 128, generator=g) + 0.7) * 0.05).to(torch.bfloat16).to(DEV))   # not mean-free
        dt = ops.new_act(N, H, H, cin, DEV)
        ops.conv_dgrad(dy, w, dt)
        ops.bn_backward(dt, xin, st, bn, G_ref.window(0, cin), 2, accumulate=True)
        if owed:                                  # the slice [cin, prev) is complete: nothing below touches it again
            ops.bn_deferred_fix(G.window(cin, prev - cin), buf.window(cin, prev - cin), K[:, cin:prev])
        part = ops.conv_dgrad_bn_deferred(dy, w, G.window(0, cin), xin, st, accumulate=True,
                                          pending=K[:, :cin] if owed else None)
        ops.bn_bwd_finalize_deferred(part, xin.npix, bn, st, K[:, :cin], accumulate=False)
        owed, prev = True, cin
    ops.bn_deferred_fix(G.window(0, prev), buf.window(0, prev), K[:, :prev])
    torch.cuda.synchronize()
    assert _rel(G.t, G_ref.t) <= 8e-3, _rel(G.t, G_ref.t)


def test_densenet_training_step_with_and_without_the_deferred_backward():
    from yolo_v1_amd import ops
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    from yolo_v1_amd.v1Loss import YOLOLossV1
    images, target = synthetic_batch(8, 4, hw=256, device=DEV)
    runs = []
    default = ops.BN_DEFERRED
    try:
        for flag in (False, True, True):
            ops.BN_DEFERRED = flag
            torch.manual_seed(3)
            net = densenet121(S=7)
            gen = torch.Generator().manual_seed(5)
            with torch.no_grad():               # off the init point gamma = 1, beta = 0: there relu(gamma * xhat) = gamma * relu(xhat)
                for m in net.modules():         # and the NEXT BatchNorm removes the scale again -- dgamma is pure rounding noise
                    if isinstance(m, torch.nn.BatchNorm2d):
                        m.weight.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                        m.bias.copy_(torch.randn(m.num_features, generator=gen) * 0.3)
            net = net.to(DEV).train()
            crit = YOLOLossV1(8, 4, 2, 20, _quiet=True)
            loss = crit(net(images), target)
            loss.backward()
            torch.cuda.synchronize()
            runs.append((float(loss.item()), {n: q.grad.detach().clone() for n, q in net.named_parameters()}))
    finally:
        ops.BN_DEFERRED = default
    (l0, g0), (l1, g1), (l2, g2) = runs
    assert l0 == l1 == l2
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n
    # features.norm0 is followed by ReLU + max-pool only and then by the norm1 of every layer of block 1: the pooled features'
    # gradient g is orthogonal to span{1, f} (BatchNorm's invariances), so d/d(gamma0, beta0) is what is left after the large
    # terms cancel -- rounding noise on BOTH paths (tests/test_gpu_densenet.py skips it against the oracle for the same
    # reason: cosine 0.3-0.7 there).  Reported, not asserted.
    worst, bad = ("", 0.0), []
    for n in g0:
        if n.startswith(("layer6.", "bn_end.", "features.norm5.")):
            assert torch.equal(g1[n], g0[n]), n
            continue
        r = _rel(g1[n], g0[n])
        c = float(torch.nn.functional.cosine_similarity(g1[n].flatten().float(), g0[n].flatten().float(), dim=0))
        if n.startswith("features.norm0."):
            print("%s: rel-L2 %.3g cosine %.3g (rounding noise on both paths, not asserted)" % (n, r, c))
            continue
        if r > worst[1]:
            worst = (n, r)
        if not (r <= 8e-2 and c >= 0.995):
            bad.append((n, r, c))
    print("\nworst rel-L2 between the deferred and the pass-based backward: %s %.3g" % worst)
    assert not bad, bad[:12]
