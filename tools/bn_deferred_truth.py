"""Precision of the deferred BatchNorm backward against the pass-based one when the gradient is NOT mean-free: L BatchNorms
over nested channel ranges of one feature tensor (a dense block), each followed by a 1x1 convolution whose output gradient
has mean = `bias` x its standard deviation.  fp64 truth from the same bf16 operands; rel-L2 of the accumulated feature gradient.
    python tools/bn_deferred_truth.py [N H layers bias]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_v1_amd import ops

DEV = "cuda:0"
N, H, L = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 56, 6)
bias = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
c0, growth = 64, 32
ctot = c0 + growth * L
gen = torch.Generator(device=DEV).manual_seed(5)
xt = (torch.randn(N, H, H, ctot, generator=gen, device=DEV) * 1.1 + 0.3).to(torch.bfloat16)
buf = ops.Act(xt)
M = buf.npix
G_ref, G = ops.Act(torch.zeros_like(xt)), ops.Act(torch.zeros_like(xt))
K = torch.empty((2, ctot), dtype=torch.float32, device=DEV)
Gt = torch.zeros(M, ctot, dtype=torch.float64, device=DEV)
owed, prev = False, ctot
for li in reversed(range(L)):
    cin = c0 + growth * li + growth          # this "layer" normalises everything produced so far
    cin = min(cin, ctot)
    xin = buf.window(0, cin)
    bn = torch.nn.BatchNorm2d(cin).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(cin, generator=gen, device=DEV) + 0.5)
        bn.bias.copy_(torch.randn(cin, generator=gen, device=DEV) * 0.3)
    st = ops.bn_finalize(ops.bn_stats(xin), xin.npix, bn)
    param = torch.nn.Parameter((torch.randn(128, cin, 1, 1, generator=gen, device=DEV) * (2.0 / cin) ** 0.5)
                               .contiguous(memory_format=torch.channels_last))
    w = ops.ConvWeights(param, 1, 1, 0)
    w.refresh()
    dy = ops.Act(((torch.randn(N, H, H, 128, generator=gen, device=DEV) + bias) * 0.05).to(torch.bfloat16))
    dt = ops.new_act(N, H, H, cin, DEV)
    ops.conv_dgrad(dy, w, dt)
    ops.bn_backward(dt, xin, st, bn, G_ref.window(0, cin), 2, accumulate=True)
    if owed and prev > cin:                       # the executor's order (OriginDenseNet.layer_backward)
        ops.bn_deferred_fix(G.window(cin, prev - cin), buf.window(cin, prev - cin), K[:, cin:prev])
    part = ops.conv_dgrad_bn_deferred(dy, w, G.window(0, cin), xin, st, accumulate=True, pending=K[:, :cin] if owed else None)
    ops.bn_bwd_finalize_deferred(part, xin.npix, bn, st, K[:, :cin], accumulate=False)
    owed, prev = True, cin
    # truth
    X = xt.view(M, ctot)[:, :cin].double()
    Wd = w.fwd.view(128, cin).double()
    D = (dy.t.view(M, 128).double() @ Wd)
    mu, isd = X.mean(0), 1.0 / torch.sqrt(X.var(0, unbiased=False) + 1e-5)
    a = bn.weight.detach().double() * isd
    mask = (X * a + (bn.bias.detach().double() - mu * a)) > 0
    D = D * mask
    xh = (X - mu) * isd
    Gt[:, :cin] += a * (D - D.mean(0) - xh * (D * xh).mean(0))
ops.bn_deferred_fix(G.window(0, prev), buf.window(0, prev), K[:, :prev])
torch.cuda.synchronize()
rel = lambda t: float((t.view(M, ctot).double() - Gt).norm() / Gt.norm())
unc = float((G.t.view(M, ctot).double()).norm())
print("N=%d H=%d layers=%d gradient mean = %.2g x std: feature-gradient rel-L2 against fp64 truth: passes %.3e, deferred %.3e" % (
    N, H, L, bias, rel(G_ref.t), rel(G.t)))
