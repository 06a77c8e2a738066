"""Which side of tests/test_gpu_bn3_algebra.py::test_bn3_algebra_matches_the_pass_based_backward is closer to exact arithmetic:
the pass-based BatchNorm-3 backward + conv3 wgrad, or the algebra?  fp64 truth from the same bf16 operands on the GPU.
    python tools/bn3_algebra_truth.py [N H p]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_v1_amd import ops

DEV = "cuda:0"
N, H, p = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 112, 64)
C4, M = 4 * p, N * H * H
g = torch.Generator().manual_seed(7 * p + H)
z2 = ops.Act((torch.relu(torch.randn(N, H, H, p, generator=g) * 0.8 + 0.2)).to(torch.bfloat16).to(DEV))
w = torch.randn(C4, p, 1, 1, generator=g) * (2.0 / p) ** 0.5
conv3 = torch.nn.Parameter(w.to(DEV).contiguous(memory_format=torch.channels_last))
w3 = ops.ConvWeights(conv3, 1, 1, 0)
w3.refresh()
bn3 = torch.nn.BatchNorm2d(C4).to(DEV)
with torch.no_grad():
    bn3.weight.copy_(torch.rand(C4, generator=g) + 0.5)
y3 = ops.new_act(N, H, H, C4, DEV)
st3 = ops.bn_finalize(ops.conv_fwd(z2, w3, y3, True), M, bn3)
x = ops.Act(torch.randn(N, H, H, C4, generator=g).to(torch.bfloat16).to(DEV))
out = ops.new_act(N, H, H, C4, DEV)
omask = ops.bn_apply(y3, st3, out, relu=True, residual=x, want_mask=True)
gout = ops.Act((torch.randn(N, H, H, C4, generator=g) * 1e-2).to(torch.bfloat16).to(DEV))
side = ops.SideStream(torch.device(DEV), enabled=False)
dy3 = ops.new_act(N, H, H, C4, DEV)
dg_ref, db_ref = ops.bn_backward(gout, y3, st3, bn3, dy3, 3, z=omask)
dz2_ref = ops.new_act(N, H, H, p, DEV)
ops.conv_dgrad(dy3, w3, dz2_ref)
dW_ref = ops.conv_wgrad(z2, dy3, w3).clone()
bits = (out.t > 0)
gm = ops.Act(torch.where(bits, gout.t, torch.zeros((), dtype=torch.bfloat16, device=DEV)))
gsum = gm.t.float().sum((0, 1, 2)).view(1, C4).contiguous()
dz2 = ops.new_act(N, H, H, p, DEV)
dg, db, dW = ops.bn3_algebra_backward(gm, gsum, z2, w3, st3, bn3, conv3, dz2, side)
side.join()
torch.cuda.synchronize()
# fp64 truth on the bf16 operands the kernels saw (W3 = the bf16 copy the forward multiplied with)
Z = z2.t.view(M, p).double()
W = w3.fwd.view(C4, p).double()
Y = Z @ W.t()
mu, var = Y.mean(0), Y.var(0, unbiased=False)
isd = 1.0 / torch.sqrt(var + 1e-5)
Gm = gm.t.view(M, C4).double()
xh = (Y - mu) * isd
dbt = Gm.sum(0)
dgt = (Gm * xh).sum(0)
k1 = bn3.weight.double() * isd
dY = k1 * (Gm - dbt / M - xh * dgt / M)
dWt = dY.t() @ Z
dzt = dY @ W


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


print("N=%d H=%d p=%d (M=%d)" % (N, H, p, M))
print("dW3    vs fp64 truth: passes %.3e   algebra %.3e   (passes vs algebra %.3e)" % (
    rel(dW_ref.reshape(C4, p), dWt), rel(dW.reshape(C4, p), dWt), rel(dW.reshape(C4, p), dW_ref.reshape(C4, p).double())))
print("dz2    vs fp64 truth: passes %.3e   algebra %.3e" % (rel(dz2_ref.t.view(M, p), dzt), rel(dz2.t.view(M, p), dzt)))
print("dgamma vs fp64 truth: passes %.3e   algebra %.3e" % (rel(dg_ref, dgt), rel(dg, dgt)))
print("dbeta  vs fp64 truth: passes %.3e   algebra %.3e" % (rel(db_ref, dbt), rel(db, dbt)))
