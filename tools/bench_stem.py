"""Stem BatchNorm/ReLU/max-pool kernels at the ResNet/DenseNet stem size (N=64, 224x224x64): fused forward launch and the
pooled BatchNorm backward pair against the unfused chains they replace.  HIP-event timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v1_amd import ops
DEV = "cuda:0"
N, H, W, C = int(os.environ.get("BN", 64)), 224, 224, 64
y = ops.Act(torch.randn(N, H, W, C, device=DEV).to(torch.bfloat16))
bn = torch.nn.BatchNorm2d(C).to(DEV)
st = ops.bn_finalize(ops.bn_stats(y), y.npix, bn)
z = ops.new_act(N, H, W, C, DEV)
pooled = ops.new_act(N, H // 2, W // 2, C, DEV)
ops.bn_apply(y, st, z, relu=True)
pidx = ops.maxpool_fwd(z, pooled, want_index=True)
gp = ops.Act(torch.randn(N, H // 2, W // 2, C, device=DEV).to(torch.bfloat16))
dz, dy = ops.new_act(N, H, W, C, DEV), ops.new_act(N, H, W, C, DEV)


def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def fwd_chain():
    ops.bn_apply(y, st, z, relu=True)
    ops.maxpool_fwd(z, pooled, want_index=True)


def bwd_chain():
    ops.maxpool_bwd(None, gp, dz, pidx)
    ops.bn_backward(dz, y, st, bn, dy, 2)


print("forward : bn_apply + maxpool %.1f us   fused %.1f us" % (t(fwd_chain), t(lambda: ops.bn_act_maxpool_fwd(y, st, pooled, True, True))))
print("backward: maxpool_bwd + bn_backward %.1f us   pooled bn_backward %.1f us" % (t(bwd_chain), t(lambda: ops.bn_backward(gp, y, st, bn, dy, 2, pool_idx=pidx))))
