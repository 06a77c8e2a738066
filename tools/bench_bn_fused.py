"""BatchNorm finalize fused into the apply launch against the two-launch form, per ResNet-50 tensor shape (batch 64):
forward (finalize + apply) and backward (reduce + finalize + apply), eager launches timed with HIP events over a loop and
as a captured hipGraph chain (what the training step replays).  python tools/bench_bn_fused.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v1_amd import ops
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SHAPES = [(256, 112), (64, 112), (512, 56), (128, 56), (1024, 28), (256, 28), (2048, 14), (512, 14), (2048, 7), (512, 7)]


def timed(fn, graph):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        g.reset()
        return ms / iters * 1e3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("%-10s | %28s | %28s" % ("C@H", "forward us: 2 launches / fused", "backward us: 3 launches / 2"))
for C, H in SHAPES:
    N = 64
    y = ops.Act(torch.randn(N, H, H, C, device=dev).to(torch.bfloat16))
    z = ops.new_act(N, H, H, C, dev)
    dz = ops.Act(torch.randn(N, H, H, C, device=dev).to(torch.bfloat16))
    dy = ops.new_act(N, H, H, C, dev)
    bn = torch.nn.BatchNorm2d(C).to(dev)
    part = ops.bn_stats(y)
    st = ops.bn_finalize(part, y.npix, bn)
    row = []
    for graph in (False, True):
        def fwd2():
            s = ops.bn_finalize(part, y.npix, bn)
            ops.bn_apply(y, s, z, relu=True)
        def fwd1():
            ops.bn_finalize_apply(part, y.npix, bn, y, z, relu=True)
        def bwd():
            ops.bn_backward(dz, y, st, bn, dy, 2)
        a, b = timed(fwd2, graph), timed(fwd1, graph)
        ops.BN_FUSED = False
        c = timed(bwd, graph)
        ops.BN_FUSED = True
        d = timed(bwd, graph)
        row.append("%s %7.1f / %7.1f | %7.1f / %7.1f" % ("graph" if graph else "eager", a, b, c, d))
    print("%4d@%-4d | %s || %s   rows %d" % (C, H, row[0], row[1], part.shape[0]), flush=True)
print("fault word:", ops.fused_sync_fault())
