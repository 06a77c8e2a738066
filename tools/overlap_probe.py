"""Can a weight-gradient kernel stream overlap with a chain of BatchNorm kernels inside one captured hipGraph?  Main chain:
n x bn_apply on a [64,112,112,256] tensor (HBM-bound, ~230 us each); side stream: one 1x1 weight gradient per `every` main
kernels, forked with the event pattern of ops.SideStream.  Prints the graph replay time against the two parts alone
(no profiler attached).  python tools/overlap_probe.py [--n 60] [--every 1] [--side 256,1024,28 | 64,256,112 ...]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v1_amd import ops
from yolo_v1_amd.engine import ConvParam
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=60); ap.add_argument("--every", type=int, default=1)
ap.add_argument("--side", default="256,1024,28"); ap.add_argument("--main", default="bn", help="bn | conv")
ap.add_argument("--one-fork", type=int, default=0, help="1: the side stream waits for the main stream once, at the start")
a = ap.parse_args()
dev = "cuda:0"; N = 64
cin, cout, H = [int(v) for v in a.side.split(",")]
conv = ConvParam(cin, cout, 1).to(dev)
w = ops.ConvWeights(conv.weight, conv.kernel_size, conv.stride, conv.padding); w.refresh()
xs = ops.Act(torch.randn(N, H, H, cin, device=dev).to(torch.bfloat16))
dys = ops.Act(torch.randn(N, H, H, cout, device=dev).to(torch.bfloat16))
y = ops.Act(torch.randn(N, 112, 112, 256, device=dev).to(torch.bfloat16)); z = ops.new_act(N, 112, 112, 256, dev)
bn = torch.nn.BatchNorm2d(256).to(dev)
st = ops.bn_finalize(ops.bn_stats(y), y.npix, bn)
c2 = ConvParam(256, 256, 3).to(dev); w2 = ops.ConvWeights(c2.weight, c2.kernel_size, c2.stride, c2.padding); w2.refresh()
x28 = ops.Act(torch.randn(N, 28, 28, 256, device=dev).to(torch.bfloat16)); y28 = ops.new_act(N, 28, 28, 256, dev)
def main_op():
    if a.main == "bn":
        ops.bn_apply(y, st, z, relu=True)
    else:
        ops.conv_fwd(x28, w2, y28, False)
def chain(with_main, with_side):
    side = ops.SideStream(torch.device(dev), enabled=True)
    if a.one_fork and with_side:
        mk0 = side.mark()
    for i in range(a.n):
        mk = side.mark()
        if with_main:
            main_op()
        if with_side and (i + 1) % a.every == 0:
            ops.conv_wgrad(xs, dys, w, side, after=(mk0 if a.one_fork else mk))
    side.join()
def timed(with_main, with_side):
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        chain(with_main, with_side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain(with_main, with_side)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)
tm, tsd, tb = timed(True, False), timed(False, True), timed(True, True)
print("main=%s x%d, side=wgrad %s every %d, one_fork=%d:  main alone %.2f ms, side alone %.2f ms, together %.2f ms  -> hidden %.0f %% of the side work" % (
    a.main, a.n, a.side, a.every, a.one_fork, tm, tsd, tb, 100 * (tm + tsd - tb) / tsd))
