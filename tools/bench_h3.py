"""3x3 stride-1 convolutions (ResNet-50 Bottleneck conv2, DenseNet-121 growth convolutions) at batch 64: forward and data
gradient under the current YV1_CONV_H3 / YV1_CONV_H3_NST setting.  Run once per setting and compare:
  YV1_CONV_H3=0 python tools/bench_h3.py ; python tools/bench_h3.py ; YV1_CONV_H3_NST=3 python tools/bench_h3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v1_amd import _lib, ops
DEV = "cuda:0"
N = 64
SHAPES = [(64, 64, 112), (128, 128, 56), (256, 256, 28), (512, 512, 14), (512, 512, 7),
          (128, 32, 112), (128, 32, 56), (128, 32, 28), (128, 32, 14)]


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("H3=%s NST=%s" % (os.environ.get("YV1_CONV_H3", "1"), os.environ.get("YV1_CONV_H3_NST", "-")))
for ci, co, h in SHAPES:
    x = ops.Act(torch.randn(N, h, h, ci, device=DEV).to(torch.bfloat16))
    w = torch.nn.Parameter((torch.randn(co, ci, 3, 3, device=DEV) * 0.05).contiguous(memory_format=torch.channels_last))
    cw = ops.ConvWeights(w, 3, 1, 1); cw.refresh()
    y = ops.new_act(N, h, h, cw.Opad, DEV)
    dy = ops.Act(torch.randn(N, h, h, cw.Opad, device=DEV).to(torch.bfloat16))
    dx = ops.new_act(N, h, h, ci, DEV)
    tf = timeit(lambda: ops.conv_fwd(x, cw, y, True)); cf = _lib.last_config()
    td = timeit(lambda: ops.conv_dgrad(dy, cw, dx)); cd = _lib.last_config()
    fl = 2.0 * N * h * h * co * ci * 9
    print("%4d->%4d @%3d  fwd %7.1f us %5.0f TF/s %-34s | dgrad %7.1f us %5.0f TF/s %s" % (
        ci, co, h, tf, fl / tf / 1e6, cf[0], td, fl / td / 1e6, cd[0]), flush=True)
