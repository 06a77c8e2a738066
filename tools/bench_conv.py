"""Per-layer micro-bench of the conv kernels on the ResNet-50 (448x448, N=64) shapes: fwd / dgrad / wgrad."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v1_amd import ops
DEV = "cuda:0"
N = int(os.environ.get("BN", 64))
# (Cin, Cout, k, stride, Hin)  unique ResNet-50 S=7 shapes
SHAPES = [(64, 64, 1, 1, 112), (64, 64, 3, 1, 112), (64, 256, 1, 1, 112), (256, 64, 1, 1, 112), (256, 128, 1, 1, 112),
          (128, 128, 3, 2, 112), (128, 512, 1, 1, 56), (256, 512, 1, 2, 112), (512, 128, 1, 1, 56), (128, 128, 3, 1, 56),
          (512, 256, 1, 1, 56), (256, 256, 3, 2, 56), (256, 1024, 1, 1, 28), (512, 1024, 1, 2, 56), (1024, 256, 1, 1, 28),
          (256, 256, 3, 1, 28), (1024, 512, 1, 1, 28), (512, 512, 3, 2, 28), (512, 2048, 1, 1, 14), (1024, 2048, 1, 2, 28),
          (2048, 512, 1, 1, 14), (512, 512, 3, 1, 14), (512, 512, 3, 2, 14), (512, 2048, 1, 1, 7), (2048, 2048, 1, 2, 14),
          (2048, 512, 1, 1, 7), (512, 512, 3, 1, 7)]
COUNT = {0: 1, 1: 3, 2: 4, 3: 2, 4: 1, 5: 1, 6: 4, 7: 1, 8: 3, 9: 3, 10: 1, 11: 1, 12: 6, 13: 1, 14: 5, 15: 5, 16: 1, 17: 1,
         18: 3, 19: 1, 20: 3, 21: 2, 22: 1, 23: 3, 24: 1, 25: 2, 26: 2}
which = sys.argv[1] if len(sys.argv) > 1 else "all"

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print("%-28s %8s %8s %8s | TF/s fwd dgr wgr | GB/s fwd" % ("shape", "fwd us", "dgrad", "wgrad"))
for i, (ci, co, k, st, h) in enumerate(SHAPES):
    pad = 1 if k == 3 else 0
    oh = (h + 2 * pad - k) // st + 1
    x = ops.Act(torch.randn(N, h, h, ci, device=DEV).to(torch.bfloat16))
    w = torch.nn.Parameter((torch.randn(co, ci, k, k, device=DEV) * 0.05).contiguous(memory_format=torch.channels_last))
    cw = ops.ConvWeights(w, k, st, pad); cw.refresh()
    y = ops.new_act(N, oh, oh, co, DEV)
    dy = ops.Act(torch.randn(N, oh, oh, co, device=DEV).to(torch.bfloat16))
    dx = ops.new_act(N, h, h, ci, DEV)
    t_f = timeit(lambda: ops.conv_fwd(x, cw, y, True)) if which in ("all", "fwd") else 0
    t_d = timeit(lambda: ops.conv_dgrad(dy, cw, dx)) if which in ("all", "dgrad") else 0
    t_w = timeit(lambda: ops.conv_wgrad(x, dy, cw)) if which in ("all", "wgrad") else 0
    fl = 2.0 * N * oh * oh * co * ci * k * k
    by = 2.0 * N * (h * h * ci + oh * oh * co)
    c = COUNT[i]
    tot["fwd"] += c * t_f; tot["dgrad"] += c * t_d; tot["wgrad"] += c * t_w
    tf = lambda t: fl / t / 1e6 if t else 0
    print("%4d->%4d k%d s%d @%3d x%d      %8.1f %8.1f %8.1f | %5.0f %5.0f %5.0f | %5.0f" % (
        ci, co, k, st, h, c, t_f, t_d, t_w, tf(t_f), tf(t_d), tf(t_w), by / t_f / 1e3 if t_f else 0))
print("network totals (ms): fwd %.2f dgrad %.2f wgrad %.2f" % (tot["fwd"] / 1e3, tot["dgrad"] / 1e3, tot["wgrad"] / 1e3))
