"""Times one conv shape (fwd) under the current env: python tools/bench_one_conv.py Cin Cout k stride H [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v1_amd import ops
ci, co, k, st, h = [int(v) for v in sys.argv[1:6]]
N = int(sys.argv[6]) if len(sys.argv) > 6 else 64
pad = 1 if k == 3 else 0
oh = (h + 2 * pad - k) // st + 1
DEV = "cuda:0"
x = ops.Act(torch.randn(N, h, h, ci, device=DEV).to(torch.bfloat16))
w = torch.nn.Parameter((torch.randn(co, ci, k, k, device=DEV) * 0.05).contiguous(memory_format=torch.channels_last))
cw = ops.ConvWeights(w, k, st, pad); cw.refresh()
y = ops.new_act(N, oh, oh, co, DEV)
for _ in range(3): ops.conv_fwd(x, cw, y, True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.conv_fwd(x, cw, y, True)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e3
print("%d->%d k%d s%d @%d N=%d: %.1f us  %.0f TF/s" % (ci, co, k, st, h, N, t, 2.0 * N * oh * oh * co * ci * k * k / t / 1e6))
