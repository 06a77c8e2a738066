import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import backbones as ob, fp8 as o8
from yolo_v1_amd.backbones.OriginResNet import resnet50
from yolo_v1_amd.infer_fp8 import ResNetFp8
dev = torch.device("cuda:0")
S = 14
torch.manual_seed(0)
net = resnet50(S=S)
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith("bn3.weight"): p.mul_(0.2)
P = {k: v.detach().clone() for k, v in net.state_dict().items()}
x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(7))
old = ob._bn.__defaults__; ob._bn.__defaults__ = (1.0, 1e-5)
with torch.no_grad(): ob.resnet50_forward(x, P, S=S, training=True)
ob._bn.__defaults__ = old
net.load_state_dict(P); net = net.to(dev).eval()
tr_o = []
with torch.no_grad():
    want8 = o8.resnet50_eval_fp8(x, P, S, trace=tr_o)
    want32 = ob.resnet50_forward(x, P, S=S, training=False)
eng = ResNetFp8(net); eng.trace = []
got = eng(x.to(dev)).cpu()
for (ln, to), (lg, tg) in zip(tr_o, eng.trace):
    g = tg.cpu().view(torch.float8_e4m3fn).float().permute(0, 3, 1, 2)
    d = (g - to).abs()
    print("%-12s %-8s mismatch %.4f  mean|d| %.5f  mean|v| %.4f  max|v| %.2f" % (ln, lg, (d > 0).float().mean(), d.mean(), to.abs().mean(), to.abs().max()))
print("pred: fp8 vs oracle8 max %.4f mean %.5f | fp8 vs fp32 mean %.5f | oracle8 vs fp32 mean %.5f" % (
    (got - want8).abs().max(), (got - want8).abs().mean(), (got - want32).abs().mean(), (want8 - want32).abs().mean()))
with torch.no_grad(): got16 = net(x.to(dev)).cpu()
print("bf16 eval vs fp32 mean %.5f" % (got16 - want32).abs().mean())
