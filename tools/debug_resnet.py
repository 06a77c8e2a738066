"""Layer-by-layer comparison of the HIP ResNet executor against the CPU oracle (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import backbones as ob
from yolo_v1_amd.backbones.OriginResNet import resnet50

DEV = "cuda:0"
S, N, hw = 7, 4, 192
P = ob.init_params(ob.resnet50_param_shapes(S), "resnet", seed=11)
net = resnet50(S=S)
net.load_state_dict(P)
net = net.to(DEV).train()
x = torch.randn(N, 3, hw, hw, generator=torch.Generator().manual_seed(1))
with torch.no_grad():
    pred, rec = net._run_forward(x.to(DEV), True, True)
torch.cuda.synchronize()

def nchw(a, C=None):
    t = a.t.float().cpu()
    if C: t = t[..., :C]
    return t.permute(0, 3, 1, 2).contiguous()

def rep(name, got, want):
    d = (got - want).abs()
    print("%-28s max|d| %.4g  rel %.4g   |want| %.3g" % (name, d.max(), d.max() / (want.abs().max() + 1e-9), want.abs().max()))

P2 = ob.init_params(ob.resnet50_param_shapes(S), "resnet", seed=11)
xp, y0, s0, z0, H, W = rec["stem"][:6]
xb = x.to(torch.bfloat16).float()
wb = P2["conv1.weight"].to(torch.bfloat16).float()
r = F.conv2d(xb, wb, stride=2, padding=3)
rep("stem conv", nchw(y0), r)
rz = F.relu(F.batch_norm(nchw(y0), None, None, P2["bn1.weight"], P2["bn1.bias"], True))
rep("stem bn relu", nchw(z0), rz)
blk0 = rec["blocks"][0]
rep("maxpool", nchw(blk0[1]), F.max_pool2d(nchw(z0), 3, 2, 1))
names = []
for st in net._stage_names:
    for i in range(len(getattr(net, st))):
        names.append("%s.%d" % (st, i))
for name, (blk, xin, y1, s1, z1, y2, s2, z2, y3, s3, yd, sd, out, _m) in zip(names, rec["blocks"]):
    xi = nchw(xin)
    w1 = P2[name + ".conv1.weight"].to(torch.bfloat16).float()
    rep(name + " y1", nchw(y1), F.conv2d(xi, w1))
    rz1 = F.relu(F.batch_norm(nchw(y1), None, None, P2[name + ".bn1.weight"], P2[name + ".bn1.bias"], True))
    rep(name + " z1", nchw(z1), rz1)
    w2 = P2[name + ".conv2.weight"].to(torch.bfloat16).float()
    rep(name + " y2", nchw(y2), F.conv2d(nchw(z1), w2, stride=blk.stride, padding=1))
    rz2 = F.relu(F.batch_norm(nchw(y2), None, None, P2[name + ".bn2.weight"], P2[name + ".bn2.bias"], True))
    rep(name + " z2", nchw(z2), rz2)
    w3 = P2[name + ".conv3.weight"].to(torch.bfloat16).float()
    rep(name + " y3", nchw(y3), F.conv2d(nchw(z2), w3))
    b3 = F.batch_norm(nchw(y3), None, None, P2[name + ".bn3.weight"], P2[name + ".bn3.bias"], True)
    if yd is not None:
        wd = P2[name + ".downsample.0.weight"].to(torch.bfloat16).float()
        rep(name + " yd", nchw(yd), F.conv2d(xi, wd, stride=blk.stride))
        idt = F.batch_norm(nchw(yd), None, None, P2[name + ".downsample.1.weight"], P2[name + ".downsample.1.bias"], True)
    else:
        idt = xi
    rep(name + " out", nchw(out), F.relu(b3 + idt))
xh, yh, sh, pr = rec["head"]
wh = P2["layer6.weight"].to(torch.bfloat16).float()
rep("head conv", nchw(yh, 30), F.conv2d(nchw(xh), wh))
rp = torch.sigmoid(F.batch_norm(nchw(yh, 30), None, None, P2["bn_end.weight"], P2["bn_end.bias"], True)).permute(0, 2, 3, 1)
rep("pred", pr.cpu(), rp)
