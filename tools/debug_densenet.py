"""Teacher-forced layer-by-layer check of the HIP DenseNet executor (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import backbones as ob
from yolo_v1_amd.backbones.OriginDenseNet import densenet121
DEV = "cuda:0"
S, N, hw = 7, 4, 192
P = ob.init_params(ob.densenet121_param_shapes(S), "densenet", seed=5)
net = densenet121(S=S); net.load_state_dict(P); net = net.to(DEV).train()
x = torch.randn(N, 3, hw, hw, generator=torch.Generator().manual_seed(2))
with torch.no_grad():
    pred, rec = net._run_forward(x.to(DEV), True, True)
torch.cuda.synchronize()
bfw = lambda k: P[k].to(torch.bfloat16).float()
def nchw(a, c0=0, C=None):
    t = a.t.float().cpu()
    C = C if C is not None else a.C
    return t[..., a.c0 + c0: a.c0 + c0 + C].permute(0, 3, 1, 2).contiguous()
worst = {}
def rep(name, got, want, show=False):
    rel = float((got - want).abs().max() / (want.abs().max() + 1e-9))
    key = name.split(" ")[-1]
    if rel > worst.get(key, (0, ""))[0]:
        worst[key] = (rel, name)
    if show or rel > 1e-2:
        print("%-50s rel %.4g |want| %.3g" % (name, rel, float(want.abs().max())))
bnf = lambda t, k: F.batch_norm(t, None, None, P[k + ".weight"], P[k + ".bias"], True)
xp, y0, s0, z0, H, W = rec["stem"][:6]
rep("stem conv", nchw(y0), F.conv2d(x.to(torch.bfloat16).float(), bfw("features.conv0.weight"), stride=2, padding=3), True)
rep("stem bn", nchw(z0), F.relu(bnf(nchw(y0), "features.norm0")), True)
bi = 0
prev_yc = None
for st in rec["stages"]:
    if st[0] == "block":
        bi += 1
        _, buf, lrecs, nf = st
        first = nchw(buf, 0, nf)
        if bi == 1:
            rep("pool0", first, F.max_pool2d(nchw(z0), 3, 2, 1), True)
        else:
            rep("avgpool%d" % bi, first, F.avg_pool2d(prev_yc, 2, 2), True)
        for li, (layer, cin, st1, t1, y1, st2, t2) in enumerate(lrecs):
            p = "features.denseblock%d.denselayer%d" % (bi, li + 1)
            rep(p + " t1", nchw(t1), F.relu(bnf(nchw(buf, 0, cin), p + ".norm1")))
            rep(p + " y1", nchw(y1), F.conv2d(nchw(t1), bfw(p + ".conv1.weight")))
            rep(p + " t2", nchw(t2), F.relu(bnf(nchw(y1), p + ".norm2")))
            rep(p + " h", nchw(buf, cin, 32), F.conv2d(nchw(t2), bfw(p + ".conv2.weight"), padding=1))
    else:
        _, tr, buf, stt, t, yc = st
        p = "features.transition%d" % bi
        rep(p + " t", nchw(t), F.relu(bnf(nchw(buf), p + ".norm")), True)
        rep(p + " yc", nchw(yc), F.conv2d(nchw(t), bfw(p + ".conv.weight")), True)
        prev_yc = nchw(yc)
buf, st5, t5, yh, sh, pr = rec["head"]
rep("norm5 t5", nchw(t5), F.relu(bnf(nchw(buf), "features.norm5")), True)
rep("head conv", nchw(yh, 0, 30), F.conv2d(nchw(t5), bfw("layer6.weight")), True)
rp = torch.sigmoid(bnf(nchw(yh, 0, 30), "bn_end")).permute(0, 2, 3, 1)
rep("pred", pr.cpu(), rp, True)
print("worst per kind:", worst)
