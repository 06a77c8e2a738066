import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v1_amd.backbones.OriginResNet import resnet50
from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
from yolo_v1_amd.v1Loss import YOLOLossV1
DEV = "cuda:0"
images, target = synthetic_batch(4, 2, hw=128, device=DEV)
torch.manual_seed(0)
a = resnet50(S=7).to(DEV).train()
crit = YOLOLossV1(4, 2, 2, 20, _quiet=True)
res = []
for rep in range(3):
    for p in a.parameters():
        p.grad = None
    pred = a(images)
    loss = crit(pred, target)
    loss.backward()
    torch.cuda.synchronize()
    res.append((loss.item(), pred.detach().clone(), {k: p.grad.clone() for k, p in a.named_parameters()}))
print("losses", [r[0] for r in res])
print("pred equal", torch.equal(res[0][1], res[1][1]), torch.equal(res[1][1], res[2][1]))
bad = [k for k in res[0][2] if not torch.equal(res[0][2][k], res[1][2][k])]
print("grads differing between run0/run1:", len(bad), bad[:12])
bad = [k for k in res[1][2] if not torch.equal(res[1][2][k], res[2][2][k])]
print("grads differing between run1/run2:", len(bad), bad[:12])
