import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v1_amd.backbones.OriginResNet import resnet50
from yolo_v1_amd.optim import FusedSGD
from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
from yolo_v1_amd.v1Loss import YOLOLossV1
DEV = "cuda:0"
images, target = synthetic_batch(4, 2, hw=128, device=DEV)
torch.manual_seed(0)
a = resnet50(S=7).to(DEV).train()
b = resnet50(S=7).to(DEV).train()
b.load_state_dict(a.state_dict())
cs = lambda n: float(sum(p.detach().double().sum() for p in n.parameters()))
gcs = lambda n: float(sum(p.grad.detach().double().abs().sum() for p in n.parameters()))
def body(net, crit, opt):
    pred = net(images); loss = crit(pred, target); opt.zero_grad(); loss.backward(); opt.step(); return loss
oa = FusedSGD(a.parameters(), lr=1e-3, momentum=0.99)
ob_ = FusedSGD(b.parameters(), lr=1e-3, momentum=0.99)
la, lb = YOLOLossV1(4, 2, 2, 20, _quiet=True), YOLOLossV1(4, 2, 2, 20, _quiet=True)
l = body(a, la, oa); torch.cuda.synchronize()
print("a eager: loss", l.item(), "gradsum", gcs(a), "checksum", cs(a))
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    l = body(b, lb, ob_)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("b side-stream eager: loss", l.item(), "gradsum", gcs(b), "checksum", cs(b))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    lg = body(b, lb, ob_)
torch.cuda.synchronize()
print("b after capture: checksum", cs(b))
l2 = body(a, la, oa); torch.cuda.synchronize()
print("a eager step2: loss", l2.item(), "checksum", cs(a))
g.replay(); torch.cuda.synchronize()
print("b replay: loss", lg.item(), "checksum", cs(b))
