import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v1_amd.backbones.OriginResNet import resnet50
from yolo_v1_amd.optim import FusedSGD
from yolo_v1_amd.train import GraphedStep, train_step
from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
from yolo_v1_amd.v1Loss import YOLOLossV1
DEV = "cuda:0"
images, target = synthetic_batch(4, 2, hw=128, device=DEV)
torch.manual_seed(0)
a = resnet50(S=7).to(DEV).train()
b = resnet50(S=7).to(DEV).train()
b.load_state_dict(a.state_dict())
cs = lambda n: float(sum(p.double().sum() for p in n.parameters()))
print("init checksum", cs(a), cs(b))
oa = FusedSGD(a.parameters(), lr=0.0, momentum=0.99)
ob_ = FusedSGD(b.parameters(), lr=0.0, momentum=0.99)
la, lb = YOLOLossV1(4, 2, 2, 20, _quiet=True), YOLOLossV1(4, 2, 2, 20, _quiet=True)
l0 = train_step(a, la, oa, images, target, 1e-3).item()
print("a step0 loss", l0, "checksum after", cs(a))
l1 = train_step(a, la, oa, images, target, 2e-3).item()
print("a step1 loss", l1, "checksum after", cs(a))
for grp in ob_.param_groups:
    grp['lr'] = 1e-3
gs = GraphedStep(b, lb, ob_, images, target, warmup=1)
torch.cuda.synchronize()
print("b after warmup+capture checksum", cs(b), "lr dev", float(ob_._lr_dev[0]))
r = gs(2e-3).item()
print("b replay1 loss", r, "checksum after", cs(b))
