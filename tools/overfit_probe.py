"""Sanity of the restructured backward paths as a TRAINING run, not a parity check: a captured step (hipGraph replay, fused
SGD momentum 0.9) on ONE fixed synthetic batch for a few hundred iterations -- the loss has to fall monotonically-ish and stay
finite.    python tools/overfit_probe.py [resnet|densenet] [steps] [batch] [hw]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolo_v1_amd.optim import FusedSGD
from yolo_v1_amd.train import GraphedStep
from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
from yolo_v1_amd.v1Loss import YOLOLossV1

kind = sys.argv[1] if len(sys.argv) > 1 else "resnet"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
hw = int(sys.argv[4]) if len(sys.argv) > 4 else 448
if kind == "resnet":
    from yolo_v1_amd.backbones.OriginResNet import resnet50 as ctor
else:
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121 as ctor
dev = "cuda:0"
S = hw // 64
images, target = synthetic_batch(batch, S, hw=hw, device=dev)
torch.manual_seed(0)
net = ctor(S=7).to(dev).train()
opt = FusedSGD(net.parameters(), lr=0.0, momentum=0.9)
with GraphedStep(net, YOLOLossV1(batch, S, 2, 20, _quiet=True), opt, images, target, warmup=1, preserve_state=True) as gs:
    hist = []
    for it in range(steps):
        loss = gs(1e-3)
        if it % max(1, steps // 10) == 0 or it == steps - 1:
            hist.append((it, float(loss.item())))
print(kind, " ".join("%d:%.3f" % h for h in hist))
ok = all(l == l for _, l in hist) and hist[-1][1] < 0.5 * hist[0][1]
print("finite and falling:", ok)
sys.exit(0 if ok else 1)
