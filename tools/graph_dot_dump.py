"""Dumps the captured training step's hipGraph as DOT (hipGraphDebugDotPrint through torch's CUDAGraph.debug_dump) so that
its shape -- how many chains the runtime can run side by side -- can be analysed off the box (tools/graph_dot_width.py).
    python tools/graph_dot_dump.py <out.dot> [resnet|densenet] [S] [batch] [hw]
Environment switches (YV1_BN3_ALGEBRA, YV1_WGRAD_SIDE_STREAM, ...) select the variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

out = os.path.abspath(sys.argv[1])
kind = sys.argv[2] if len(sys.argv) > 2 else "resnet"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 7
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 4
hw = int(sys.argv[5]) if len(sys.argv) > 5 else 128

_begin = torch.cuda.CUDAGraph.capture_begin


def capture_begin(self, *a, **k):
    self.enable_debug_mode()
    return _begin(self, *a, **k)


torch.cuda.CUDAGraph.capture_begin = capture_begin

from yolo_v1_amd.optim import FusedSGD
from yolo_v1_amd.train import GraphedStep
from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
from yolo_v1_amd.v1Loss import YOLOLossV1

if kind == "resnet":
    from yolo_v1_amd.backbones.OriginResNet import resnet50 as ctor
else:
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121 as ctor
dev = "cuda:0"
grid = hw // 64 if hw != 448 else S
images, target = synthetic_batch(batch, grid, hw=hw, device=dev)
torch.manual_seed(0)
net = ctor(S=S).to(dev).train()
opt = FusedSGD(net.parameters(), lr=0.0, momentum=0.99)
gs = GraphedStep(net, YOLOLossV1(batch, grid, 2, 20, _quiet=True), opt, images, target, warmup=1)
gs.graph.debug_dump(out)
print("loss", float(gs(0.0).item()), "dot written to", out, flush=True)
gs.close()
