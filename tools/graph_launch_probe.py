"""Host-side cost of one hipGraph replay of the training step: how long graph.replay() keeps the calling thread busy
(the HIP runtime builds and submits every node's AQL packet inside hipGraphLaunch, stream list after stream list) against
the GPU time of the step.  python tools/graph_launch_probe.py [--backbone resnet|densenet] [--S 7]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="resnet"); ap.add_argument("--S", type=int, default=7); ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
from yolo_v1_amd.train import GraphedStep
from yolo_v1_amd.optim import FusedSGD
from yolo_v1_amd.v1Loss import YOLOLossV1
from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
if a.backbone == "resnet":
    from yolo_v1_amd.backbones.OriginResNet import resnet50 as ctor
else:
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121 as ctor
dev = "cuda:0"
torch.manual_seed(0)
net = ctor(S=a.S).to(dev).train()
opt = FusedSGD(net.parameters(), lr=1e-3, momentum=0.99)
crit = YOLOLossV1(a.batch, a.S, 2, 20, _quiet=True)
x, t = synthetic_batch(a.batch, a.S, seed=1)
gs = GraphedStep(net, crit, opt, x.to(dev), t.to(dev))
for _ in range(3):
    gs(1e-3)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gs(1e-3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print("graph replay: host call returns after %.2f ms (median), step complete after %.2f ms" % (host[5], total[5]))
# back-to-back replays: steady state
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    gs(1e-3)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("10 back-to-back replays: host %.2f ms per call, %.2f ms per step" % ((t1 - t0) * 100, (t2 - t0) * 100))
# the learning rate changes every iteration (train.py:22-32 warm-up): an eager fill of the device-side lr between replays
for name, lrs in (("constant lr", [1e-3] * 30), ("lr changing every step", [1e-3 + 1e-6 * i for i in range(30)])):
    for lr in lrs[:5]:
        gs(lr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for lr in lrs[5:]:
        gs(lr)
    torch.cuda.synchronize()
    print("%-24s %.3f ms per step" % (name, (time.perf_counter() - t0) * 1e3 / 25))
