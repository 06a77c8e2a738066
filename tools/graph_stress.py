"""How many full-size training-step hipGraphs can one process build and replay?  (Investigation of a segfault inside
hipGraphLaunch seen when the GPU test suite captured two batch-64 steps on top of ~20 smaller graphs.)
python tools/graph_stress.py [count] [keep]   -- keep=1 keeps every GraphedStep alive."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yolo_v1_amd.train import GraphedStep, build
from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
keep = len(sys.argv) > 2 and sys.argv[2] == "1"
dev = torch.device("cuda:0")
alive = []
for i in range(n):
    bb, S = (("resnet", 7), ("densenet", 7), ("resnet", 14))[i % 3]
    net, loss_layer, opt = build(bb, S, 2, 20, 64, dev, quiet=True, fused_optimizer=True)
    images, target = synthetic_batch(64, S, seed=1, device=dev)
    gs = GraphedStep(net, loss_layer, opt, images, target, None, warmup=1)
    l = [float(gs(1e-6).item()) for _ in range(2)]
    print("graph %d (%s S=%d): losses %s, reserved %.1f GB" % (i, bb, S, l, torch.cuda.memory_reserved() / 1e9), flush=True)
    if keep:
        alive.append(gs)
    else:
        del gs, net, opt, loss_layer
        gc.collect(); torch.cuda.empty_cache()
print("done")
