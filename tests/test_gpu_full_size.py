"""BASELINE.json configs 3 and 5 at their full size on the GPU ("DenseNet-121 at 448^2 N=64 and the S=14
training step at N=64 are never run by a -m gpu test").

At batch 64 / 448x448 the CPU oracle can still follow the FORWARD (training-mode BatchNorm over the whole batch; bf16
storage emulated) in seconds: DenseNet-121 S=7 (OriginDenseNet.py:114-129, 1.5 TFLOP) and ResNet-50 S=14
(OriginResNet.py:173-195, 2.1 TFLOP).  The HIP executors run at the tile configurations the bench dispatches
(per-layer element-wise parity of those: test_gpu_bench_configs.py); here the whole network is compared end to end
with the tolerances of the reduced-size whole-net tests (sigmoid outputs: ResNet max 5e-2 / mean 1e-2, DenseNet max
2e-1 / mean 4e-2), then two training steps (forward + loss + backward + fused SGD, captured into a hipGraph and replayed --
the form bench.py times) run at that size: finite gradients for every parameter, the first step's loss equals the forward's,
the second differs (the weights moved).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("backbone,S,max_tol,mean_tol", [("densenet", 7, 2e-1, 4e-2), ("resnet", 14, 5e-2, 1e-2)])
def test_full_size_forward_vs_oracle_and_training_step(backbone, S, max_tol, mean_tol):
    from oracle import backbones as ob
    from oracle import loss as ol
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    from yolo_v1_amd.v1Loss import YOLOLossV1
    N = 64
    print("\nGPU memory at test start: reserved %.1f GB, allocated %.1f GB" % (torch.cuda.memory_reserved() / 1e9,
                                                                            torch.cuda.memory_allocated() / 1e9))
    if backbone == "densenet":
        from yolo_v1_amd.backbones.OriginDenseNet import densenet121 as ctor
        P = ob.init_params(ob.densenet121_param_shapes(S), "densenet", seed=5)
        fwd = ob.densenet121_forward
    else:
        from yolo_v1_amd.backbones.OriginResNet import resnet50 as ctor
        P = ob.init_params(ob.resnet50_param_shapes(S), "resnet", seed=11)
        for k in P:
            if k.endswith("bn3.weight"):
                P[k] = P[k] * 0.2
        fwd = ob.resnet50_forward
    net = ctor(S=S)
    net.load_state_dict(P, strict=True)
    net = net.to(DEV).train()
    images, target = synthetic_batch(N, S, seed=77)
    with torch.no_grad():
        ref = fwd(images, {k: v.clone() for k, v in P.items()}, S, training=True, q=ob.bf16_ste)
        pred = net(images.to(DEV))
    assert tuple(pred.shape) == (N, S, S, 30)
    d = (pred.cpu() - ref).abs()
    print("\n%s S=%d N=64 448x448: sigmoid outputs max |err| %.3g mean %.3g" % (backbone, S, d.max().item(), d.mean().item()))
    assert d.max().item() <= max_tol and d.mean().item() <= mean_tol
    ref_loss, _ = ol.yolo_loss(ref, target, S, 2, 20, batch_size=N)
    loss = YOLOLossV1(N, S, 2, 20, _quiet=True)(pred, target.to(DEV))
    np.testing.assert_allclose(loss.item(), float(ref_loss), rtol=5e-2)
    # two training steps at full size, CAPTURED (one hipGraph: forward + loss + backward + fused SGD) and replayed -- the
    # form bench.py times and train.main runs.  Round 2 this capture segfaulted inside hipGraphLaunch at this point of the
    # suite; DESIGN.md section 4 has the diagnosis (hipGraphExec objects of earlier tests still alive) -- every captured
    # step is now closed deterministically, and the number of execs still alive is printed and bounded here.
    net.load_state_dict(P, strict=True)
    from yolo_v1_amd import ops
    from yolo_v1_amd.train import GraphedStep
    import gc
    gc.collect()
    live = GraphedStep.live_graphs()
    raw = sum(1 for o in gc.get_objects() if isinstance(o, torch.cuda.CUDAGraph))
    print("hipGraphExec objects alive before the full-size capture: %d owned by GraphedStep, %d torch.cuda.CUDAGraph in all"
          % (live, raw))
    assert live == 0, "an earlier test left %d captured training steps alive (GraphedStep.close() missing)" % live
    ops.bump_weight_epoch()
    opt = FusedSGD(net.parameters(), lr=1e-3, momentum=0.99)
    crit = YOLOLossV1(N, S, 2, 20, _quiet=True)
    xd, td = images.to(DEV), target.to(DEV)
    with GraphedStep(net, crit, opt, xd, td, None, warmup=1, preserve_state=True) as gs:
        l1 = float(gs(1e-3).item())
        l2 = float(gs(1e-3).item())
    assert GraphedStep.live_graphs() == 0
    assert np.isfinite(l1) and np.isfinite(l2) and l1 != l2
    np.testing.assert_allclose(l1, loss.item(), rtol=1e-5)            # same weights, same batch: the step's loss is the forward's
    assert all(torch.isfinite(p).all() for p in net.parameters())
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
