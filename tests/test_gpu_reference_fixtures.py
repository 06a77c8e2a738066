"""The HIP path against the two fixtures that hold outputs of the REFERENCE's own modules for whole-network runs
("What's missing" item 3): until now they met the HIP path only through the oracle.

  tests/golden/train_steps.json   three iterations of the loop body train.py:155-172 around the reference's resnet50(S=7)
                                  and YOLOLossV1, N=2, 448x448, weights ob.init_params(seed 0), SGD momentum 0.99
                                  (oracle/gen_golden.py:gen_train_steps)
  tests/golden/wholenet_fwd.npz   training-mode forward of the reference's resnet50 / densenet121 (S=7 and 14), N=2,
                                  128x128, weights ob.init_params(seed S) (oracle/gen_golden.py:gen_wholenet)

Both fixtures are fp32 runs on *unscaled random-init* weights with a batch of TWO: BatchNorm then normalises over as few as
8 values (bn_end of the 2x2 grid), and one bf16 rounding anywhere upstream moves a sigmoid output by tenths.  What bf16
storage alone costs on exactly these inputs is measured with the oracle in its bf16-storage mode (``q=ob.bf16_ste``; the
same oracle in fp32 mode reproduces the fixtures to 1e-3 / 1e-4, tests/test_oracle_golden.py):

    train_steps.json   loss of the bf16-storage oracle vs the fixture, steps 1/2/3: 2.5 / 7.5 / 6.7 % (warm-up run) and
                       2.5 / 1.6 / 1.6 % in the build container (8 threads); 2.6 / 0.7 / 1.2 % and 2.6 / 5.1 / 9.0 % on the GPU
                       box's host (16 threads) -- the SAME CPU code on two machines differs by up to 8 % from step 2 on:
                       with N=2 the trajectory is chaotic under any change of fp32 summation order, let alone bf16 storage
    wholenet_fwd.npz   sigmoid outputs, mean |err| (max): resnet S=7 0.19 (0.65), S=14 0.12 (0.54), densenet 0.055 / 0.022

Stated tolerances (measured HIP values in brackets):
  * train steps: learning rates exact; loss within 8 % of the reference fixture at step 1 (no update has happened yet:
    [4.8 %]) and within 12 % at steps 2 and 3 [2.5 %, 8.2 %; 4.0 %, 1.0 %]; the two large component sums within 15 %.
  * whole-net forward, three-way like the fp8 end-to-end test: the HIP path must sit (a) no further from the reference
    fixture than 1.5x the bf16-storage oracle does (+1e-2) [0.186 vs 0.192, 0.118 vs 0.115, 0.051 vs 0.053, 0.022 vs 0.022]
    and (b) no further from that oracle than the oracle is from the fixture (+1e-2); bn_end's running mean likewise.
Kernel exactness is carried by the per-layer tests (test_gpu_bench_configs.py 1e-2 / 2e-3, test_gpu_blocks_golden.py);
this file closes the loop to the reference's own whole-network numbers.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag,epoch,lr_map", [("warmup", 0, {}), ("epoch1", 1, {1: 0.001})])
def test_hip_train_step_on_the_reference_train_steps_fixture(tag, epoch, lr_map):
    """train.py:155-172 x3 on the HIP path: LR policy, backbone, fused loss, zero_grad, backward, fused SGD."""
    from oracle import backbones as ob
    from oracle import train_step as ots
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.train import learning_rate_policy, train_step
    from yolo_v1_amd.v1Loss import YOLOLossV1
    ref = json.load(open(os.path.join(GOLDEN, "train_steps.json")))[tag]
    images, target = ots.synthetic_batch(2, 7)
    P = ots.make_state("resnet", 7, seed=0)
    net = resnet50(S=7)
    net.load_state_dict({k: v.detach().clone() for k, v in P.items()}, strict=True)
    net = net.to(DEV).train()
    opt = FusedSGD(net.parameters(), lr=0.0, momentum=0.99)
    crit = YOLOLossV1(2, 7, 2, 20, 5.0, 0.5, _quiet=True)
    xd, td = images.to(DEV), target.to(DEV)
    lr, got = 0.0, []
    for it in range(1, 4):
        lr = learning_rate_policy(it, epoch, lr, lr_map)
        loss = train_step(net, crit, opt, xd, td, lr)
        got.append({"loss": float(loss.item()), "comps": [float(c) for c in crit.last_components.tolist()], "lr": lr})
    emu = ots.train_steps(P, images, target, 7, 3, "resnet", epoch=epoch, lr_map=lr_map, fwd_kwargs={"q": ob.bf16_ste})
    for k, (g, e, r) in enumerate(zip(got, emu, ref)):
        print("%s step %d: HIP loss %.4f | reference fixture %.4f | bf16-storage oracle on this host %.4f | lr %g" % (
            tag, k + 1, g["loss"], r["loss"], e["loss"], g["lr"]))
        assert abs(g["lr"] - r["lr"]) < 1e-15
        assert np.isfinite(g["loss"])
        assert abs(g["loss"] - r["loss"]) <= (8e-2 if k == 0 else 1.2e-1) * r["loss"], (tag, k, g["loss"], r["loss"], e["loss"])
        # the four raw component sums (location, contain, not-contain, classify): the two large ones within 15 %
        for j in (2, 3):
            assert abs(g["comps"][j] - r["comps"][j]) <= 0.15 * r["comps"][j], (tag, k, j, g["comps"], r["comps"])
    # same first step in both runs of the fixture (the LR only acts on the update): the HIP path is deterministic
    if tag == "epoch1":
        assert got[1]["loss"] != got[0]["loss"]


@pytest.mark.parametrize("kind,S", [("resnet", 7), ("resnet", 14), ("densenet", 7), ("densenet", 14)])
def test_hip_forward_on_the_reference_wholenet_fixture(kind, S):
    """OriginResNet.py:173-195 / OriginDenseNet.py:114-129 in training mode on the fixture's inputs."""
    from oracle import backbones as ob
    z = np.load(os.path.join(GOLDEN, "wholenet_fwd.npz"))
    want = torch.from_numpy(z["%s_S%d_y" % (kind, S)])
    want_rm = z["%s_S%d_bn_end_rm" % (kind, S)]
    if kind == "resnet":
        from yolo_v1_amd.backbones.OriginResNet import resnet50 as ctor
        shapes, fwd = ob.resnet50_param_shapes(S), ob.resnet50_forward
    else:
        from yolo_v1_amd.backbones.OriginDenseNet import densenet121 as ctor
        shapes, fwd = ob.densenet121_param_shapes(S), ob.densenet121_forward
    P = ob.init_params(shapes, kind, seed=S)
    x = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(100 + S))
    net = ctor(S=S)
    net.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    net = net.to(DEV).train()
    Pe = {k: v.clone() for k, v in P.items()}
    with torch.no_grad():
        got = net(x.to(DEV)).cpu()
        emu = fwd(x, Pe, S, training=True, q=ob.bf16_ste)          # updates Pe's running statistics in place
    assert tuple(got.shape) == tuple(want.shape) and bool(torch.isfinite(got).all())
    d_ref, d_emu, floor = (got - want).abs(), (got - emu).abs(), (emu - want).abs()
    print("\n%s S=%d: HIP vs reference fixture mean %.3g max %.3g | bf16-storage oracle vs fixture mean %.3g max %.3g | "
          "HIP vs that oracle mean %.3g max %.3g" % (kind, S, d_ref.mean(), d_ref.max(), floor.mean(), floor.max(),
                                                     d_emu.mean(), d_emu.max()))
    assert float(d_ref.mean()) <= 1.5 * float(floor.mean()) + 1e-2
    assert float(d_emu.mean()) <= float(floor.mean()) + 1e-2
    # bn_end's running mean after one training-mode forward (momentum 0.1): same three-way criterion
    rm = net.state_dict()["bn_end.running_mean"].cpu().numpy()
    rm_emu = Pe["bn_end.running_mean"].numpy()
    scale = float(np.abs(want_rm).max())
    f_rm = float(np.abs(rm_emu - want_rm).max())
    print("bn_end.running_mean: HIP vs fixture max %.3g | bf16-storage oracle vs fixture max %.3g | scale %.3g" % (
        float(np.abs(rm - want_rm).max()), f_rm, scale))
    assert float(np.abs(rm - want_rm).max()) <= 1.5 * f_rm + 5e-2 * scale
    assert float(np.abs(rm - rm_emu).max()) <= 1.5 * f_rm + 5e-2 * scale
    assert int(net.state_dict()["bn_end.num_batches_tracked"]) == 1
