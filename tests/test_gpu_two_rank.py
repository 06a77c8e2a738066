"""Two ranks of the data-parallel training step on ONE MI355X (VERDICT r1 row e / item 2).

The driver measures N = 2, 4, 8 on a node this build never sees; RCCL refuses two ranks on one device, so here the two
processes share cuda:0 and exchange gradients over gloo (which moves CUDA tensors through the host).  Everything else
is the production path: `sync_replicas` (rank 1 starts from different weights), `GradSync`, the two-hipGraph
`GraphedStep` with the collective of the deep stages issued between the graphs, the fused SGD -- and, separately, the
eager `train_step` with buckets issued from inside the backward executor.

Checked against a single-process restatement of what data parallelism means for the reference (nn.DataParallel,
train.py:34,:80: per-replica BatchNorm statistics, gradients averaged): two replicas stepped in one process, each on
its rank's batch, with the parameter gradients averaged by hand before both optimizer steps.  gloo's host sum of two
fp32 values and the 0.5 scale are exact, so the comparison is bit-for-bit.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LRS = [1e-3, 2e-3, 5e-4]
KEYS = ("conv1.weight", "bn1.weight", "layer1.0.conv1.weight", "layer2.1.conv2.weight", "layer4.0.downsample.0.weight",
        "layer5.2.bn3.bias", "layer6.weight", "bn_end.bias", "layer3.2.bn2.running_mean", "bn1.running_var")


def _data(rank):
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    return synthetic_batch(4, 2, hw=128, seed=1234 + rank, device=DEV)


def _pick(net):
    sd = net.state_dict()
    return {k: sd[k].detach().float().cpu().numpy().copy() for k in KEYS}


def _worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from yolo_v1_amd import distributed as ydist
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.train import GraphedStep, sync_replicas, train_step
    from yolo_v1_amd.v1Loss import YOLOLossV1
    r, w, device = ydist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and device.type == "cuda"
    torch.manual_seed(100 + rank)                       # replicas start apart, as separate processes do
    net = resnet50(S=7).to(device).train()
    sync_replicas(net)
    images, target = _data(rank)
    opt = FusedSGD(net.parameters(), lr=0.0, momentum=0.99)
    loss_layer = YOLOLossV1(4, 2, 2, 20, _quiet=True)
    sync = ydist.GradSync(net)
    if mode == "graphed":
        gs = GraphedStep(net, loss_layer, opt, images, target, sync, warmup=1, preserve_state=True)
        assert gs.two_phase and not gs.in_graph_step
        losses = [float(gs(lr).item()) for lr in LRS]
    else:
        losses = [float(train_step(net, loss_layer, opt, images, target, lr, sync).item()) for lr in LRS]
    torch.cuda.synchronize()
    if mode == "graphed":
        # every collective of the graphed path ran in place on a contiguous range of the gradient arena
        assert sync.in_place_buckets == sync.buckets_issued and sync.buckets_issued > 0, (sync.in_place_buckets, sync.buckets_issued)
        assert all(p.grad.untyped_storage().data_ptr() == gs.arena.flat.untyped_storage().data_ptr() for p in net.parameters())
    q.put((rank, losses, _pick(net), sync.buckets_issued))
    if mode == "graphed":
        gs.close()
    dist.destroy_process_group()


def _run(mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() + (7 if mode == "graphed" else 0)) % 400
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _reference():
    """Both replicas in this process: per-replica forward/backward, hand-averaged gradients, two optimizer steps."""
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.v1Loss import YOLOLossV1
    torch.manual_seed(100)                               # rank 0's initial weights, broadcast to rank 1
    nets = [resnet50(S=7).to(DEV).train()]
    torch.manual_seed(101)
    nets.append(resnet50(S=7).to(DEV).train())
    nets[1].load_state_dict(nets[0].state_dict())
    from yolo_v1_amd import ops
    ops.bump_weight_epoch()
    opts = [FusedSGD(n.parameters(), lr=0.0, momentum=0.99) for n in nets]
    crit = [YOLOLossV1(4, 2, 2, 20, _quiet=True) for _ in nets]
    data = [_data(0), _data(1)]
    losses = [[], []]
    for lr in LRS:
        for r in range(2):
            for g in opts[r].param_groups:
                g['lr'] = lr
            loss = crit[r](nets[r](data[r][0]), data[r][1])
            opts[r].zero_grad()
            loss.backward()
            losses[r].append(float(loss.item()))
        with torch.no_grad():
            for pa, pb in zip(nets[0].parameters(), nets[1].parameters()):
                avg = (pa.grad + pb.grad) * 0.5
                pa.grad.copy_(avg)
                pb.grad.copy_(avg)
        for o in opts:
            o.step()
    torch.cuda.synchronize()
    return losses, [_pick(n) for n in nets]


@pytest.mark.parametrize("mode", ["graphed", "eager"])
def test_two_ranks_on_one_gpu_match_hand_averaged_replicas(mode):
    (r0, l0, p0, b0), (r1, l1, p1, b1) = _run(mode)
    ref_losses, ref_params = _reference()
    assert l0 == ref_losses[0] and l1 == ref_losses[1], (l0, ref_losses[0], l1, ref_losses[1])
    for k in KEYS:
        if "running" in k:                               # BatchNorm buffers stay per rank (what DataParallel does)
            np.testing.assert_array_equal(p0[k], ref_params[0][k], err_msg=k)
            np.testing.assert_array_equal(p1[k], ref_params[1][k], err_msg=k)
        else:                                            # parameters: identical on both ranks and equal to the reference
            np.testing.assert_array_equal(p0[k], p1[k], err_msg=k)
            np.testing.assert_array_equal(p0[k], ref_params[0][k], err_msg=k)
    assert not np.array_equal(p0["bn1.running_var"], p1["bn1.running_var"])      # the ranks did see different batches
    assert b0 == b1 and b0 >= len(LRS)
