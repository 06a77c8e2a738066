"""Lifetime of captured training steps: the hipGraphLaunch segfault of round 2, diagnosed.

What was found (tools/graph_accumulate_probe.py on MI355X, native backtrace under rocgdb):
  * NOT an accumulation limit: 1024 forked hipGraphExec objects alive in one process replay fine.
  * The fault is in the HIP runtime torch ships (libamdhip64.so of ROCm 7.0 in torch/lib): after a hipGraphExec whose graph
    the runtime spreads over FOUR queues (main chain + three parallel branches) has been DESTROYED, launching a newly
    instantiated exec faults in amd::NDRangeKernelCommand::AllocCaptureSetValidate <- hip::GraphKernelNode::CreateCommand
    <- hip::Graph::RunNodes <- hipGraphLaunch, within 16 build/replay/destroy cycles.  Two-queue graphs (main chain + one
    side branch) and single-queue graphs survive 512 such cycles; so do four-queue graphs when destruction is deferred.
  * Round 2's suite hit it because every test's captured step was destroyed when the test returned, while the backward's
    graph still hopped over four queues; it "went away" when the main chain was pinned to one queue (two-queue graphs).
  * Round 3, later: two queues are necessary, not sufficient.  With the bn3 algebra in the step (more work on the side
    branch) the full suite died 4 times out of 7 -- glibc abort in free() / SIGSEGV in a TensorImpl destructor of an unrelated
    tensor, each time shortly after a captured full-size step had been destroyed: host-heap corruption, not a fault in the
    launch.  The graphs' shape is the same with and without the algebra (tools/graph_dot_width.py on the runtime's own DOT
    dump: DAG width 2, two runtime stream ids, 63 forks, one join).  It never happened with the execs parked instead of
    destroyed, without the side stream, or with the algebra off (5 runs of the same reproducer).
The product therefore (a) keeps every captured step at two queues (ops.SideStream refuses a capture order that would move
the main chain off its queue, and there is ONE side stream), (b) retires graphs deterministically (train.GraphedStep.close)
instead of whenever the collector gets to them, and (c) destroys a retired exec only after GraphedStep.PARK (8) newer
ones have been retired behind it -- the deferral that made four-queue graphs survive 512 cycles in the probe.  This test
runs the diagnosed pattern on the REAL training step: capture, replay, close, 20 times in one process (so that execs ARE
destroyed, twelve of them), then once more at a different network.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_capture_replay_close_cycles_of_the_training_step():
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.train import GraphedStep
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    from yolo_v1_amd.v1Loss import YOLOLossV1
    images, target = synthetic_batch(4, 2, hw=128, device=DEV)
    torch.manual_seed(0)
    net = resnet50(S=7).to(DEV).train()
    opt = FusedSGD(net.parameters(), lr=0.0, momentum=0.99)
    base = GraphedStep.live_graphs()
    first = None
    destroyed = 0
    for cycle in range(20):
        before = GraphedStep.parked_graphs()
        with GraphedStep(net, YOLOLossV1(4, 2, 2, 20, _quiet=True), opt, images, target, warmup=1, preserve_state=True) as gs:
            assert GraphedStep.live_graphs() == base + 1
            losses = [float(gs(0.0).item()) for _ in range(2)]         # lr 0: every cycle replays the same step
        assert GraphedStep.live_graphs() == base
        assert gs.graph is None and gs.net is None
        assert GraphedStep.parked_graphs() <= GraphedStep.PARK           # retired execs wait in a bounded queue ...
        destroyed += before + 1 - GraphedStep.parked_graphs()            # ... and the oldest are destroyed as it overflows
        first = first or losses
        assert losses == first, (cycle, losses, first)                  # bitwise the same step from a fresh capture each time
    # and a different executor right after the last destroyed exec
    dn = densenet121(S=7).to(DEV).train()
    od = FusedSGD(dn.parameters(), lr=0.0, momentum=0.99)
    with GraphedStep(dn, YOLOLossV1(4, 2, 2, 20, _quiet=True), od, images, target, warmup=1) as gd:
        a, b = float(gd(1e-4).item()), float(gd(1e-4).item())
    assert a == a and b == b and a != b
    assert GraphedStep.live_graphs() == base
    assert destroyed >= 20 - GraphedStep.PARK
    GraphedStep.drain()                                                  # a process that wants its memory back
    assert GraphedStep.parked_graphs() == 0
    x = torch.ones(1 << 20, device=DEV)                                  # and the device is still usable afterwards
    assert float((x * 2).sum().item()) == float(2 << 20)


def test_a_closed_step_refuses_to_run():
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.optim import FusedSGD
    from yolo_v1_amd.train import GraphedStep
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    from yolo_v1_amd.v1Loss import YOLOLossV1
    images, target = synthetic_batch(2, 2, hw=128, device=DEV)
    net = resnet50(S=7).to(DEV).train()
    gs = GraphedStep(net, YOLOLossV1(2, 2, 2, 20, _quiet=True), FusedSGD(net.parameters(), lr=0.0, momentum=0.99), images, target,
                     warmup=1)
    gs(1e-4)
    gs.close()
    with pytest.raises(RuntimeError):
        gs(1e-4)
