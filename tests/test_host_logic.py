"""CPU: host-side logic of the product (target encoder, VOC AP, LR policy, state_dict surface,
gradient-sync bucketing over gloo with world_size 2) against the golden vectors / the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_cases


def test_product_encoder_matches_reference_fixture():
    from yolo_v1_amd.utils.YOLODataLoader import encode_boxes, yoloDataset
    for c in load_cases("encoder_cases.npz"):
        got = encode_boxes(torch.tensor(c["boxes"]).reshape(-1, 4), torch.tensor(c["labels"]), int(c["S"]))
        np.testing.assert_array_equal(got.numpy(), c["target"])
    ds = yoloDataset(None, S=14, length=3, with_file_path=True)
    img, tgt, fname = ds[1]
    assert tuple(img.shape) == (3, 448, 448) and tuple(tgt.shape) == (14, 14, 30) and fname.endswith(".jpg")
    assert 1 <= int((tgt[..., 0] == 1).sum()) <= 3 and len(ds) == 3


def test_product_voc_eval_known_answer():
    from yolo_v1_amd.utils.utils import voc_ap, voc_eval
    k = json.load(open(os.path.join(GOLDEN, "voc_eval_known.json")))
    target = lambda: {tuple(s.split("|")): [list(b) for b in v] for s, v in k["target"].items()}

    class Q:
        def info(self, m):
            pass
    assert abs(voc_eval(k["preds"], target(), VOC_CLASSES=k["classes"], logger=Q()) - 0.9166666666666666) < 1e-12
    assert abs(voc_eval(k["preds"], target(), VOC_CLASSES=k["classes_quirk"], logger=Q()) - k["mAP_quirk"]) < 1e-12
    assert abs(voc_ap(np.array(k["rec"]), np.array(k["prec"])) - k["ap_area"]) < 1e-12
    assert abs(voc_ap(np.array(k["rec"]), np.array(k["prec"]), True) - k["ap_07"]) < 1e-12


def test_product_lr_policy_and_state_dict_surface():
    from yolo_v1_amd.train import learning_rate_policy
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    lr = 0.0
    for it in range(1, 1200):
        lr = learning_rate_policy(it, 0, lr, {1: 0.001})
    assert abs(lr - 1e-3) < 1e-12
    assert learning_rate_policy(7, 75, 0.3, {75: 1e-4}) == 1e-4
    inv = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["inventory"]
    net = resnet50(S=7)
    assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == inv["resnet_S7"]
    assert sum(p.numel() for p in net.parameters()) == 41155708
    with pytest.raises(Exception):
        net(torch.zeros(1, 3, 64, 64))          # CPU input: the HIP path refuses, no fallback


def test_densenet_state_dict_surface():
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121
    inv = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["inventory"]
    for S in (7, 14):
        net = densenet121(S=S)
        assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == inv["densenet_S%d" % S]


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yolo_v1_amd.distributed import GradSync
    torch.manual_seed(rank)
    params = [torch.nn.Parameter(torch.zeros(8, 4, 3, 3).contiguous(memory_format=torch.channels_last)),
              torch.nn.Parameter(torch.zeros(16)), torch.nn.Parameter(torch.zeros(5, 7))]
    sync = GradSync(None, bucket_mb=0.0005)           # tiny buckets: several flushes
    grads = []
    for p in params:                                   # what the backward executor hands over
        g = torch.randn(p.shape).contiguous(memory_format=torch.channels_last) if p.dim() == 4 else torch.randn(p.shape)
        grads.append(g)
        p.grad = g.clone()
        sync.on_ready([(p, g)])
    sync.finish()
    first = [p.grad.clone() for p in params]
    # the sequence of the two-graph data-parallel step: early gradients start their collective, the rest follow in
    # reduce_all, which also waits for the early one
    for p, g in zip(params, grads):
        p.grad = g.clone()
    sync2 = GradSync(None)
    sync2.start([(params[0], params[0].grad)])
    sync2.reduce_all([(p, p.grad) for p in params[1:]])
    second = [p.grad.clone() for p in params]
    tonp = lambda ts: [t.detach().contiguous().numpy().copy() for t in ts]      # numpy, not tensors: a tensor in an mp.Queue
    q.put((rank, tonp(grads), tonp(first), sync.buckets_issued, tonp(second), sync2.buckets_issued))   # needs the sender alive
    dist.destroy_process_group()


def test_gradsync_gloo_world2_averages_into_param_grad():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29400 + os.getpid() % 500
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = [(r, [torch.from_numpy(v) for v in g], [torch.from_numpy(v) for v in o], nb, [torch.from_numpy(v) for v in s_], nc)
           for (r, g, o, nb, s_, nc) in res]
    (_, g0, out0, nb0, sec0, nc0), (_, g1, out1, nb1, sec1, nc1) = res
    assert nb0 >= 2 and nb0 == nb1 and nc0 == nc1 == 2
    for a, b, o0, o1, s0, s1 in zip(g0, g1, out0, out1, sec0, sec1):
        want = (a + b) / 2
        torch.testing.assert_close(o0, want)
        torch.testing.assert_close(o1, want)
        torch.testing.assert_close(s0, want)
        torch.testing.assert_close(s1, want)


def test_checkpoint_roundtrip_and_pretrained_by_name(tmp_path):
    """N4: reference-compatible files (train.py:59-78 by-name ImageNet init, :207-209 'module.' prefix, eval.py:63-68)."""
    import torch
    from yolo_v1_amd import checkpoint
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    torch.manual_seed(3)
    net = resnet50(S=14)
    ref = {k: v.clone() for k, v in net.state_dict().items()}
    # a torchvision-shaped ImageNet file: backbone keys + the classifier the YOLO net does not have
    g = torch.Generator().manual_seed(5)
    pre = {k: torch.randn(v.shape, generator=g) for k, v in ref.items()
           if k.startswith(('conv1', 'bn1', 'layer1', 'layer2')) and v.dtype.is_floating_point}
    pre['fc.weight'] = torch.randn(1000, 2048, generator=g)
    pre['fc.bias'] = torch.randn(1000, generator=g)
    taken = checkpoint.init_from_pretrained(net, pre)
    assert set(taken) == set(pre) - {'fc.weight', 'fc.bias'}
    sd = net.state_dict()
    for k in ref:
        assert torch.equal(sd[k], pre[k] if k in taken else ref[k]), k
    bad = dict(pre)
    bad['conv1.weight'] = torch.zeros(64, 3, 3, 3)
    with pytest.raises(ValueError):
        checkpoint.init_from_pretrained(net, bad)

    path = str(tmp_path / "resnet_sgd_S14_yolo.pth")
    checkpoint.save(net, path)
    on_disk = torch.load(path, weights_only=True)
    assert all(k.startswith('module.') for k in on_disk) and len(on_disk) == len(sd)
    assert on_disk['module.layer1.0.conv2.weight'].shape == (64, 64, 3, 3) and on_disk['module.conv1.weight'].is_contiguous()
    net2 = resnet50(S=14)
    checkpoint.load(net2, path)
    for k, v in net2.state_dict().items():
        assert torch.equal(v, sd[k]), k
    plain = str(tmp_path / "plain.pth")
    checkpoint.save(net, plain, data_parallel_prefix=False)
    net3 = resnet50(S=14)
    checkpoint.load(net3, plain)
    assert torch.equal(net3.state_dict()['layer6.weight'], sd['layer6.weight'])


def test_collate_raw_pads_boxes_to_batch_max():
    from yolo_v1_amd.utils.YOLODataLoader import collate_raw, yoloDataset
    ds = yoloDataset(None, S=7, length=4, objs=3, raw_targets=True, image_size=32)
    samples = [ds[0], ds[1], (ds[2][0], torch.zeros(0, 4), torch.zeros(0, dtype=torch.long))]
    imgs, boxes, labels, counts = collate_raw(samples)
    assert tuple(imgs.shape) == (3, 3, 32, 32) and tuple(boxes.shape) == (3, 3, 4) and tuple(labels.shape) == (3, 3)
    assert counts.tolist() == [3, 3, 0] and counts.dtype == torch.int32 and labels.dtype == torch.int64
    assert torch.equal(boxes[1], ds[1][1]) and float(boxes[2].abs().sum()) == 0.0


def test_list_file_dataset_with_pil_loader(tmp_path):
    """Darknet-style list file (utils/YOLODataLoader.py:94-106): image decoded + resized + normalised by the PIL loader,
    labels read from the sibling ``labels`` directory and encoded."""
    from PIL import Image
    from yolo_v1_amd.utils.YOLODataLoader import collate_raw, encode_boxes, pil_image_loader, yoloDataset
    (tmp_path / "JPEGImages").mkdir()
    (tmp_path / "labels").mkdir()
    rng = np.random.default_rng(0)
    lines = []
    for i, (w, h) in enumerate([(500, 375), (320, 480)]):
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(tmp_path / "JPEGImages" / ("%06d.jpg" % i))
        (tmp_path / "labels" / ("%06d.txt" % i)).write_text("11 0.5 0.4 0.3 0.2\n3 0.1 0.9 0.05 0.1\n" if i == 0 else "7 0.7 0.2 0.5 0.3\n")
        lines.append(str(tmp_path / "JPEGImages" / ("%06d.jpg" % i)))
    lst = tmp_path / "train.txt"
    lst.write_text("\n".join(lines) + "\n")
    ds = yoloDataset(str(lst), train=False, with_file_path=True, S=7)
    img, tgt, fname = ds[0]
    assert tuple(img.shape) == (3, 448, 448) and img.dtype == torch.float32 and fname == lines[0]
    assert -2.2 < float(img.min()) and float(img.max()) < 2.7            # (x - mean) / std of values in [0, 1]
    want = encode_boxes(torch.tensor([[0.5, 0.4, 0.3, 0.2], [0.1, 0.9, 0.05, 0.1]]), torch.tensor([11, 3]), 7)
    assert torch.equal(tgt, want)
    # BGR order like cv2.imread: channel 0 of the tensor is the file's blue channel
    rgb = pil_image_loader(lines[0], bgr=False)
    b0 = (img[0] * 0.229 + 0.485)
    r2 = (rgb[2] * 0.225 + 0.406)
    assert torch.allclose(b0, r2, atol=1e-5)
    raw = yoloDataset(str(lst), train=False, S=7, raw_targets=True)
    imgs, boxes, labels, counts = collate_raw([raw[0], raw[1]])
    assert counts.tolist() == [2, 1] and labels[1, 0].item() == 7 and tuple(imgs.shape) == (2, 3, 448, 448)


# ---------------------------------------------------------------------------------------------------------------
# world-size-2 gloo tests of the data-parallel HOST logic around the training step (ADVICE high):
# replicas made identical before the first step, and the train_step / GraphedStep sequencing (gradient-ready hooks from
# inside the backward executor, phase boundary, start -> reduce_all -> optimizer) on a stub backbone whose executors are
# plain torch-CPU code with the HipBackbone interface.
def _spawn2(target, *extra):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() * 7 + len(extra)) % 500
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + extra) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def _sync_replicas_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yolo_v1_amd import ops
    from yolo_v1_amd.backbones.OriginDenseNet import densenet121
    from yolo_v1_amd.train import sync_replicas
    torch.manual_seed(100 + rank)                     # what separate processes get: different initial weights
    net = densenet121(S=14)
    with torch.no_grad():
        for b in net.buffers():
            if b.dtype.is_floating_point:
                b.add_(float(rank))
            else:
                b.add_(rank * 3)
    import hashlib
    digest = lambda: {k: hashlib.sha1(v.contiguous().numpy().tobytes()).hexdigest() for k, v in net.state_dict().items()}
    before = digest()                                 # digests, not tensors: 700+ tensors through an mp.Queue stall
    epoch0 = ops._WEIGHT_EPOCH[0]
    sync_replicas(net)
    after = digest()
    q.put((rank, before, after, ops._WEIGHT_EPOCH[0] - epoch0))
    dist.destroy_process_group()


def test_sync_replicas_gloo_world2_makes_every_rank_equal_to_rank0():
    (_, b0, a0, e0), (_, b1, a1, e1) = _spawn2(_sync_replicas_worker)
    assert sum(b0[k] != b1[k] for k in b0) > 300                   # the ranks really started apart
    for k in b0:
        assert a0[k] == b0[k], k                                   # rank 0 unchanged
        assert a1[k] == b0[k], k                                   # rank 1 == rank 0, parameters and BatchNorm buffers
    assert e0 >= 1 and e1 >= 1                                     # bf16 weight shadows were invalidated


def _make_stub_net():
    import torch.nn as nn
    from yolo_v1_amd.engine import HipBackbone

    class StubFn(torch.autograd.Function):                        # engine.BackboneFn without the CUDA requirement
        @staticmethod
        def forward(ctx, net, x, *params):
            with torch.no_grad():
                pred, saved = net._run_forward(x, True, True)
            ctx.net, ctx.saved, ctx.params = net, saved, params
            return pred

        @staticmethod
        def backward(ctx, gpred):
            with torch.no_grad():
                grads = ctx.net._run_backward(ctx.saved, gpred)
            return (None, None) + tuple(grads.get(p) for p in ctx.params)

    class Stub(HipBackbone):
        """pred = (tanh(x @ w1) @ w_l4) @ w3: three 'layers'; layer4 is where the executor calls the phase boundary."""

        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(0)
            self.w1 = nn.Parameter(torch.randn(12, 16, generator=g) * 0.3)
            self.layer4 = nn.ParameterList([nn.Parameter(torch.randn(16, 16, generator=g) * 0.3)])
            self.w3 = nn.Parameter(torch.randn(16, 5, generator=g) * 0.3)
            self.emitted = []

        def forward(self, x):
            return StubFn.apply(self, x, *list(self.parameters()))

        def _run_forward(self, x, train, save):
            h1 = torch.tanh(x @ self.w1)
            h2 = h1 @ self.layer4[0]
            return h2 @ self.w3, (x, h1, h2)

        def _run_backward(self, rec, gpred):
            x, h1, h2 = rec
            grads = {self.w3: h2.t() @ gpred}
            self._emit(grads, [self.w3])
            gh2 = gpred @ self.w3.t()
            grads[self.layer4[0]] = h1.t() @ gh2
            self._emit(grads, [self.layer4[0]])
            if self._phase_boundary is not None:
                self._phase_boundary(grads)
            gh1 = (gh2 @ self.layer4[0].t()) * (1 - h1 * h1)
            grads[self.w1] = x.t() @ gh1
            self._emit(grads, [self.w1])
            return grads
    return Stub()


class _StubLoss:
    quiet = True

    def __call__(self, pred, target):
        return ((pred - target) ** 2).sum() / pred.shape[0]

    def loss_and_grad(self, pred, target):
        return ((pred - target) ** 2).sum() / pred.shape[0], 2.0 * (pred - target) / pred.shape[0]


def _train_seq_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yolo_v1_amd.distributed import GradSync
    from yolo_v1_amd.train import GraphedStep, sync_replicas, train_step
    g = torch.Generator().manual_seed(50 + rank)                  # each rank its own shard of the batch
    x, t = torch.randn(6, 12, generator=g), torch.randn(6, 5, generator=g)
    out = {}
    # (a) eager train_step: buckets issued from inside the backward executor through the gradient-ready hook
    net = _make_stub_net()
    with torch.no_grad():
        net.w1.add_(0.01 * rank)                                  # replicas start apart ...
    sync_replicas(net)                                            # ... and are made equal, as train.main does
    opt = torch.optim.SGD(net.parameters(), lr=0.0, momentum=0.99)
    sync = GradSync(net, bucket_mb=1e-4)                          # tiny buckets: one collective per layer, mid-backward
    losses = [float(train_step(net, _StubLoss(), opt, x, t, 0.05, sync)) for _ in range(3)]
    out["eager"] = ([p.detach().clone() for p in net.parameters()], losses, sync.buckets_issued)
    # (b) the GraphedStep multi-rank sequence (phase boundary -> start(early) -> rest -> optimizer), executors driven
    # directly; the two "replays" run the direct path eagerly (hipGraph capture itself needs the GPU)
    net2 = _make_stub_net()
    opt2 = torch.optim.SGD(net2.parameters(), lr=0.05, momentum=0.99)
    gs = GraphedStep.__new__(GraphedStep)
    gs.net, gs.loss_layer, gs.opt, gs.sync = net2, _StubLoss(), opt2, GradSync(None)
    gs.images, gs.target, gs.two_phase, gs.in_graph_step, gs.phase1, gs.steps_done, gs.arena = x, t, True, False, None, 0, None

    gs.phases, gs.graphs = [], []

    def replay1():
        gs.phases = []
        gs.loss = gs._direct(lambda grads: gs.phases.append(list(grads.items())))
    for _ in range(3):
        gs._dp_sequence([replay1, lambda: None])
    out["graphed"] = ([p.detach().clone() for p in net2.parameters()], [id(p) for p, _ in gs.phases[0]] ==
                      [id(net2.w3), id(net2.layer4[0])], gs.sync.buckets_issued)
    tonp = lambda ts: [v.numpy().copy() for v in ts]
    out["eager"] = (tonp(out["eager"][0]),) + out["eager"][1:]
    out["graphed"] = (tonp(out["graphed"][0]),) + out["graphed"][1:]
    q.put((rank, out, x.numpy().copy(), t.numpy().copy()))
    dist.destroy_process_group()


def test_train_step_and_graphed_sequence_gloo_world2_match_global_batch_sgd():
    (_, o0, x0, t0), (_, o1, x1, t1) = _spawn2(_train_seq_worker)
    x0, t0, x1, t1 = (torch.from_numpy(v) for v in (x0, t0, x1, t1))
    for o in (o0, o1):
        o["eager"] = ([torch.from_numpy(v) for v in o["eager"][0]],) + o["eager"][1:]
        o["graphed"] = ([torch.from_numpy(v) for v in o["graphed"][0]],) + o["graphed"][1:]
    # single-process reference: the same three SGD steps on the mean of the two ranks' losses (== averaged gradients)
    ref = _make_stub_net()
    opt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.99)
    crit = _StubLoss()
    for _ in range(3):
        opt.zero_grad()
        loss = 0.5 * (crit(ref._run_forward(x0, True, True)[0], t0) + crit(ref._run_forward(x1, True, True)[0], t1))
        w = list(ref.parameters())
        pred0, rec0 = ref._run_forward(x0, True, True)
        pred1, rec1 = ref._run_forward(x1, True, True)
        g0 = ref._run_backward(rec0, crit.loss_and_grad(pred0, t0)[1])
        g1 = ref._run_backward(rec1, crit.loss_and_grad(pred1, t1)[1])
        for p in w:
            p.grad = 0.5 * (g0[p] + g1[p])
        opt.step()
    want = [p.detach() for p in ref.parameters()]
    for o in (o0, o1):
        for got, w in zip(o["eager"][0], want):
            torch.testing.assert_close(got, w, rtol=1e-5, atol=1e-6)
        for got, w in zip(o["graphed"][0], want):
            torch.testing.assert_close(got, w, rtol=1e-5, atol=1e-6)
        assert o["eager"][2] >= 3 * 3            # one bucket per layer and step, issued mid-backward
        assert o["graphed"][1]                   # the phase boundary handed over exactly the deep layers' gradients
        assert o["graphed"][2] == 3 * 2          # per step: one early collective + one for the rest
    for a, b in zip(o0["eager"][0], o1["eager"][0]):
        assert torch.equal(a, b)                 # replicas stay bit-identical
    for a, b in zip(o0["graphed"][0], o1["graphed"][0]):
        assert torch.equal(a, b)


def test_bench_self_launches_ranks_before_touching_the_gpu():
    """`python bench.py --gpus 2` must start its own ranks.  Without a GPU each rank stops at the
    'needs a GPU' check -- which proves both that two ranks were started under torch.distributed.run and that the
    parent got that far without any CUDA call -- and the parent relays the failure as its exit code."""
    import subprocess
    import sys
    from conftest import ROOT
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "launch with torch.distributed.run" not in r.stderr
    assert r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-2000:]


# ---------------------------------------------------------------------------------------------------------------
# rank-0 gate between epochs (only rank 0 validates; the others must not sit in a collective meanwhile)
def _gate_worker(rank, world, port, q, _tag=None):
    import time
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from yolo_v1_amd import distributed as ydist
    r, w, _ = ydist.init_from_env(backend="gloo", timeout_s=1234)
    assert (r, w) == (rank, world)
    t0 = time.perf_counter()
    if rank == 0:
        time.sleep(1.5)                               # "validation + checkpoint" on rank 0 only
    ydist.rank0_gate("epoch0", rank, world)
    waited = time.perf_counter() - t0
    ydist.rank0_gate("epoch1", rank, world) if rank == 0 else None      # a gate opened early is simply passed later
    if rank == 1:
        time.sleep(0.3)
        ydist.rank0_gate("epoch1", rank, world)
    t = torch.ones(1) * (rank + 1)
    dist.all_reduce(t)                                # the group still works afterwards
    q.put((rank, waited, float(t)))
    dist.destroy_process_group()


def test_rank0_gate_gloo_world2_holds_the_other_rank_without_a_collective():
    (_, w0, s0), (_, w1, s1) = _spawn2(_gate_worker, "gate")
    assert w1 >= 1.0, w1                              # rank 1 waited for rank 0's "validation"
    assert s0 == s1 == 3.0


def test_rank0_gate_is_a_no_op_for_one_rank():
    from yolo_v1_amd import distributed as ydist
    ydist.rank0_gate("anything", 0, 1)


def test_grad_arena_is_sized_from_padded_conv_requests():
    """conv_wgrad asks for Opad*taps*Ipad elements (output channels rounded up to 32): a network with several
    padded convolutions must fit its arena -- sized from the requests, no fixed slack."""
    import torch.nn as nn
    from yolo_v1_amd import ops
    from yolo_v1_amd.engine import ConvParam, HipBackbone

    class Net(HipBackbone):
        def __init__(self):
            super().__init__()
            self.stem = ConvParam(3, 64, 7, 2, 3)
            self.heads = nn.ModuleList([ConvParam(2048, 30, 1) for _ in range(40)])     # 40 x 2 padded rows x 2048
            self.wide = ConvParam(64, 33, 3, 1, 1)                                       # 33 -> 64 output rows
            self.bn = nn.BatchNorm2d(30)

    net = Net()
    arena = ops.GradArena(net, "cpu")
    for m in net.heads:
        assert arena.get(m.weight, 32 * 1 * 2048).numel() == 32 * 2048
    assert arena.get(net.wide.weight, 64 * 9 * 64).numel() == 64 * 9 * 64
    assert arena.get(net.stem.weight, 64 * 7 * 7 * 3).numel() == 64 * 147
    assert arena.get(net.bn.weight, 30).numel() == 30 and arena.get(net.bn.bias, 30).numel() == 30
    assert arena.top == arena.flat.numel()                                                # exactly what was requested
    with pytest.raises(Exception):
        arena.get(nn.Parameter(torch.zeros(8)), 8)
