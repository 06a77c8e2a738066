"""Training entry point with the reference's train.py surface (train.py:22-209), MI355X-native.

The reference's file is module-level code that no longer parses (unresolved merge conflicts at
:49-53, :136-141, :193-197); this module reproduces its behaviour as functions:
  * ``warmming_up_policy`` / ``learning_rate_policy``       train.py:22-32
  * ``build`` (backbone, loss layer, SGD momentum 0.99)     train.py:56-101
  * ``train_step`` = the loop body                          train.py:157-172
  * ``main`` = the epoch loop with its log line format      train.py:144-184
One process per GPU; with WORLD_SIZE > 1 the gradients are averaged over RCCL by
``yolo_v1_amd.distributed.GradSync`` while the backward is still running.
Evaluation (train.py:187-198) lives in ``yolo_v1_amd.eval``.
"""
import argparse
import collections
import os
import time
import weakref

import torch

from .v1Loss import YOLOLossV1

# defaults of the reference's module-level constants (train.py:34-57); the merge conflict at :49-53
# leaves two candidates for the last LR drop, 115 (HEAD) is taken
DEFAULTS = dict(learning_rate=0.0, num_epochs=200, batch_size=12, B=2, S=14, clsN=20, lbd_coord=5., lbd_no_obj=.5,
                lr_adjust_map={1: 0.001, 75: 0.0001, 115: 0.00001}, backbone='densenet')


def warmming_up_policy(now_iter, now_lr, stop_down_iter=1000):
    if now_iter <= stop_down_iter:
        now_lr += 0.000001
    return now_lr


def learning_rate_policy(now_iter, now_epoch, now_lr, lr_adjust_map, stop_down_iter=1000):
    now_lr = warmming_up_policy(now_iter, now_lr, stop_down_iter)
    if now_epoch in lr_adjust_map.keys():
        now_lr = lr_adjust_map[now_epoch]
    return now_lr


def build(backbone='resnet', S=7, B=2, clsN=20, batch_size=16, device='cuda:0', lbd_coord=5., lbd_no_obj=.5,
          logger=None, vis=None, quiet=False, with_sgd=True, fused_optimizer=False):
    """Backbone + loss layer + optimizer as train.py:56-101 sets them up (no pretrained download:
    that needs the network, train.py:61/:72)."""
    if backbone == 'resnet':
        from .backbones.OriginResNet import resnet50
        net = resnet50(S=S).to(device)
    else:
        from .backbones.OriginDenseNet import densenet121
        net = densenet121(S=S).to(device)
    net.train()
    if with_sgd:
        if fused_optimizer:
            from .optim import FusedSGD
            opt = FusedSGD(net.parameters(), lr=0.0, momentum=0.99)
        else:
            opt = torch.optim.SGD(net.parameters(), lr=0.0, momentum=0.99)          # train.py:84
    else:
        opt = torch.optim.Adam(net.parameters(), lr=0.0, weight_decay=1e-8)         # train.py:88
    loss_layer = YOLOLossV1(batch_size, S, B, clsN, lbd_coord, lbd_no_obj, _device=device, _logger=logger, _vis=vis,
                            _quiet=quiet)
    return net, loss_layer, opt


def train_step(net, loss_layer, optimizer, images, target, lr, grad_sync=None):
    """One iteration of train.py:158-172: set LR, forward, loss, zero_grad, backward, step.
    Returns the loss as a 0-dim device tensor (no host sync)."""
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr
    pred = net(images)
    loss = loss_layer(pred, target)
    optimizer.zero_grad()
    loss.backward()
    if grad_sync is not None:
        grad_sync.finish()
    optimizer.step()
    return loss


class GraphedStep:
    """The loop body of train.py:158-172 captured once into hipGraphs and replayed.

    A training step is ~800 short kernel launches; issued one by one from Python the GPU idles ~20 %
    of the step waiting for the host.  Captured, one ``hipGraphLaunch`` replays them all.  What makes the
    step capturable: static input buffers (copy each batch into ``.images`` / ``.target``), the
    learning rate in device memory (``FusedSGD``), no host sync inside (``_quiet`` loss layer).

    One rank: a single graph holds forward + loss + backward + optimizer.
    Several ranks: no collective is captured.  The step is two (optionally three) graphs split inside the backward pass where
    parameter-heavy stages are finished (``HipBackbone.set_phase_boundary``: ResNet after layer4 and after layer3):
    after replaying a graph, the RCCL all-reduce of the gradients it finished (79 %, then 17 % of ResNet-50's bytes)
    is issued asynchronously and runs on RCCL's stream beside the replay of the next graph; the last 4 % (layer2,
    layer1, stem: 6 MB) follow in one more collective, then the fused optimizer step.  Gradients live in one flat arena
    (``ops.GradArena``), so every collective runs in place on a contiguous range.  Executors without a phase
    boundary (DenseNet-121: 38 MB of gradients) use one graph and one collective.  ``YV1_DP_PHASES`` = 1 / 2 / 3 graphs.

    Lifetime: a captured step owns hipGraphExec objects, and the HIP runtime torch ships gives EVERY exec whose graph forks
    (the weight-gradient side stream) parallel streams of its own, plus a kernel-argument pool and the graph's private
    memory pool (all saved activations: ~23 GB for ResNet-50 at batch 64).  None of that is returned before the exec is
    destroyed, and Python only destroys it when the last reference goes -- which for an object reachable from a frame,
    a closure or an autograd graph means "at some later garbage collection".  ``close()`` (or ``with GraphedStep(...) as
    step:``) retires them deterministically (destruction itself is deferred by ``PARK`` retirements, see close()); a process
    that builds one captured step after another (bench.py's ``other_configs``, a train-then-evaluate script, the test suite)
    must call it.  ``GraphedStep.live_graphs()`` counts the
    execs that are still alive in this process (DESIGN.md section 4 has the fault this was found through).
    """
    _LIVE = weakref.WeakSet()
    # closed steps' execs, oldest first: destroyed only once PARK newer ones have been closed after them (see close())
    _PARKED = collections.deque()
    PARK = int(os.environ.get("YV1_GRAPH_PARK", "8"))

    @classmethod
    def parked_graphs(cls):
        return len(cls._PARKED)

    @classmethod
    def drain(cls, keep=0):
        """Destroys parked execs (oldest first) until ``keep`` are left; the device is synchronised first."""
        if len(cls._PARKED) > keep and torch.cuda.is_available():
            torch.cuda.synchronize()
        while len(cls._PARKED) > keep:
            cls._PARKED.popleft().reset()

    @classmethod
    def live_graphs(cls):
        """hipGraphExec objects owned by GraphedStep instances that have not been closed / collected yet."""
        return sum(len(g._all_graphs()) for g in list(cls._LIVE))

    def _all_graphs(self):
        gs = list(getattr(self, "graphs", None) or [])
        g = getattr(self, "graph", None)
        if g is not None and not any(g is x for x in gs):
            gs.append(g)
        return gs

    def close(self):
        """Retires the hipGraphExec objects and drops every reference this object holds -- the static buffers, the network,
        the gradient arena.  Idempotent; the object is unusable afterwards.

        The execs are not destroyed on the spot: they go to the back of a queue and ``CUDAGraph.reset()`` (the runtime's
        per-exec streams, argument pools and the graph's private memory pool) runs on the OLDEST one once ``PARK`` (8,
        ``YV1_GRAPH_PARK``) newer execs have been retired after it -- deterministic, but late.  Reason (DESIGN.md section
        4): with the HIP runtime torch 2.10+rocm7.0 ships, destroying an exec that has a side branch and then building /
        launching the next one corrupts the host heap now and then (round 2: SIGSEGV in hipGraphLaunch; round 3: glibc
        abort / SIGSEGV in unrelated tensor destructors, 4 of 7 suite runs); it never did with destruction deferred
        (tools/graph_accumulate_probe.py --defer 8: 512 cycles; the suite with parked execs), without the side stream, or
        in a process that destroys nothing.  ``GraphedStep.drain()`` empties the queue (end of a process that wants its
        memory back: a parked exec keeps its private pool, ~23 GB for ResNet-50 at batch 64)."""
        graphs = self._all_graphs()
        if graphs and torch.cuda.is_available():
            torch.cuda.synchronize()
        GraphedStep._PARKED.extend(graphs)
        GraphedStep.drain(keep=max(0, GraphedStep.PARK))
        self.graphs, self.graph, self.graph2 = [], None, None
        self.loss = None
        self.phases, self.phase1 = [], None
        if getattr(self, "sync", None) is not None and getattr(self.sync, "arena", None) is self.arena:
            self.sync.arena = None
        self.arena = None
        self.net = self.loss_layer = self.opt = self.sync = None
        self.images = self.target = None
        GraphedStep._LIVE.discard(self)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __init__(self, net, loss_layer, optimizer, images, target, grad_sync=None, warmup=3, two_phase=None,
                 preserve_state=False):
        """``preserve_state``: the eager warm-up steps that precede the capture (lazy initialisation, allocator warm-up)
        are real training steps; with this flag the parameters, BatchNorm buffers and momentum buffers are put back
        afterwards, so the first replay IS iteration 1 of the run (``train.main``); bench.py counts them as warm-up."""
        from . import ops
        from .optim import FusedSGD
        if not isinstance(optimizer, FusedSGD):
            raise TypeError("GraphedStep needs yolo_v1_amd.optim.FusedSGD (device-side learning rate)")
        self.net, self.loss_layer, self.opt, self.sync = net, loss_layer, optimizer, grad_sync
        self.images, self.target = images, target
        self.graph = self.graph2 = self.loss = None
        GraphedStep._LIVE.add(self)
        self.loss_layer.quiet = True
        self.in_graph_step = grad_sync is None
        # measured with a 1-rank RCCL group: a third graph (boundary after layer3) costs 0.4 ms per step -- the boundary
        # joins the weight-gradient stream, which lags the main stream by milliseconds that deep into the backward --
        # about what overlapping another 28 MB of all-reduce is estimated to save on 8 GPUs: two graphs by default
        nph = int(os.environ.get("YV1_DP_PHASES", "2"))
        if two_phase is None:
            two_phase = nph != 1
        self.two_phase = bool(two_phase) and grad_sync is not None and hasattr(net, "layer4")
        if self.two_phase:
            net.phase_boundaries = 2 if nph >= 3 else 1
        if grad_sync is not None:
            net.set_grad_ready_hook(None)          # buckets are issued between / after the replays, not from inside a capture
        self.steps_done = 0
        self.phase1 = None                          # [(param, grad)] finished at the first phase boundary
        self.phases = []                            # per boundary: the (param, grad) pairs finished since the previous one
        self.graphs = []                            # data-parallel mode: the graphs of the step, in replay order
        # data-parallel mode: every parameter gradient lives in one flat arena, in the order the backward produces them;
        # the collectives then run in place on contiguous ranges (no flatten / copy-back passes over 165 MB per step)
        self.arena = None
        if grad_sync is not None and os.environ.get("YV1_GRAD_ARENA", "1") != "0":
            self.arena = ops.GradArena(net, images.device)
            grad_sync.arena = self.arena
        snap = [(t, t.detach().clone()) for t in net.state_dict().values()] if preserve_state else None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up on a side stream: lazy inits, allocator, hipFuncSetAttribute
            for _ in range(warmup):
                self._body(eager=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if snap is not None:
            with torch.no_grad():
                for t, c in snap:
                    t.copy_(c)
                for p in net.parameters():
                    st = optimizer.state.get(p)
                    if st and 'momentum_buffer' in st:
                        st['momentum_buffer'].zero_()      # zeros == "no buffer yet": buf = momentum*0 + g on the first step
            ops.bump_weight_epoch()
            self.steps_done = 0
            torch.cuda.synchronize()
        if self.in_graph_step:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = self._body(eager=False)
        else:
            self._capture_data_parallel(side)

    # ---- one rank: autograd drives the step
    def _body(self, eager):
        if not self.in_graph_step:
            loss = self._direct()
            if eager:
                self._reduce_and_step(None)
                self.steps_done += 1
            return loss
        pred = self.net(self.images)
        loss = self.loss_layer(pred, self.target)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        if eager:
            self.steps_done += 1
        return loss

    # ---- several ranks: the executors are driven directly (same kernels, same order as the autograd path), so the
    # capture can be cut in the middle of the backward pass from this thread
    def _direct(self, boundary=None):
        from . import ops
        net = self.net
        if not net.training:
            raise RuntimeError("GraphedStep needs the network in training mode")
        with torch.no_grad():
            pred, rec = net._run_forward(self.images, True, True)
            loss, gpred = self.loss_layer.loss_and_grad(pred, self.target)
            net.set_phase_boundary(boundary)
            ops.set_grad_arena(self.arena)
            try:
                grads = net._run_backward(rec, gpred)
            finally:
                net.set_phase_boundary(None)
                ops.set_grad_arena(None)
        for p in net.parameters():
            p.grad = grads.get(p)
        return loss

    def _capture_data_parallel(self, stream):
        import gc
        self.graphs = [torch.cuda.CUDAGraph()]
        self.phases = []
        gc.collect()
        torch.cuda.synchronize()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            # ``state`` instead of attributes of self: the callback handed to the backward executor must not keep this object
            # alive (it only needs the lists it fills), and ``open`` says whether a capture is in progress, so that an
            # exception between two captures is re-raised as it is instead of being buried under a second capture_end()
            state = {"graphs": self.graphs, "phases": self.phases, "seen": set(), "open": False}
            state["graphs"][0].capture_begin()
            state["open"] = True

            def boundary(grads, state=state):
                # gradients finished since the previous boundary; end this graph, open the next one in the same pool
                seen = state["seen"]
                state["phases"].append([(p, g) for p, g in grads.items() if id(p) not in seen])
                seen.update(id(p) for p in grads)
                state["graphs"][-1].capture_end()
                state["open"] = False
                nxt = torch.cuda.CUDAGraph()
                nxt.capture_begin(pool=state["graphs"][0].pool())
                state["graphs"].append(nxt)
                state["open"] = True
            try:
                self.loss = self._direct(boundary if self.two_phase else None)
            except BaseException:
                if state["open"]:                      # leave capture mode, but let the ORIGINAL error travel
                    try:
                        state["graphs"][-1].capture_end()
                    except Exception:
                        pass
                    state["open"] = False
                raise
            state["graphs"][-1].capture_end()
            state["open"] = False
            state["seen"] = None
        torch.cuda.current_stream().wait_stream(stream)
        if self.two_phase and not self.phases:
            raise RuntimeError("the backward executor never reached its phase boundary")
        self.graph = self.graphs[0]
        self.graph2 = self.graphs[1] if len(self.graphs) > 1 else None
        self.phase1 = self.phases[0] if self.phases else None

    def _reduce_and_step(self, early):
        """``early``: parameters whose all-reduce is already in flight."""
        done = set(id(p) for p, _ in early) if early else ()
        self.sync.reduce_all([(p, p.grad) for p in self.net.parameters() if p.grad is not None and id(p) not in done])
        self.opt.step()

    def _dp_sequence(self, replays):
        """The multi-rank step around the replays: graph k -> asynchronous all-reduce of the gradients finished at its
        phase boundary (RCCL's stream, beside the next replay) -> ... -> last graph -> one more collective for the rest
        -> optimizer.  Kept apart from the graph objects so the CPU gloo test drives this exact sequence."""
        from . import ops
        early = []
        for k, replay in enumerate(replays):
            replay()
            if k < len(self.phases) and k + 1 < len(replays):
                self.sync.start(self.phases[k])
                early.extend(self.phases[k])
        ops.bump_weight_epoch()
        self._reduce_and_step(early or None)

    def __call__(self, lr):
        from . import ops
        if self.opt is None:
            raise RuntimeError("this GraphedStep has been closed")
        self.opt.set_lr(lr)
        if self.in_graph_step:
            self.graph.replay()
            ops.bump_weight_epoch()
            return self.loss
        self._dp_sequence([g.replay for g in self.graphs])
        return self.loss


def sync_replicas(net):
    """Makes every rank's replica identical to rank 0's before the first step: broadcasts every state_dict tensor
    (parameters AND BatchNorm buffers).  The reference has a single replica (nn.DataParallel(device_ids=[0]),
    train.py:34,:80); here each process builds its own, and gradient averaging only keeps replicas equal if they start
    equal.  c10d writes the tensors in place without bumping torch's version counter, so the bf16 weight shadows are
    told explicitly."""
    import torch.distributed as dist
    from . import ops
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in net.state_dict().values():
            dist.broadcast(t, 0)
    ops.bump_weight_epoch()


def main(argv=None, return_losses=False):
    """The epoch loop of train.py:144-209.  Default path = the one bench.py measures: fused HIP SGD, the whole step
    replayed from hipGraphs (``GraphedStep``), static input buffers filled by the device prefetcher, the loss read back
    only when the log line needs it (every 5th iteration, train.py:175-177) -- ``average_loss`` is accumulated on the
    device (and the loss layer's four per-iteration component log lines, v1Loss.py:107-116, which need a host sync each,
    are off).  ``--eager`` launches kernel by kernel (same kernels, same results bit for bit, the component log lines on);
    ``--torch-sgd`` additionally swaps the fused optimizer for torch.optim.SGD (train.py:84)."""
    ap = argparse.ArgumentParser(description="YOLO-v1 training on MI355X (reference train.py surface)")
    ap.add_argument("--backbone", default=DEFAULTS["backbone"], choices=["densenet", "resnet"])
    ap.add_argument("--S", type=int, default=DEFAULTS["S"])
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--epochs", type=int, default=DEFAULTS["num_epochs"])
    ap.add_argument("--iters-per-epoch", type=int, default=20, help="synthetic data: iterations per epoch")
    ap.add_argument("--save-dir", default=None)
    ap.add_argument("--seed", type=int, default=0, help="weight-init seed (every rank ends up with rank 0's weights)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying hipGraphs")
    ap.add_argument("--torch-sgd", action="store_true", help="torch.optim.SGD (train.py:84) instead of the fused HIP SGD; "
                                                             "implies --eager")
    ap.add_argument("--pretrained", default=None, help="ImageNet state_dict file (torchvision keys) for the by-name "
                                                       "initialisation of train.py:59-78; no network access here")
    ap.add_argument("--resume", default=None, help="checkpoint written by this script or by the reference")
    ap.add_argument("--fp8-forward", action="store_true", help="ResNet: forward convolutions on the fp8 (e4m3) MFMA path")
    ap.add_argument("--list-file", default=None, help="darknet-style image list (labels next to the images, as the reference's "
                                                     "datasets/train.txt); implies --loader; images are decoded by PIL")
    ap.add_argument("--loader", action="store_true",
                    help="feed every step from a DataLoader (synthetic yoloDataset, 4 workers as train.py:119) through "
                         "the device prefetcher + device target encoder instead of one resident batch")
    ap.add_argument("--val-list", default=None, help="validation image list (train.py:114-131): per-epoch little-mAP, full "
                                                    "mAP above --full-map-thresh, <name>_best.pth on improvement")
    ap.add_argument("--val-synthetic", type=int, default=0, help="validate on this many synthetic samples instead (plumbing)")
    ap.add_argument("--little-val-num", type=int, default=750)                                   # train.py:129
    ap.add_argument("--full-map-thresh", type=float, default=0.585)                              # train.py:141 (HEAD side)
    ap.add_argument("--workers", type=int, default=4)                                            # train.py:119
    args = ap.parse_args(argv)
    from . import checkpoint
    from . import distributed as ydist
    from . import ops
    from .utils.YOLODataLoader import synthetic_batch
    from .utils.utils import create_logger
    validating = bool(args.val_list or args.val_synthetic)
    rank, world, device = ydist.init_from_env(timeout_s=4 * 3600 if validating else None)
    bs = args.batch_size or (16 if args.backbone == 'resnet' else DEFAULTS["batch_size"])      # train.py:68
    opt_name = 'sgd'
    base = args.save_dir or '%s_%s_cellSize%d/' % (args.backbone, opt_name, args.S)              # train.py:91
    logger = create_logger(base, 'train') if rank == 0 else None
    torch.manual_seed(args.seed)
    fast = not (args.eager or args.torch_sgd)
    net, loss_layer, opt = build(args.backbone, args.S, DEFAULTS["B"], DEFAULTS["clsN"], bs, device, logger=logger,
                                 quiet=(rank != 0 or fast), fused_optimizer=not args.torch_sgd)
    if args.fp8_forward:
        net.fp8_forward = True
    if args.pretrained:
        taken = checkpoint.init_from_pretrained(net, torch.load(args.pretrained, map_location='cpu', weights_only=True))
        if logger:
            logger.info('initialised %d tensors by name from %s' % (len(taken), args.pretrained))
    if args.resume:
        checkpoint.load(net, args.resume, device)
    sync_replicas(net)                               # after init / --pretrained / --resume, before the first forward
    sync = ydist.GradSync(net) if world > 1 else None

    # ---- data: one resident synthetic batch, or a DataLoader behind the device prefetcher (sharded over the ranks)
    images, target = synthetic_batch(bs, args.S, seed=1234 + rank, device=device)
    feed = sampler = None
    if args.loader or args.list_file:
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        from .utils.YOLODataLoader import DevicePrefetcher, collate_raw, yoloDataset
        ds = yoloDataset(args.list_file, S=args.S, B=DEFAULTS["B"], C=DEFAULTS["clsN"], raw_targets=True, seed=1234,
                         length=bs * args.iters_per_epoch * world)
        if world > 1:        # every rank walks its own 1/world of each epoch's permutation (the reference: one process)
            sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=args.seed, drop_last=True)
        loader = DataLoader(ds, batch_size=bs, shuffle=(sampler is None), sampler=sampler, num_workers=args.workers,
                            collate_fn=collate_raw, drop_last=True)
        args.iters_per_epoch = max(1, len(loader))
        feed = DevicePrefetcher(loader, device, args.S, DEFAULTS["B"], DEFAULTS["clsN"],
                                out_images=images if fast else None, out_target=target if fast else None)

    # ---- validation set (train.py:114-131)
    val_ds = gt_little = gt_full = None
    if args.val_list or args.val_synthetic:
        from .utils.utils import prep_test_data
        from .utils.YOLODataLoader import yoloDataset
        if args.val_list:
            val_ds = yoloDataset(args.val_list, train=False, with_file_path=True, S=args.S)
            gt_full = prep_test_data(args.val_list, little_test=None)
            gt_little = prep_test_data(args.val_list, little_test=args.little_val_num)
        else:
            val_ds = yoloDataset(None, train=False, with_file_path=True, S=args.S, length=args.val_synthetic, seed=4321)
            gt_full = val_ds.synthetic_ground_truth()
            gt_little = val_ds.synthetic_ground_truth(min(args.little_val_num, args.val_synthetic))

    graphed = None
    if fast:
        for g in opt.param_groups:
            g['lr'] = 0.0                                          # capture warm-up steps must not move the weights
        graphed = GraphedStep(net, loss_layer, opt, images, target, sync, warmup=2, preserve_state=True)
    total_dev = torch.zeros((), dtype=torch.float32, device=device)
    history = torch.zeros(max(1, args.epochs * args.iters_per_epoch), dtype=torch.float32, device=device) if return_losses else None
    lr, it = DEFAULTS["learning_rate"], 0
    best_mAP, last_little_mAP = 0.0, 0.0
    for epoch in range(args.epochs):
        net.train()
        if sampler is not None:
            sampler.set_epoch(epoch)
        if logger:
            logger.info('\n\nStarting epoch %d / %d' % (epoch + 1, args.epochs))
            logger.info('Learning Rate for this epoch: {}'.format(opt.param_groups[0]['lr']))
        t_epoch = time.perf_counter()
        total_dev.zero_()
        batches = iter(feed) if feed is not None else None
        for i in range(args.iters_per_epoch):
            t0 = time.perf_counter()
            if batches is not None:
                images, target = next(batches)          # fast path: the prefetcher wrote the graph's static buffers
            it += 1
            lr = learning_rate_policy(it, epoch, lr, DEFAULTS["lr_adjust_map"])
            if graphed is not None:
                loss = graphed(lr)
            else:
                loss = train_step(net, loss_layer, opt, images, target, lr, sync)
            total_dev += loss.detach()                                                           # train.py:168, no host sync
            if history is not None:
                history[it - 1].copy_(loss.detach())
            if (i + 1) % 5 == 0 and logger:
                now, tot = float(loss.item()), float(total_dev.item())     # the only host syncs of the loop
                dt = time.perf_counter() - t0
                logger.info('Epoch [%d/%d], Iter [%d/%d] expect end in %.2f min. Loss: %.4f, average_loss: %.4f, '
                            'now learning rate: %f' % (epoch + 1, args.epochs, i + 1, args.iters_per_epoch,
                                                       dt * (args.iters_per_epoch - i + 1) // 60, now,
                                                       tot / (i + 1), lr))                       # train.py:177
        torch.cuda.synchronize(device) if torch.cuda.is_available() else None
        ep_s = time.perf_counter() - t_epoch
        if logger:
            logger.info('Epoch {} / {} finished, cost time {:.2f} min. expect {} min finish train.'.format(
                epoch, args.epochs, ep_s / 60, (ep_s / 60) * (args.epochs - epoch + 1)))
            logger.info('train throughput: %.1f img/s on this rank (%d iterations of batch %d)' % (
                args.iters_per_epoch * bs / ep_s, args.iters_per_epoch, bs))
        # ---- validation + checkpoints (train.py:187-209), rank 0 (DataParallel's single replica)
        if rank == 0:
            os.makedirs(base, exist_ok=True)
            if val_ds is not None:
                from copy import deepcopy
                from .utils.utils import run_test_mAP
                net.eval()
                test_mAP = 0.0
                data_len = int(len(val_ds) / bs)
                little = min(args.little_val_num, len(val_ds))
                now_little_mAP = run_test_mAP(net, deepcopy(gt_little), val_ds, data_len, S=args.S, device=device,
                                              logger=logger, little_test=little)
                if now_little_mAP > last_little_mAP and now_little_mAP > args.full_map_thresh:
                    test_mAP = run_test_mAP(net, deepcopy(gt_full), val_ds, data_len, S=args.S, device=device, logger=logger)
                last_little_mAP = now_little_mAP
                if test_mAP > best_mAP:
                    best_mAP = test_mAP
                    logger.info('get best test mAP %.5f' % best_mAP)
                    checkpoint.save(net, '%s/%s_%s_S%d_best.pth' % (base, args.backbone, opt_name, args.S))
                net.train()
            checkpoint.save(net, '%s/%s_%s_S%d_yolo.pth' % (base, args.backbone, opt_name, args.S))   # train.py:209
        # the other ranks wait (on the rendezvous store, not in a collective) until rank 0 has validated and saved
        ydist.rank0_gate("epoch%d" % epoch, rank, world)
    if graphed is not None:
        graphed.close()
    if return_losses:
        return history.cpu().tolist()[:it]


if __name__ == "__main__":
    main()
