"""Data-parallel gradient averaging, one process per GPU, RCCL over xGMI.

The reference only has ``nn.DataParallel(device_ids=[0])`` (train.py:34,:80): single process,
one device.  Here every rank owns one MI355X and a full replica; images shard across ranks
(weak scaling), BatchNorm statistics stay per-rank (what DataParallel does), and the only
exchange is the gradient average.

``GradSync`` plugs into the backbone's backward executor: as soon as a residual block's weight
gradients have been launched, they are flattened into a bucket and ``all_reduce`` is issued
asynchronously -- torch's NCCL(=RCCL) process group runs it on its own HIP stream, ordered after
the producing kernels by an event, so it overlaps with the dgrad/wgrad launches of the earlier
layers.  ``finish()`` (called before ``optimizer.step()``) waits, scales by 1/world and scatters the
averaged values back into the gradient tensors autograd handed to the parameters.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets default to 32 MB so each ring step
moves a few MB per link -- large enough to run at link rate, small enough that the last bucket
(stem + layer1) does not leave a long un-overlapped tail.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, timeout_s=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE (torch.distributed.run contract).  Returns
    (rank, world_size, device).  ``timeout_s``: collective timeout of the process group (default: the backend's, 10 min
    for RCCL) -- train.main raises it when rank 0 validates between epochs while the other ranks already wait."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    force = os.environ.get("YV1_FORCE_DIST") == "1"      # rehearse the multi-rank code path with a group of one
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {"device_id": device} if use_cuda else {}
        if timeout_s is not None:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world, **kw)
    return rank, world, device


def rank0_gate(tag, rank, world, timeout_s=6 * 3600):
    """Host-side gate on the rendezvous store (NOT a collective): rank 0 calls it when it is done with ``tag`` (e.g. the
    per-epoch validation + checkpoint, train.py:187-209, which only rank 0 runs -- DataParallel's single replica), every
    other rank blocks here until then.  Without it the other ranks enter the next epoch, enqueue their first all-reduce and
    sit in it while rank 0 still walks the validation set; past the process group's timeout the RCCL watchdog would abort
    the job.  A store wait has no watchdog and holds no GPU work."""
    if world <= 1 or not dist.is_initialized():
        return
    import datetime
    store = dist.distributed_c10d._get_default_store()
    key = "yv1_gate/%s" % tag
    if rank == 0:
        store.set(key, "1")
    else:
        store.wait([key], datetime.timedelta(seconds=float(timeout_s)))


def _flat_memory_view(t):
    """1-D view over a dense tensor's storage range (memory order, no copy)."""
    if t.is_contiguous():
        return t.view(-1)
    return t.as_strided((t.numel(),), (1,), t.storage_offset())


class GradSync:
    """Bucketed, asynchronous gradient averaging across the process group."""

    def __init__(self, net=None, bucket_mb=32, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.pending, self.pending_bytes = [], 0
        self.inflight = []
        self.buckets_issued = 0
        self.in_place_buckets = 0             # buckets reduced in place inside an ops.GradArena
        self.arena = None                     # set by train.GraphedStep (data-parallel mode)
        if net is not None:
            net.set_grad_ready_hook(self.on_ready)

    def on_ready(self, pairs):
        """Called by the backward executor with (param, grad) pairs whose kernels have been launched."""
        for p, g in pairs:
            self.pending.append((p, g))
            self.pending_bytes += g.numel() * g.element_size()
        if self.pending_bytes >= self.bucket_bytes:
            self._flush()

    def start(self, pairs):
        """Issue one asynchronous collective for ``pairs`` now (whatever the bucket size); ``finish()`` /
        ``reduce_all()`` later waits for it."""
        self.pending.extend(pairs)
        self.pending_bytes += sum(g.numel() * g.element_size() for _, g in pairs)
        self._flush()

    def reduce_all(self, pairs):
        """Everything in ONE collective (nothing left to overlap with: used after a hipGraph replay of
        forward+backward, where one large message runs the xGMI links at their best rate)."""
        self.pending.extend(pairs)
        self._flush()
        self.finish()

    def _arena_range(self):
        """The pending gradients as ONE contiguous range of the active ops.GradArena, or None (no arena, a gradient that
        does not live in it, or a range that would cover other parameters' slices)."""
        from . import ops
        arena = ops._ARENA[0] if self.arena is None else self.arena
        if arena is None:
            return None
        base = arena.flat.untyped_storage().data_ptr()
        need = 0
        for p, g in self.pending:
            r = arena.ranges.get(id(p))
            if r is None or g.untyped_storage().data_ptr() != base:
                return None
            need += (r[1] + 3) // 4 * 4
        lo, hi = arena.span([p for p, _ in self.pending])
        if hi - lo != need:
            return None
        return arena.flat[lo:hi]

    def _flush(self):
        if not self.pending:
            return
        flat = self._arena_range()
        in_place = flat is not None
        if not in_place:
            flat = torch.cat([_flat_memory_view(g) for _, g in self.pending])
        work, averaged = None, False
        if dist.is_initialized():
            # RCCL averages in the collective itself (no extra pass over the bucket); gloo only sums
            averaged = flat.is_cuda and dist.get_backend(self.group) == "nccl"
            op = dist.ReduceOp.AVG if averaged else dist.ReduceOp.SUM
            work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        self.inflight.append((work, averaged, flat, [p for p, _ in self.pending],
                              [(g.numel(), tuple(g.shape), tuple(g.stride())) for _, g in self.pending], in_place))
        self.in_place_buckets += 1 if in_place else 0
        self.pending, self.pending_bytes = [], 0
        self.buckets_issued += 1

    def finish(self):
        """Wait for every bucket, average, write the result into ``param.grad``.  Call once per step,
        after ``loss.backward()`` and before ``optimizer.step()``."""
        self._flush()
        for work, averaged, flat, params, metas, in_place in self.inflight:
            if work is not None:
                work.wait()
            if self.world > 1 and not averaged:
                flat.mul_(1.0 / self.world)
            if in_place:                   # param.grad IS a view of the reduced range (ops.GradArena): nothing to copy
                continue
            chunks = flat.split([m[0] for m in metas])
            # each chunk holds the gradient in the memory order it was produced in; view it with that
            # layout so the copy is a straight memcpy when param.grad has the same strides
            src = [c.as_strided(m[1], m[2]) for c, m in zip(chunks, metas)]
            torch._foreach_copy_([p.grad for p in params], src)
        self.inflight = []
