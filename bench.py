#!/usr/bin/env python
"""Training-throughput bench for the YOLO-v1 hot path on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one synthetic batch already resident in HBM:
forward (ResNet-50, 448x448, S=7) -> YOLOLossV1 -> zero_grad -> backward -> (RCCL gradient
average, overlapped with backward, when N > 1) -> SGD(momentum 0.99) step, i.e. the loop body
of the reference's train.py:158-172.  Per-GPU batch is fixed at 64 (weak scaling).
Rank 0 prints ONE JSON line; ``value`` is whole-job images/sec.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per image, SURVEY.md 8d / BASELINE.md 3 (conv MACs x2, fwd + dgrad + wgrad, stem has no dgrad)
TRAIN_GFLOP_PER_IMG = {("resnet", 7): 103.25, ("resnet", 14): 97.22, ("densenet", 7): 68.30, ("densenet", 14): 67.09}
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def source_fingerprint():
    """sha1 over the kernel sources and the launch sequences (csrc/*, backbones/*.py, ops.py, engine.py): what decides
    the HBM traffic of a step.  tools/pmc_traffic.py stores it with the measurement."""
    import hashlib
    h = hashlib.sha1()
    pk = os.path.join(ROOT, "yolo_v1_amd")
    files = [os.path.join(pk, "csrc", f) for f in sorted(os.listdir(os.path.join(pk, "csrc")))]
    files += [os.path.join(pk, "backbones", f) for f in sorted(os.listdir(os.path.join(pk, "backbones"))) if f.endswith(".py")]
    files += [os.path.join(pk, "ops.py"), os.path.join(pk, "engine.py")]
    for f in files:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(backbone, S, batch):
    """HBM bytes per step from the PMC passes committed under profiles/ (tools/pmc_traffic.py).  Returns
    (bytes or None, provenance dict): the value is refused (None) when the committed measurement is for another
    workload or was taken with different kernel sources than the ones running now."""
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
    for name in reversed(cands):                      # newest round first
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if t.get("workload") != "%s S=%d batch %d" % (backbone, S, batch):
            continue
        prov = {"file": "profiles/" + name, "measured_at_source": t.get("source_fingerprint"),
                "current_source": source_fingerprint()}
        if prov["measured_at_source"] != prov["current_source"]:
            prov["stale_value"] = t.get("hbm_bytes_per_step")
            prov["note"] = "refused: kernel sources changed since the PMC passes were taken"
            return None, prov
        return t.get("hbm_bytes_per_step"), prov
    return None, {"note": "no PMC measurement committed for this workload"}


def cpu_baseline(steps=60, warmup=3):
    """BASELINE.json configs[0]: ResNet-50 448x448 S=7 batch 2 fp32 on the host cores, through the CPU
    oracle (our restatement of the reference modules; the reference itself never travels)."""
    from oracle import train_step as ots
    # the GPU box gives a 1-GPU job a share of ~16 host cores; os.cpu_count() reports the whole machine, and
    # oversubscribing OpenMP threads on a quota stalls for minutes
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, avail)))
    P = ots.make_state("resnet", 7, seed=0)
    images, target = ots.synthetic_batch(2, 7)
    times = []
    ots.train_steps(P, images, target, 7, warmup + steps, "resnet", timings=times)
    t = sorted(times[warmup:])[len(times[warmup:]) // 2]
    return {"value": round(2.0 / t, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d timed steps (median) after %d warm-up of batch 2, ResNet-50 448x448 S=7 fp32, "
                      "oracle port of train.py:158-172" % (steps, warmup)}


# (Cin, Cout, k, stride, Hin, count): the distinct Bottleneck convolutions of ResNet-50 at 448x448, S=7
RESNET50_S7_CONVS = [(64, 64, 1, 1, 112, 1), (64, 64, 3, 1, 112, 3), (64, 256, 1, 1, 112, 4), (256, 64, 1, 1, 112, 2),
                     (256, 128, 1, 1, 112, 1), (128, 128, 3, 2, 112, 1), (128, 512, 1, 1, 56, 4), (256, 512, 1, 2, 112, 1),
                     (512, 128, 1, 1, 56, 3), (128, 128, 3, 1, 56, 3), (512, 256, 1, 1, 56, 1), (256, 256, 3, 2, 56, 1),
                     (256, 1024, 1, 1, 28, 6), (512, 1024, 1, 2, 56, 1), (1024, 256, 1, 1, 28, 5), (256, 256, 3, 1, 28, 5),
                     (1024, 512, 1, 1, 28, 1), (512, 512, 3, 2, 28, 1), (512, 2048, 1, 1, 14, 3), (1024, 2048, 1, 2, 28, 1),
                     (2048, 512, 1, 1, 14, 3), (512, 512, 3, 1, 14, 2), (512, 512, 3, 2, 14, 1), (512, 2048, 1, 1, 7, 3),
                     (2048, 2048, 1, 2, 14, 1), (2048, 512, 1, 1, 7, 2), (512, 512, 3, 1, 7, 2)]


def conv_kernel_roofline(batch, device, iters=5):
    """The dominant kernel family of the step -- the implicit-GEMM convolution (k_conv_ps / k_conv_dma: forward and data
    gradient of the 58 Bottleneck convolutions) -- on its own: every distinct layer shape launched ``iters`` times
    between HIP events on the launch stream, weighted by how often the step runs it.  Algorithmic flops / that time."""
    from yolo_v1_amd import ops
    flops = t_us = 0.0
    per_launch = []
    for ci, co, k, st, h, cnt in RESNET50_S7_CONVS:
        pad = 1 if k == 3 else 0
        oh = (h + 2 * pad - k) // st + 1
        x = ops.Act(torch.randn(batch, h, h, ci, device=device).to(torch.bfloat16))
        w = torch.nn.Parameter((torch.randn(co, ci, k, k, device=device) * 0.05).contiguous(memory_format=torch.channels_last))
        cw = ops.ConvWeights(w, k, st, pad)
        cw.refresh()
        y = ops.new_act(batch, oh, oh, co, device)
        dy = ops.Act(torch.randn(batch, oh, oh, co, device=device).to(torch.bfloat16))
        dx = ops.new_act(batch, h, h, ci, device)
        fl = 2.0 * batch * oh * oh * co * ci * k * k
        for fn in (lambda: ops.conv_fwd(x, cw, y, True), lambda: ops.conv_dgrad(dy, cw, dx)):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / iters * 1e3
            flops += cnt * fl
            t_us += cnt * us
            per_launch.append(us)
    n = 2 * sum(c[5] for c in RESNET50_S7_CONVS)
    tf = flops / t_us / 1e6
    return {"name": "k_conv_ps / k_conv_dma: implicit-GEMM NHWC convolution (LDS-DMA ring, persistent workgroups), forward + data "
                    "gradient of the Bottleneck convolutions",
            "launches_per_step": n, "avg_launch_us": round(t_us / n, 1), "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4),
            "note": "each distinct layer shape timed alone with HIP events (%d launches), weighted by its count in the "
                    "step; inside the step the same launches share the chip with the weight-gradient stream" % iters}


def host_input_rate(graphed, step, images, target, steps):
    """The boundary of the reference hands over host tensors (DataLoader batches, train.py:119-127): the same steps
    with every batch starting in pinned host memory.  Batch t+1 crosses PCIe on a copy stream while step t runs; the
    launch stream waits on the copy's event and moves the batch into the graph's static input buffers."""
    dev = images.device
    host = [(images.cpu().pin_memory(), target.cpu().pin_memory()) for _ in range(2)]
    stage = [(torch.empty_like(images), torch.empty_like(target)) for _ in range(2)]
    copy_stream = torch.cuda.Stream(dev)
    ready = [torch.cuda.Event(), torch.cuda.Event()]
    consumed = [torch.cuda.Event(), torch.cuda.Event()]
    cur = torch.cuda.current_stream(dev)

    def upload(slot):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[slot])            # the staging slot was drained by an earlier step
            stage[slot][0].copy_(host[slot][0], non_blocking=True)
            stage[slot][1].copy_(host[slot][1], non_blocking=True)
            ready[slot].record(copy_stream)

    for s_ in range(2):
        consumed[s_].record(cur)
    upload(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        slot = i & 1
        upload(slot ^ 1)                                      # next batch, overlapped with this step
        cur.wait_event(ready[slot])
        graphed.images.copy_(stage[slot][0])
        graphed.target.copy_(stage[slot][1])
        consumed[slot].record(cur)
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    mb = (images.numel() * images.element_size() + target.numel() * target.element_size()) / 1e6
    return {"value": round(images.shape[0] * steps / dt, 2), "unit": "images/sec", "ms_per_step": round(dt / steps * 1e3, 3),
            "h2d_MB_per_step": round(mb, 1), "note": "pinned host fp32 batch -> copy stream -> static graph inputs, "
            "one batch ahead; PCIe-inclusive, informative only"}


def other_training_config(backbone, S, batch, device, fp8_forward=False, steps=8, warmup=3):
    """One more BASELINE.json configuration in the same process, after the headline timed region: the same step
    (forward + loss + backward + fused SGD, hipGraph replay) for another backbone / grid, timed with HIP events on the
    launch stream.  Reported under ``other_configs`` -- never in ``value``."""
    from yolo_v1_amd.train import GraphedStep, build
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    torch.manual_seed(0)
    net, loss_layer, opt = build(backbone, S, 2, 20, batch, device, quiet=True, fused_optimizer=True)
    if fp8_forward:
        net.fp8_forward = True
    images, target = synthetic_batch(batch, S, seed=1234, device=device)
    for g in opt.param_groups:
        g['lr'] = 1e-6
    graphed = GraphedStep(net, loss_layer, opt, images, target, None, warmup=warmup)
    it = graphed.steps_done
    for _ in range(2):
        it += 1
        graphed(it * 1e-6)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        it += 1
        loss = graphed(it * 1e-6)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    gflop = TRAIN_GFLOP_PER_IMG[(backbone, S)]
    tf = batch * gflop / ms                      # GFLOP / ms = TFLOP/s
    peak = PEAK_BF16_TFLOPS
    out = {"workload": "%s 448x448 S=%d B=2 C=20, per-GPU batch %d, fwd+loss+bwd+SGD, hipGraph replay"
                       % ("ResNet-50" if backbone == "resnet" else "DenseNet-121", S, batch),
           "dtype": "fp8-e4m3 forward GEMMs, bf16 backward" if fp8_forward else "bf16",
           "value": round(batch / ms * 1e3, 1), "unit": "images/sec", "ms_per_step": round(ms, 3), "steps": steps,
           "achieved_TFLOPs": round(tf, 1), "frac_of_bf16_peak": round(tf / peak, 4), "final_loss": round(float(loss.item()), 5)}
    if not fp8_forward:                          # PMC traffic of this configuration, when committed for these kernel sources
        traffic, prov = measured_traffic(backbone, S, batch)
        out["traffic"] = traffic
        out["hbm_GBps"] = round(traffic / (ms * 1e-3) / 1e9, 1) if traffic else None
        out["hbm_frac"] = round(traffic / (ms * 1e-3) / 8e12, 4) if traffic else None
        if not traffic:
            out["traffic_note"] = prov.get("note")
    graphed.close()            # retire the exec (train.GraphedStep.close: destruction is deferred, its pool stays until then)
    del graphed, net, opt, loss_layer
    torch.cuda.empty_cache()
    return out


def other_eval_config(S, batch, device, iters=10):
    """BASELINE config 5's evaluation leg: eval-mode ResNet-50 forward + batched decoder/NMS (utils/utils.py:389-418 as
    one batch), bf16 fused-epilogue path and the fp8 e4m3 executor, hipGraph replay."""
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.infer_fp8 import ResNetFp8
    from yolo_v1_amd.utils.utils import decode_batch
    torch.manual_seed(0)
    net = resnet50(S=S).to(device).eval()
    x = torch.randn(batch, 3, 448, 448, device=device)
    eng = ResNetFp8(net)
    res = {}
    for name, fwd in (("bf16", lambda: net(x)), ("fp8_e4m3", lambda: eng(x))):
        def fn():
            with torch.no_grad():
                return decode_batch(fwd(), grid_num=S, B=2, thresh=0.005, nms_th=0.45)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res[name] = {"value": round(batch / ms * 1e3, 1), "unit": "images/sec", "ms_per_batch": round(ms, 3)}
        g.reset()
        del g
    res["workload"] = "ResNet-50 448x448 S=%d eval-mode forward + batched decoder/NMS, batch %d, hipGraph replay" % (S, batch)
    del eng, net
    torch.cuda.empty_cache()
    return res


def self_launch(nproc, argv=None):
    """Runs this script as ``nproc`` ranks under torch.distributed.run (one process per GPU, RCCL rendezvous on
    127.0.0.1, a free port) as a CHILD process and returns its exit code.  The children inherit stdout/stderr, so rank 0's
    JSON line is the only line on stdout."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    cmd += list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--backbone", default="resnet", choices=["resnet", "densenet"])
    ap.add_argument("--S", type=int, default=7)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-roofline", type=int, default=1,
                    help="after the timed region, time the dominant kernel family (implicit-GEMM conv) per layer shape and "
                         "report roofline.dominant_kernel (0 when profiling: keeps the trace to the training steps)")
    ap.add_argument("--host-input", type=int, default=1,
                    help="after the timed region, also time the same steps fed from pinned host memory (fp32 batch per "
                         "step over PCIe, copy stream, one batch ahead) and report it as host_input -- never as value")
    ap.add_argument("--other-configs", type=int, default=1,
                    help="after the headline timed region also time BASELINE configs 3 and 5 (DenseNet-121 S=7, ResNet-50 "
                         "S=14 bf16 / fp8-forward training, S=14 eval + batched NMS) and report them under other_configs")
    ap.add_argument("--fused-sgd", type=int, default=1)
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a captured hipGraph")
    ap.add_argument("--fp8-forward", action="store_true",
                    help="BASELINE config 5: forward convolutions on the fp8 (e4m3) MFMA path, bf16 backward (ResNet only; "
                         "NOT the headline configuration, which computes in bf16)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start N fresh ranks (one per GPU) BEFORE this process has touched the GPU -- no
        # torch.cuda call, no libyv1 load so far -- relay their output and exit with their code.  Never exec.
        raise SystemExit(self_launch(args.gpus))

    from yolo_v1_amd import distributed as ydist
    from yolo_v1_amd import _lib
    from yolo_v1_amd.train import GraphedStep, build, learning_rate_policy, train_step
    from yolo_v1_amd.utils.YOLODataLoader import synthetic_batch
    _lib.lib()                                   # fail loudly if the HIP library is missing
    rank, world, device = ydist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))

    torch.manual_seed(0)
    net, loss_layer, opt = build(args.backbone, args.S, 2, 20, args.batch, device, quiet=True,
                                 fused_optimizer=bool(args.fused_sgd))
    if args.fp8_forward:
        if args.backbone != "resnet":
            raise SystemExit("--fp8-forward: the fp8 forward path covers the ResNet backbone")
        net.fp8_forward = True
    use_dist = torch.distributed.is_initialized()
    sync = ydist.GradSync(net) if use_dist else None
    if use_dist:                                 # same initial weights on every rank
        for p in net.state_dict().values():
            torch.distributed.broadcast(p, 0)
    images, target = synthetic_batch(args.batch, args.S, seed=1234 + rank, device=device)
    lr_map = {1: 0.001, 75: 0.0001, 115: 0.00001}

    lr, it = 0.0, 0
    use_graph = bool(args.graph) and bool(args.fused_sgd)
    graphed = None
    if use_graph:
        for g in opt.param_groups:
            g['lr'] = 1e-6
        graphed = GraphedStep(net, loss_layer, opt, images, target, sync, warmup=min(3, max(1, args.warmup)))
        it = graphed.steps_done
        lr = it * 1e-6

    def step():
        nonlocal lr, it
        it += 1
        lr = learning_rate_policy(it, 0, lr, lr_map)
        if graphed is not None:
            return graphed(lr)
        return train_step(net, loss_layer, opt, images, target, lr, sync)

    for _ in range(max(0, args.warmup - (graphed.steps_done if graphed else 0))):
        step()
    torch.cuda.synchronize()
    if use_dist:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        loss = step()
    ev1.record()
    torch.cuda.synchronize()
    if use_dist:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)               # HIP events on the stream every kernel of the step is launched on
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())

    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / elapsed
        gflop = TRAIN_GFLOP_PER_IMG[(args.backbone, args.S)]
        step_s_dev = dev_ms / 1e3 / args.steps
        achieved = args.batch * gflop / 1e3 / step_s_dev          # TFLOP/s on one GPU, device time of the step
        traffic, traffic_prov = (measured_traffic(args.backbone, args.S, args.batch) if not args.fp8_forward
                                 else (None, {"note": "fp8 forward: not measured"}))
        out = {
            "metric": "images/sec training (ResNet-50 448^2, S=7)" if args.backbone == "resnet" and args.S == 7
            else "images/sec training (%s 448^2, S=%d)" % (args.backbone, args.S),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8-e4m3 forward GEMMs, bf16 backward" if args.fp8_forward else "bf16",
            "data": "synthetic",
            "config": {"workload": "%s 448x448 S=%d B=2 C=20 bf16, per-GPU batch %d, fwd+loss+bwd+SGD(momentum 0.99)%s"
                                   % ("ResNet-50" if args.backbone == "resnet" else "DenseNet-121", args.S, args.batch,
                                      "+RCCL grad all-reduce" if world > 1 else ""),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "optimizer": "fused HIP SGD" if args.fused_sgd else "torch.optim.SGD",
                       "launch": "hipGraph replay of the whole step" if use_graph else "eager launches"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                         "traffic": traffic, "traffic_provenance": traffic_prov,
                         # informative: the same step against the HBM roofline (PMC bytes / device time, 8 TB/s peak)
                         "hbm_GBps": round(traffic / step_s_dev / 1e9, 1) if traffic else None,
                         "hbm_frac": round(traffic / step_s_dev / 8e12, 4) if traffic else None,
                         "note": "whole training step of one GPU: %.2f algorithmic conv GFLOP/img x %d img / %.3f ms "
                                 "(HIP-event time of the step on the launch stream)" % (gflop, args.batch, step_s_dev * 1e3)},
            "final_loss": round(final_loss, 5),
        }
        if world == 1 and args.host_input and graphed is not None:
            out["host_input"] = host_input_rate(graphed, step, images, target, args.steps)
        if world == 1 and args.kernel_roofline and args.backbone == "resnet" and args.S == 7 and not args.fp8_forward:
            out["roofline"]["dominant_kernel"] = conv_kernel_roofline(args.batch, device)
        headline = args.backbone == "resnet" and args.S == 7 and not args.fp8_forward
        if world == 1 and args.other_configs and headline and graphed is not None:
            # BASELINE.json configs 3 and 5 in the same run (driver-visible), after the headline timed region
            graphed.close()                # retire the headline step's hipGraphExec before the next captures (train.GraphedStep)
            graphed = None
            torch.cuda.empty_cache()
            oc = {}
            for key, kw in (("densenet121_S7", dict(backbone="densenet", S=7)),
                            ("resnet50_S14", dict(backbone="resnet", S=14)),
                            ("resnet50_S14_fp8_forward", dict(backbone="resnet", S=14, fp8_forward=True))):
                try:
                    oc[key] = other_training_config(batch=args.batch, device=device, **kw)
                except Exception as e:                     # an auxiliary measurement must not take the headline line down
                    oc[key] = {"error": repr(e)}
            try:
                oc["resnet50_S14_eval_nms"] = other_eval_config(14, args.batch, device)
            except Exception as e:
                oc["resnet50_S14_eval_nms"] = {"error": repr(e)}
            out["other_configs"] = oc
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
