"""Inference throughput of the eval-mode forward (+ batched decoder/NMS): bf16 path vs the fp8 executor (config 5).

    python tools/bench_infer.py [--S 14] [--batch 64] [--iters 20]
Prints one JSON line per variant.  Synthetic images, random-init weights (no checkpoints offline)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=14)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--graph", type=int, default=1)
    args = ap.parse_args()
    from yolo_v1_amd.backbones.OriginResNet import resnet50
    from yolo_v1_amd.infer_fp8 import ResNetFp8
    from yolo_v1_amd.utils.utils import decode_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = resnet50(S=args.S).to(dev).eval()
    x = torch.randn(args.batch, 3, 448, 448, device=dev)
    eng = ResNetFp8(net)
    gflop = {7: 34.733, 14: 32.721}[args.S]

    def run16():
        with torch.no_grad():
            return decode_batch(net(x), grid_num=args.S, B=2, thresh=0.005, nms_th=0.45)

    def run8():
        return decode_batch(eng(x), grid_num=args.S, B=2, thresh=0.005, nms_th=0.45)

    for name, fn in (("bf16 eval", run16), ("fp8 e4m3 eval", run8)):
        ms = timed(fn, args.iters)
        out = {"variant": name, "launch": "eager", "S": args.S, "batch": args.batch, "ms_per_batch": round(ms, 3),
               "images_per_sec": round(args.batch / ms * 1e3, 1), "conv_TFLOPs": round(args.batch * gflop / ms, 1)}
        print(json.dumps(out), flush=True)
        if args.graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            ms = timed(g.replay, args.iters)
            out.update({"launch": "hipGraph replay", "ms_per_batch": round(ms, 3),
                        "images_per_sec": round(args.batch / ms * 1e3, 1), "conv_TFLOPs": round(args.batch * gflop / ms, 1)})
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
